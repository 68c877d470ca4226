// bf16 MFMA GEMM for gfx950 with fused epilogues: the contraction engine of the T5 blocks.
//
//   C[M,N] (+)= op(A) . op(B)^T-style product, fp32 accumulation, where
//     trans_a = 0 : A is stored [M][K] (K contiguous)      trans_a = 1 : A is stored [K][M]
//     trans_b = 0 : B is stored [N][K] (K contiguous)      trans_b = 1 : B is stored [K][N]
//   so (0,0) is y = x W^T (forward), (0,1) is dx = dy W (input gradient) and (1,1) is
//   dW = dy^T x (weight gradient, contraction over tokens, split over workgroups, fp32 atomics).
//
// Structure: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave, 4x4
// v_mfma_f32_16x16x32_bf16 accumulators), K step 64.  Operand tiles go HBM -> LDS directly with
// buffer_load_dwordx4 ... lds (16 B per lane, 1 KiB per wave instruction); the LDS image is
// lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and again
// on the fragment read.  K-contiguous tiles are read with ds_read_b128; K-strided tiles
// (transposed operands) are read with ds_read_b64_tr_b16.  Out-of-range rows rely on the buffer
// descriptor's range check (loads beyond num_records return zero), so ragged M/N and a ragged
// token contraction need no host-side padding.
#include "gemm.h"

namespace {

constexpr int BK = 64;

// Tile configurations: {BM, BN, waves along M, waves along N, LDS buffers}.
//   small: 128x128, 4 waves, one LDS buffer, two barriers per K step; 3-4 workgroups per CU hide the
//          staging latency.  Used for small outputs (decoder, 768-wide weight gradients).
//   large: 256x256, 8 waves (2x4, 128x64 per wave), two LDS buffers (128 KiB): the LDS-DMA of K step
//          t+1 is in flight while step t is multiplied; one barrier per K step; one workgroup per CU.
}  // namespace
struct CfgSmall { static constexpr int BM = 128, BN = 128, WGM = 2, WGN = 2, NBUF = 1; };
struct CfgLarge { static constexpr int BM = 256, BN = 256, WGM = 2, WGN = 4, NBUF = 2; };
namespace {

// Per-lane source offset (bytes, relative to the tile's first row / column) of the 16-byte chunk
// that lands at lane-linear LDS position g (16-byte units) of a tile image.
//   K-contiguous operand : image [ROWS][64] bf16, 128-byte rows; chunk c of row r sits at c ^ ((r>>1)&7).
//   K-strided operand    : image [64][ROWS] bf16; the 32-byte unit u of k-row k sits at u ^ h(k),
//                          h(k) = (k&3) | ((k>>3)&1)<<2  (conflict-free ds_read_b64_tr_b16).
template <bool TRANS, int ROWS>
__device__ __forceinline__ uint32_t stage_offset(int g, int ld) {
    if (!TRANS) {
        const int r = g >> 3, cs = g & 7;
        const int c = cs ^ ((r >> 1) & 7);
        return (uint32_t)(r * ld + c * 8) * 2u;
    } else {
        constexpr int CPR = ROWS / 8;                 // 16-byte chunks per k-row
        const int k = g / CPR, cs = g % CPR;
        const int hk = (k & 3) | (((k >> 3) & 1) << 2);
        const int c = (((cs >> 1) ^ hk) << 1) | (cs & 1);
        return (uint32_t)(k * ld + c * 8) * 2u;
    }
}

// 16 rows x 32 k fragment (one v_mfma_f32_16x16x32_bf16 operand) for rows rb..rb+15, k-step kk.
template <bool TRANS, int ROWS>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int rb, int kk, int lane) {
    if (!TRANS) {
        const int row = rb + (lane & 15);
        const int c = kk * 4 + (lane >> 4);
        const int addr = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
        return *reinterpret_cast<const bf16x8*>(tile + addr);
    } else {
        constexpr int RB = ROWS * 2;                  // bytes per k-row
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int u = rb >> 4;
        const int hk = q | ((g & 1) << 2);
        const int k0 = kk * 32 + 8 * g + q;
        const int addr0 = k0 * RB + ((u ^ hk) << 5) + p * 8;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + addr0));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + addr0 + 4 * RB));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <int EPI>
__device__ __forceinline__ void store_quad(const GemmParams& p, int m, int n, const f32x4& a) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = a[r] * p.alpha;
    const size_t ci = (size_t)m * p.ldc + n;
    const uint64_t di = (uint64_t)m * (uint64_t)p.N + n;   // dropout counter: logical (row, col)
    if (EPI == EPI_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + ci) = f32x4{v[0], v[1], v[2], v[3]};
        return;
    }
    float f[4] = {1.f, 1.f, 1.f, 1.f};
    if (EPI == EPI_RES_DROP || EPI == EPI_RELU_DROP) {
        if ((p.N & 1) == 0) {                       // di is even: one hash per two elements
            drop_factor2(p.drop, di, f[0], f[1]);
            drop_factor2(p.drop, di + 2, f[2], f[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) f[r] = drop_factor(p.drop, di + r);
        }
    }
    if (EPI == EPI_RES_DROP || EPI == EPI_MASK) {
        const bf16x4 x = *reinterpret_cast<const bf16x4*>(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (EPI == EPI_RES_DROP) v[r] = bf2f(x[r]) + v[r] * f[r];
            else v[r] = bf2f(x[r]) > 0.f ? v[r] * p.mask_scale : 0.f;
        }
    } else if (EPI == EPI_RELU_DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f) * f[r];
    }
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + ci) = o;
}

template <class Cfg, bool TA, bool TB>
struct TileOps {
    static constexpr int BM = Cfg::BM, BN = Cfg::BN, NW = Cfg::WGM * Cfg::WGN;
    static constexpr int TM = BM / Cfg::WGM, TN = BN / Cfg::WGN, MI = TM / 16, NJ = TN / 16;
    static constexpr int IA = BM / (8 * NW), IB = BN / (8 * NW);     // LDS-DMA instructions per wave per K step
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + B_BYTES;

    // issue the LDS-DMA of one K step into LDS buffer `buf`
    static __device__ __forceinline__ void stage(char* smem, int buf, uint32_t sA, uint32_t sB, int w,
                                                 __amdgpu_buffer_rsrc_t rA, __amdgpu_buffer_rsrc_t rB,
                                                 const uint32_t (&offA)[IA], const uint32_t (&offB)[IB]) {
        char* lA = smem + buf * BUF_BYTES;
        char* lB = lA + A_BYTES;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(lA + (i * NW + w) * 1024), 16, offA[i], sA, 0, 0);
#pragma unroll
        for (int i = 0; i < IB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LDS_PTR(lB + (i * NW + w) * 1024), 16, offB[i], sB, 0, 0);
    }

    // fragments of half a K step (k = 32*kk .. +32) from LDS buffer `buf`
    static __device__ __forceinline__ void read_frags(const char* smem, int buf, int kk, int wm, int wn, int lane,
                                                      bf16x8 (&fa)[MI], bf16x8 (&fb)[NJ]) {
        const char* lA = smem + buf * BUF_BYTES;
        const char* lB = lA + A_BYTES;
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = read_frag<TB, BN>(lB, wn * TN + j * 16, kk, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = read_frag<TA, BM>(lA, wm * TM + i * 16, kk, lane);
    }
    template <bool NATURAL>
    static __device__ __forceinline__ void mfma_block(const bf16x8 (&fa)[MI], const bf16x8 (&fb)[NJ], f32x4 (&acc)[MI][NJ]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (NATURAL) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    }

    // multiply the K step held in LDS buffer `buf` into the accumulators
    template <bool NATURAL>
    static __device__ __forceinline__ void multiply(const char* smem, int buf, int wm, int wn, int lane, f32x4 (&acc)[MI][NJ]) {
        const char* lA = smem + buf * BUF_BYTES;
        const char* lB = lA + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MI], fb[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = read_frag<TB, BN>(lB, wn * TN + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = read_frag<TA, BM>(lA, wm * TM + i * 16, kk, lane);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (NATURAL)   // lane -> n, registers -> m
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    else           // swapped: lane -> m, registers -> 4 consecutive n
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
        }
    }
};

}  // namespace

// One unit of work of the persistent grid: an output tile and a slice of the contraction.
struct WorkItem {
    int m0, n0, kbeg, kend;
    __amdgpu_buffer_rsrc_t rA, rB;
};

template <class Cfg, bool TA, bool TB>
__device__ __forceinline__ WorkItem decode_item(const GemmParams& p, int L, int ntm, int ntn) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN;
    WorkItem it;
    const int ntiles = ntm * ntn;
    const int split = L / ntiles, t = L - split * ntiles;
    // neighbouring items share an operand panel: walk the SHORTER tile dimension fastest
    int tm, tn;
    if (ntm >= ntn) { tm = t / ntn; tn = t - tm * ntn; } else { tn = t / ntm; tm = t - tn * ntm; }
    it.m0 = tm * BM; it.n0 = tn * BN;
    it.kbeg = split * p.k_per_split;
    it.kend = min(p.K, it.kbeg + p.k_per_split);
    // Buffer descriptors: base at the tile's first row (or column), range = exactly the valid bytes,
    // so rows >= M/N and contraction rows >= K read as zero.
    const int rowsA = min(BM, p.M - it.m0), rowsB = min(BN, p.N - it.n0);
    const bf16* baseA = TA ? p.A + it.m0 : p.A + (size_t)it.m0 * p.lda;
    const bf16* baseB = TB ? p.B + it.n0 : p.B + (size_t)it.n0 * p.ldb;
    const uint32_t recA = TA ? (uint32_t)(((size_t)(p.K - 1) * p.lda + rowsA) * 2) : (uint32_t)(((size_t)(rowsA - 1) * p.lda + p.K) * 2);
    const uint32_t recB = TB ? (uint32_t)(((size_t)(p.K - 1) * p.ldb + rowsB) * 2) : (uint32_t)(((size_t)(rowsB - 1) * p.ldb + p.K) * 2);
    it.rA = __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, recA, 0x00020000);
    it.rB = __builtin_amdgcn_make_buffer_rsrc((void*)baseB, 0, recB, 0x00020000);
    return it;
}

// Item of workgroup b in round `round` of a persistent grid of G workgroups over T items, or -1.
// Within a round the workgroups that share an XCD (equal b % 8) take consecutive items.
__device__ __forceinline__ int item_of(int b, int round, int G, int T) {
    const int first = round * G;
    if (first >= T) return -1;
    const int count = min(G, T - first);
    if (b >= count) return -1;
    return first + xcd_remap(b, count);
}

template <class Cfg, int EPI>
__device__ __forceinline__ void epilogue(const GemmParams& p, int m0, int n0, int wm, int wn, int lane,
                                         f32x4 (&acc)[Cfg::BM / Cfg::WGM / 16][Cfg::BN / Cfg::WGN / 16]) {
    constexpr int TM = Cfg::BM / Cfg::WGM, TN = Cfg::BN / Cfg::WGN, MI = TM / 16, NJ = TN / 16;
    if (EPI == EPI_TOPK_DENSE || EPI == EPI_TOPK_FILTER) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn * TN + j * 16 + (lane & 15);
            if (n >= p.N) continue;
            const float thr = (EPI == EPI_TOPK_FILTER) ? p.tk_thr[n] : 0.f;
            unsigned long long* cand = p.tk_cand + (size_t)n * p.tk_cap;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * TM + i * 16 + (lane >> 4) * 4 + r;
                    if (m >= p.M) continue;
                    const float sc = acc[i][j][r];
                    if (EPI == EPI_TOPK_DENSE) {
                        cand[m] = topk_key(sc, p.tk_row_base + (unsigned int)m);
                    } else if (sc >= thr) {
                        const unsigned int pos = atomicAdd(p.tk_cnt + n, 1u);
                        if (pos < (unsigned int)p.tk_cap) cand[pos] = topk_key(sc, p.tk_row_base + (unsigned int)m);
                    }
                }
        }
        return;
    }
    if (EPI == EPI_ATOMIC_F32) {
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + wn * TN + j * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * TM + i * 16 + (lane >> 4) * 4 + r;
                    if (m < p.M && n < p.N) atomicAdd(C + (size_t)m * p.ldc + n, acc[i][j][r] * p.alpha);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * TM + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn * TN + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;          // N is a multiple of 4 (checked on the host)
            store_quad<EPI>(p, m, n, acc[i][j]);
        }
    }
}

// Persistent kernel: gridDim.x workgroups walk `total_items` work items round by round.  With two
// LDS buffers the first K step of the NEXT item is staged during the last K step of the current
// one, so the epilogue's stores overlap the next item's loads and no pipeline refill is exposed.
template <class Cfg, bool TA, bool TB, int EPI>
__global__ void __launch_bounds__(Cfg::WGM * Cfg::WGN * 64) gemm_bf16_kernel(GemmParams p, int ntm, int ntn, int total_items) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NW = Cfg::WGM * Cfg::WGN;
    constexpr int MI = BM / Cfg::WGM / 16, NJ = BN / Cfg::WGN / 16;
    constexpr bool NATURAL = (EPI == EPI_ATOMIC_F32 || EPI == EPI_TOPK_DENSE || EPI == EPI_TOPK_FILTER);
    using Ops = TileOps<Cfg, TA, TB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / Cfg::WGN, wn = w % Cfg::WGN;
    const int G = gridDim.x, b = blockIdx.x;

    uint32_t offA[Ops::IA], offB[Ops::IB];
#pragma unroll
    for (int i = 0; i < Ops::IA; ++i) offA[i] = stage_offset<TA, BM>((i * NW + w) * 64 + lane, p.lda);
#pragma unroll
    for (int i = 0; i < Ops::IB; ++i) offB[i] = stage_offset<TB, BN>((i * NW + w) * 64 + lane, p.ldb);
    const uint32_t kstrideA = TA ? (uint32_t)p.lda * 2u : 2u;
    const uint32_t kstrideB = TB ? (uint32_t)p.ldb * 2u : 2u;

    int L = item_of(b, 0, G, total_items);
    if (L < 0) return;
    WorkItem it = decode_item<Cfg, TA, TB>(p, L, ntm, ntn);
    f32x4 acc[MI][NJ];

    if (Cfg::NBUF == 1) {
        for (int round = 0; L >= 0; ) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k0 = it.kbeg; k0 < it.kend; k0 += BK) {
                Ops::stage(smem, 0, (uint32_t)k0 * kstrideA, (uint32_t)k0 * kstrideB, w, it.rA, it.rB, offA, offB);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                Ops::template multiply<NATURAL>(smem, 0, wm, wn, lane, acc);
                __syncthreads();
            }
            epilogue<Cfg, EPI>(p, it.m0, it.n0, wm, wn, lane, acc);
            L = item_of(b, ++round, G, total_items);
            if (L >= 0) it = decode_item<Cfg, TA, TB>(p, L, ntm, ntn);
        }
        return;
    }

    // ---- two LDS buffers, two fragment register sets, one barrier per K step placed MID-step:
    //   phase A: MFMA(set0 = first half of tile t)  ||  ds_read set1 = second half of tile t
    //   wait own LDS reads + own LDS-DMA, barrier     -> tile t+1 has landed, nobody reads buf[cur] any more
    //   issue LDS-DMA of tile t+2 into buf[cur]
    //   phase B: MFMA(set1)  ||  ds_read set0 = first half of tile t+1 (from buf[cur^1])
    // so fragment reads and the DMA always run under MFMAs.  "tile t+1 / t+2" continue into the next
    // work item (every item has at least two K steps: checked on the host), whose first tiles are
    // therefore already in flight while this item's epilogue stores drain.
    bf16x8 fa0[MI], fb0[NJ], fa1[MI], fb1[NJ];
    Ops::stage(smem, 0, (uint32_t)it.kbeg * kstrideA, (uint32_t)it.kbeg * kstrideB, w, it.rA, it.rB, offA, offB);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Ops::stage(smem, 1, (uint32_t)(it.kbeg + BK) * kstrideA, (uint32_t)(it.kbeg + BK) * kstrideB, w, it.rA, it.rB, offA, offB);
    Ops::read_frags(smem, 0, 0, wm, wn, lane, fa0, fb0);
    int cur = 0;
    for (int round = 0;; ++round) {
        const int Lnext = item_of(b, round + 1, G, total_items);
        WorkItem nxt = it;
        if (Lnext >= 0) nxt = decode_item<Cfg, TA, TB>(p, Lnext, ntm, ntn);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nsteps = (it.kend - it.kbeg + BK - 1) / BK;
        for (int t = 0; t < nsteps; ++t) {
            // phase A
            Ops::read_frags(smem, cur, 1, wm, wn, lane, fa1, fb1);
            Ops::template mfma_block<NATURAL>(fa0, fb0, acc);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            // tile t+2 -> buf[cur]
            if (t + 2 < nsteps) {
                const int k2 = it.kbeg + (t + 2) * BK;
                Ops::stage(smem, cur, (uint32_t)k2 * kstrideA, (uint32_t)k2 * kstrideB, w, it.rA, it.rB, offA, offB);
            } else if (Lnext >= 0) {
                const int kn = nxt.kbeg + (t + 2 - nsteps) * BK;    // first or second tile of the next item
                Ops::stage(smem, cur, (uint32_t)kn * kstrideA, (uint32_t)kn * kstrideB, w, nxt.rA, nxt.rB, offA, offB);
            }
            // phase B
            if (t + 1 < nsteps || Lnext >= 0) Ops::read_frags(smem, cur ^ 1, 0, wm, wn, lane, fa0, fb0);
            Ops::template mfma_block<NATURAL>(fa1, fb1, acc);
            cur ^= 1;
        }
        epilogue<Cfg, EPI>(p, it.m0, it.n0, wm, wn, lane, acc);
        if (Lnext < 0) break;
        it = nxt;
    }
}

namespace {

int g_force_tile = 0;    // 0 = heuristic, 128 or 256 = forced (paced_gemm_set_tile, for benchmarking)

int cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

template <class Cfg, bool TA, bool TB, int EPI>
int launch_cfg(GemmParams p, int splits, hipStream_t stream) {
    const int ntm = (p.M + Cfg::BM - 1) / Cfg::BM, ntn = (p.N + Cfg::BN - 1) / Cfg::BN;
    const int ktiles = (p.K + BK - 1) / BK;
    const int wg_per_cu = Cfg::NBUF == 2 ? 1 : 3;
    const int slots = cu_count() * wg_per_cu;
    if (splits <= 0) {          // auto: fill the persistent grid once, keep >= 4 K steps per slice
        splits = 1;
        if (EPI == EPI_ATOMIC_F32 && ntm * ntn < slots) splits = slots / (ntm * ntn);
        if (splits > ktiles / 4) splits = ktiles / 4 > 0 ? ktiles / 4 : 1;
    }
    if (splits > ktiles) splits = ktiles;
    const int tiles_per_split = (ktiles + splits - 1) / splits;
    p.k_per_split = tiles_per_split * BK;
    splits = (ktiles + tiles_per_split - 1) / tiles_per_split;
    if (Cfg::NBUF == 2 && (ktiles < 2 || (ktiles % tiles_per_split) == 1)) {
        paced_set_error("internal: the double-buffered tile needs at least two K steps per work item");
        return PACED_ERR_ARG;
    }
    const int total = ntm * ntn * splits;
    const int grid = total < slots ? total : slots;
    constexpr int smem = Cfg::NBUF * (Cfg::BM + Cfg::BN) * BK * 2;
    if (smem > 65536) {
        static bool attr_set = false;     // one flag per template instantiation
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<Cfg, TA, TB, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) { paced_set_error("hipFuncSetAttribute(%d bytes LDS) failed: %s", smem, hipGetErrorString(e)); return PACED_ERR_HIP; }
            attr_set = true;
        }
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<Cfg, TA, TB, EPI>), dim3(grid), dim3(Cfg::WGM * Cfg::WGN * 64), smem, stream, p, ntm, ntn, total);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

template <bool TA, bool TB, int EPI>
int launch(const GemmParams& p, int splits, hipStream_t stream) {
    bool large;
    if (g_force_tile == 128) large = false;
    else if (g_force_tile == 256) large = true;
    else {
        // the large tile pays off once there is about one tile per CU (split slices count for atomics)
        const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        large = p.M >= 256 && p.N >= 256 && (t256 >= 200 || (EPI == EPI_ATOMIC_F32 && t256 >= 24));
    }
    if (large && p.K >= 2 * BK) {
        // every K slice must hold at least two K steps: settle the split here
        const int ktiles = (p.K + BK - 1) / BK;
        int sp = splits;
        if (sp <= 0) {
            const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
            const int slots = cu_count();
            sp = (EPI == EPI_ATOMIC_F32 && tiles < slots) ? slots / tiles : 1;
            if (sp > ktiles / 4) sp = ktiles / 4 > 0 ? ktiles / 4 : 1;
        }
        if (sp > ktiles / 2) sp = ktiles / 2;
        while (sp > 1) {                                   // avoid a 1-step tail slice
            const int per = (ktiles + sp - 1) / sp;
            if (ktiles % per != 1) break;
            --sp;
        }
        return launch_cfg<CfgLarge, TA, TB, EPI>(p, sp, stream);
    }
    return launch_cfg<CfgSmall, TA, TB, EPI>(p, splits, stream);
}

template <bool TA, bool TB>
int dispatch_epi(const GemmParams& p, int epi, int splits, hipStream_t s) {
    switch (epi) {
        case EPI_BF16: return launch<TA, TB, EPI_BF16>(p, splits, s);
        case EPI_RES_DROP: return launch<TA, TB, EPI_RES_DROP>(p, splits, s);
        case EPI_RELU_DROP: return launch<TA, TB, EPI_RELU_DROP>(p, splits, s);
        case EPI_MASK: return launch<TA, TB, EPI_MASK>(p, splits, s);
        case EPI_F32: return launch<TA, TB, EPI_F32>(p, splits, s);
        case EPI_ATOMIC_F32: return launch<TA, TB, EPI_ATOMIC_F32>(p, splits, s);
        case EPI_TOPK_DENSE: if (!TA && !TB) return launch<false, false, EPI_TOPK_DENSE>(p, splits, s); break;
        case EPI_TOPK_FILTER: if (!TA && !TB) return launch<false, false, EPI_TOPK_FILTER>(p, splits, s); break;
    }
    return PACED_ERR_ARG;
}

}  // namespace

extern "C" int paced_gemm_set_tile(int tile) {
    PACED_CHECK_ARG(tile == 0 || tile == 128 || tile == 256, "paced_gemm_set_tile: tile must be 0 (auto), 128 or 256");
    g_force_tile = tile;
    return PACED_OK;
}

int paced_gemm_launch(const GemmParams& p, bool trans_a, bool trans_b, int epilogue, int splits, hipStream_t s) {
    if (!trans_a && !trans_b) return dispatch_epi<false, false>(p, epilogue, splits, s);
    if (!trans_a && trans_b) return dispatch_epi<false, true>(p, epilogue, splits, s);
    return dispatch_epi<true, true>(p, epilogue, splits, s);
}

extern "C" int paced_gemm_bf16(const void* A, const void* B, void* C, const void* aux,
                               int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
                               int trans_a, int trans_b, int epilogue, float alpha, float mask_scale,
                               float drop_p, uint32_t drop_seed, int splits, void* stream) {
    PACED_CHECK_ARG(A && B && C, "paced_gemm_bf16: null operand");
    PACED_CHECK_ARG(M > 0 && N > 0 && K > 0, "paced_gemm_bf16: M, N, K must be positive (got %d %d %d)", M, N, K);
    PACED_CHECK_ARG(epilogue >= 0 && epilogue < EPI_COUNT, "paced_gemm_bf16: unknown epilogue %d", epilogue);
    PACED_CHECK_ARG(!(trans_a && !trans_b), "paced_gemm_bf16: (trans_a=1, trans_b=0) is not built");
    PACED_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0, "paced_gemm_bf16: N and ldc must be multiples of 4");
    PACED_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0, "paced_gemm_bf16: lda/ldb must be multiples of 8 (16-byte rows)");
    PACED_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0,
                    "paced_gemm_bf16: operands must be 16-byte aligned");
    if (!trans_a) PACED_CHECK_ARG(K % 64 == 0, "paced_gemm_bf16: K must be a multiple of 64 for a K-contiguous A");
    if (!trans_b) PACED_CHECK_ARG(K % 64 == 0, "paced_gemm_bf16: K must be a multiple of 64 for a K-contiguous B");
    if (trans_a) PACED_CHECK_ARG((size_t)K * lda * 2 < 0xFFFFFFF0ull && M % 8 == 0, "paced_gemm_bf16: transposed A too large for 32-bit offsets or M%%8");
    if (trans_b) PACED_CHECK_ARG((size_t)K * ldb * 2 < 0xFFFFFFF0ull && N % 8 == 0, "paced_gemm_bf16: transposed B too large for 32-bit offsets or N%%8");
    if (!trans_a) PACED_CHECK_ARG((size_t)128 * lda * 2 < 0x7FFFFFFFull, "paced_gemm_bf16: lda too large");
    if (!trans_b) PACED_CHECK_ARG((size_t)128 * ldb * 2 < 0x7FFFFFFFull, "paced_gemm_bf16: ldb too large");
    if (epilogue == EPI_RES_DROP || epilogue == EPI_MASK)
        PACED_CHECK_ARG(aux && ldaux % 4 == 0 && ((uintptr_t)aux & 7) == 0, "paced_gemm_bf16: epilogue %d needs an 8-byte aligned aux with ldaux%%4==0", epilogue);
    PACED_CHECK_ARG(splits >= 0, "paced_gemm_bf16: splits must be >= 0 (0 = choose automatically)");
    PACED_CHECK_ARG(splits <= 1 || epilogue == EPI_ATOMIC_F32, "paced_gemm_bf16: split contraction needs the atomic epilogue");
    if (epilogue != EPI_ATOMIC_F32) splits = 1;

    GemmParams p;
    p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C; p.aux = (const bf16*)aux;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
    p.k_per_split = K;
    p.alpha = alpha; p.mask_scale = mask_scale;
    p.drop = make_drop(drop_p, drop_seed);
    p.tk_cand = nullptr; p.tk_cnt = nullptr; p.tk_thr = nullptr; p.tk_cap = 0; p.tk_row_base = 0;
    return paced_gemm_launch(p, trans_a != 0, trans_b != 0, epilogue, splits, (hipStream_t)stream);
}

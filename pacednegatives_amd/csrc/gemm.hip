// bf16 MFMA GEMM for gfx950 with fused epilogues: the contraction engine of the T5 blocks.
//
//   C[M,N] (+)= op(A) . op(B)^T-style product, fp32 accumulation, where
//     trans_a = 0 : A is stored [M][K] (K contiguous)      trans_a = 1 : A is stored [K][M]
//     trans_b = 0 : B is stored [N][K] (K contiguous)      trans_b = 1 : B is stored [K][N]
//   so (0,0) is y = x W^T (forward), (0,1) is dx = dy W (input gradient) and (1,1) is
//   dW = dy^T x (weight gradient, contraction over tokens, split over workgroups, fp32 atomics).
//
// Structure: persistent grid over (output tile, K slice) work items; K step 64.  Operand tiles go
// HBM -> LDS directly with buffer_load_dwordx4 ... lds (16 B per lane, 1 KiB per wave instruction); the
// LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and
// again on the fragment read.  K-contiguous tiles are read with ds_read_b128; K-strided tiles
// (transposed operands) are read with ds_read_b64_tr_b16.  Out-of-range rows rely on the buffer
// descriptor's range check (loads beyond num_records return zero), so ragged M/N and a ragged
// token contraction need no host-side padding.  Two tile configurations: see below.
#include "gemm.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BK = 64;

// Tile configurations.
//   small:  128x128 output tile, 4 waves (2x2, 64x64 per wave), one LDS buffer, two barriers per K step;
//           three workgroups per CU hide the staging latency.  Small outputs (decoder, ragged shapes).
//   tiny:   64x64 output tile, otherwise as small: outputs too small to give every CU a 128x128 tile
//           (the decoder's 2-token sequences).
//   phased: 256x256 output tile, 8 waves (2x4), one workgroup per CU, all 160 KiB of LDS as a ring of
//           half-tiles; see the kernel below.  moff/noff give the tile-relative origin of a wave's
//           i-th / j-th 16x16 accumulator block.
}  // namespace
struct CfgSmall {
    static constexpr int BM = 128, BN = 128, WGM = 2, WGN = 2, MI = 4, NJ = 4;
    static __device__ __forceinline__ int moff(int wm, int i) { return wm * 64 + i * 16; }
    static __device__ __forceinline__ int noff(int wn, int j) { return wn * 64 + j * 16; }
};
struct CfgTiny {
    static constexpr int BM = 64, BN = 64, WGM = 2, WGN = 2, MI = 2, NJ = 2;
    static __device__ __forceinline__ int moff(int wm, int i) { return wm * 32 + i * 16; }
    static __device__ __forceinline__ int noff(int wn, int j) { return wn * 32 + j * 16; }
};
struct CfgPhased {
    static constexpr int BM = 256, BN = 256, WGM = 2, WGN = 4, MI = 8, NJ = 4;
    static __device__ __forceinline__ int moff(int wm, int i) { return (i >> 2) * 128 + wm * 64 + (i & 3) * 16; }
    static __device__ __forceinline__ int noff(int wn, int j) { return wn * 64 + j * 16; }   // B halves interleave 32-column blocks
};
namespace {

// Per-lane source offset (bytes, relative to the tile's first row / column) of the 16-byte chunk
// that lands at lane-linear LDS position g (16-byte units) of a tile image.
//   K-contiguous operand : image [ROWS][64] bf16, 128-byte rows; chunk c of row r sits at c ^ ((r>>1)&7).
//   K-strided operand    : image [64][ROWS] bf16; the 32-byte unit u of k-row k sits at u ^ h(k),
//                          h(k) = (k&3) | ((k>>3)&1)<<2, masked to the units a k-row has
//                          (conflict-free ds_read_b64_tr_b16 from 128 rows up).
// With SPLIT (the phased kernel's B halves) image row r of a half is tile row (r>>5)*64 + (r&31): a
// half holds one 32-row block out of every 64 rows, so that a wave's two 32-row slices (one per half)
// are 64 CONSECUTIVE tile rows.
template <bool TRANS, int ROWS, bool SPLIT = false>
__device__ __forceinline__ uint32_t stage_offset(int g, int ld, long long heads = 0) {
    if (!TRANS) {
        const int r = g >> 3, cs = g & 7;
        const int c = cs ^ ((r >> 1) & 7);
        const int rs = SPLIT ? ((r >> 5) * 64 + (r & 31)) : r;
        return (uint32_t)(rs * ld + c * 8) * 2u;
    } else {
        constexpr int CPR = ROWS / 8;                 // 16-byte chunks per k-row
        const int k = g / CPR, cs = g % CPR;
        const int hk = ((k & 3) | (((k >> 3) & 1) << 2)) & (ROWS / 16 - 1);
        const int c = (((cs >> 1) ^ hk) << 1) | (cs & 1);
        const int csrc = SPLIT ? ((c >> 2) * 8 + (c & 3)) : c;
        if (heads) return (uint32_t)(((csrc >> 3) * heads + k * 64 + (csrc & 7) * 8) * 2);      // head-major: 8 chunks per head
        return (uint32_t)(k * ld + csrc * 8) * 2u;
    }
}

// 16 rows x 32 k fragment (one v_mfma_f32_16x16x32_bf16 operand) for rows rb..rb+15, k-step kk.
// With RAW_ASM the transposed reads are issued as inline assembly: the compiler then neither waits
// for them (the caller must s_waitcnt lgkmcnt before the first use) nor drains every LDS-DMA in flight
// ahead of them, which it does for the builtin (it cannot tell that the DMA targets another slot).
template <bool TRANS, int ROWS, bool RAW_ASM = false>
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
        const int hk = (q | ((g & 1) << 2)) & (ROWS / 16 - 1);
        const int k0 = kk * 32 + 8 * g + q;
        const int addr0 = k0 * RB + ((u ^ hk) << 5) + p * 8;
        bf16x4 lo, hi;
        if (RAW_ASM) {
            const uint32_t a = (uint32_t)(uintptr_t)LDS_PTR(tile + addr0);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(4 * RB));
        } else {
            lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + addr0));
            hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + addr0 + 4 * RB));
        }
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// output matrix of batch member b (fp32 for the fp32 / atomic epilogues, bf16 otherwise)
template <int EPI>
__device__ __forceinline__ void* batch_c(const GemmParams& p, int b) {
    constexpr int ES = (EPI == EPI_F32 || EPI == EPI_ATOMIC_F32) ? 4 : 2;
    return reinterpret_cast<char*>(p.C) + (long long)b * p.strideC * ES;
}

template <int EPI>
__device__ __forceinline__ void store_quad(const GemmParams& p, void* Cb, int m, int n, const f32x4& a) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = a[r] * p.alpha;
    const size_t ci = (EPI == EPI_BF16 && p.c_heads) ? (size_t)(n >> 6) * p.c_heads + (size_t)m * 64 + (n & 63) : (size_t)m * p.ldc + n;
    const uint64_t di = (uint64_t)m * (uint64_t)p.N + n;   // dropout counter: logical (row, col)
    if (EPI == EPI_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Cb) + ci) = f32x4{v[0], v[1], v[2], v[3]};
        return;
    }
    float f[4] = {1.f, 1.f, 1.f, 1.f};
    if (EPI == EPI_RES_DROP || EPI == EPI_RELU_DROP) drop_factor4(p.drop, di, f);     // N % 4 == 0: di is a multiple of 4
    if (EPI == EPI_RES_DROP || EPI == EPI_MASK) {
        const bf16x4 x = *reinterpret_cast<const bf16x4*>(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (EPI == EPI_RES_DROP) v[r] = bf2f(x[r]) + v[r] * f[r];
            else v[r] = bf2f(x[r]) > 0.f ? v[r] * p.mask_scale : 0.f;
        }
    } else if (EPI == EPI_MASK_BITS) {
        // one bit per element, 8 columns per byte, ldaux bytes per row; this quad is the low or high nibble
        const unsigned int byte = reinterpret_cast<const unsigned char*>(p.aux)[(size_t)m * p.ldaux + (n >> 3)];
        const unsigned int nib = byte >> (n & 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ((nib >> r) & 1u) ? v[r] * p.mask_scale : 0.f;
    } else if (EPI == EPI_RELU_DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f) * f[r];
    }
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
    if (EPI == EPI_RELU_DROP && p.aux) {
        // sign bits of the 8 columns n&~7 .. +7 of row m: this lane's nibble and that of the lane 16 further on
        // (same row, next quad); N % 8 == 0, so the partner of an in-range even quad is in range too
        unsigned int nib = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) nib |= (bf2f(o[r]) > 0.f ? 1u : 0u) << r;
        const unsigned int other = (unsigned int)__shfl_xor((int)nib, 16, 64);
        if ((n & 4) == 0)
            const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.aux))[(size_t)m * p.ldaux + (n >> 3)] = (unsigned char)(nib | (other << 4));
    }
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(Cb) + ci) = o;
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

    // multiply the K step held in LDS buffer `buf` into the accumulators
    // RAW: transposed fragments are read by inline assembly (the builtin makes the compiler drain every LDS-DMA in
    // flight first), followed by one wait for all LDS reads
    template <bool NATURAL, bool RAW = false>
    static __device__ __forceinline__ void multiply(const char* smem, int buf, int wm, int wn, int lane, f32x4 (&acc)[MI][NJ]) {
        const char* lA = smem + buf * BUF_BYTES;
        const char* lB = lA + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MI], fb[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = read_frag<TB, BN, RAW>(lB, wn * TN + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = read_frag<TA, BM, RAW>(lA, wm * TM + i * 16, kk, lane);
            if (RAW && (TA || TB)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
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
    int m0, n0, kbeg, kend, b;
    __amdgpu_buffer_rsrc_t rA, rB;
};

template <class Cfg, bool TA, bool TB>
__device__ __forceinline__ WorkItem decode_item(const GemmParams& p, int L, int ntm, int ntn) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN;
    WorkItem it;
    const int ntiles = ntm * ntn;
    const int nsplit = (p.K + p.k_per_split - 1) / p.k_per_split;
    const int per_batch = ntiles * nsplit;
    it.b = L / per_batch;
    const int r = L - it.b * per_batch;
    const int split = r / ntiles, t = r - split * ntiles;
    // neighbouring items share an operand panel: walk the SHORTER tile dimension fastest
    int tm, tn;
    if (ntm >= ntn) { tm = t / ntn; tn = t - tm * ntn; } else { tn = t / ntm; tm = t - tn * ntm; }
    it.m0 = tm * BM; it.n0 = tn * BN;
    it.kbeg = split * p.k_per_split;
    it.kend = min(p.K, it.kbeg + p.k_per_split);
    // Buffer descriptors: base at the tile's first row (or column), range = exactly the valid bytes,
    // so rows >= M/N and contraction rows >= K read as zero.
    const int rowsA = min(BM, p.M - it.m0), rowsB = min(BN, p.N - it.n0);
    const bf16* baseA = p.A + (long long)it.b * p.strideA + (TA ? (p.a_heads ? (size_t)(it.m0 >> 6) * p.a_heads : (size_t)it.m0) : (size_t)it.m0 * p.lda);
    const bf16* baseB = p.B + (long long)it.b * p.strideB + (TB ? (p.b_heads ? (size_t)(it.n0 >> 6) * p.b_heads : (size_t)it.n0) : (size_t)it.n0 * p.ldb);
    // (a transposed A whose M is not a multiple of 8: the last 16-byte piece of a contraction row straddles M; it is kept
    // in range, lda >= M rounded up to 8 guarantees the bytes exist, and the output rows >= M it feeds are never stored)
    // Head-major operands: the range runs to the end of the operand's last head.  What a ragged tile reads past its own
    // rows is then the next head's data (finite) instead of zeros: output rows past M are never stored, and a contraction
    // row past K meets the other operand's zeros (only one operand of a product may be head-major and transposed).
    const uint32_t recA = TA ? (p.a_heads ? (uint32_t)(((size_t)((p.M + 63) / 64 - 1 - (it.m0 >> 6)) * p.a_heads + (size_t)p.K * 64) * 2)
                                          : (uint32_t)(((size_t)(p.K - 1) * p.lda + ((rowsA + 7) & ~7)) * 2))
                        : p.a_heads ? (uint32_t)(((size_t)(p.K / 64 - 1) * p.a_heads + (size_t)rowsA * 64) * 2)
                        : (uint32_t)(((size_t)(rowsA - 1) * p.lda + p.K) * 2);
    const uint32_t recB = TB ? (p.b_heads ? (uint32_t)(((size_t)((p.N + 63) / 64 - 1 - (it.n0 >> 6)) * p.b_heads + (size_t)p.K * 64) * 2)
                                          : (uint32_t)(((size_t)(p.K - 1) * p.ldb + rowsB) * 2))
                        : (uint32_t)(((size_t)(rowsB - 1) * p.ldb + p.K) * 2);
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
__device__ __forceinline__ void epilogue(const GemmParams& p, void* Cb, int m0, int n0, int wm, int wn, int lane,
                                         f32x4 (&acc)[Cfg::MI][Cfg::NJ]) {
    constexpr int MI = Cfg::MI, NJ = Cfg::NJ;
    if (EPI == EPI_TOPK_DENSE) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + Cfg::noff(wn, j) + (lane & 15);
            if (n >= p.N) continue;
            unsigned long long* cand = p.tk_cand + (size_t)n * p.tk_cap;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + Cfg::moff(wm, i) + (lane >> 4) * 4 + r;
                    if (m < p.M) cand[m] = topk_key(acc[i][j][r], p.tk_row_base + (unsigned int)m);
                }
        }
        return;
    }
    if (EPI == EPI_TOPK_FILTER) {
        // The thresholds and the slot counters go through inline-asm statements with their own waits: a
        // register-destination load or returning atomic the compiler can see makes it put `s_waitcnt vmcnt(0)`
        // ahead of every LDS read of the kernel, which drains the K loop's LDS-DMA ring once per K step.
        // A lane first marks which of its MI x 4 scores per column pass (one bit each), then reserves all the slots
        // of a column with ONE returning atomic; the NJ atomics of a lane are in flight together and waited for
        // once.  (An atomic per passing score, each waited for, cost a memory round trip per score: in the early
        // chunks of a sweep, where ~0.2 % of the scores pass, that was as long as the item's whole K loop.)
        static_assert(MI * 4 <= 32, "one pass bit per score and column");
        float thr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float* src = p.tk_thr + min(n0 + Cfg::noff(wn, j) + (lane & 15), p.N - 1);
            asm volatile("global_load_dword %0, %1, off" : "=v"(thr[j]) : "v"(src) : "memory");
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("s_waitcnt vmcnt(0) ; hidden-load wait" : "+v"(thr[j]) :: "memory");
        unsigned int pass[NJ], pos[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + Cfg::noff(wn, j) + (lane & 15);
            unsigned int bits = 0;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + Cfg::moff(wm, i) + (lane >> 4) * 4 + r;
                    bits |= (m < p.M && acc[i][j][r] >= thr[j] ? 1u : 0u) << (i * 4 + r);
                }
            pass[j] = n < p.N ? bits : 0u;
            pos[j] = 0;
            if (pass[j]) {
                const unsigned int count = __popc(pass[j]);
                unsigned int* slot = p.tk_cnt + n;
                asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=&v"(pos[j]) : "v"(slot), "v"(count) : "memory");
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("s_waitcnt vmcnt(0) ; hidden-load wait" : "+v"(pos[j]) :: "memory");
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (__ballot(pass[j] != 0) == 0) continue;               // nobody in the wave has a candidate in this column group
            const int n = n0 + Cfg::noff(wn, j) + (lane & 15);
            unsigned long long* cand = p.tk_cand + (size_t)min(n, p.N - 1) * p.tk_cap;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (__ballot(((pass[j] >> (i * 4)) & 15u) != 0) == 0) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int bit = i * 4 + r;
                    if ((pass[j] >> bit) & 1u) {
                        const unsigned int at = pos[j] + __popc(pass[j] & ((1u << bit) - 1u));
                        const int m = m0 + Cfg::moff(wm, i) + (lane >> 4) * 4 + r;
                        if (at < (unsigned int)p.tk_cap) cand[at] = topk_key(acc[i][j][r], p.tk_row_base + (unsigned int)m);
                    }
                }
            }
        }
        return;
    }
    if (EPI == EPI_ATOMIC_F32) {
        float* C = reinterpret_cast<float*>(Cb);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + Cfg::noff(wn, j) + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + Cfg::moff(wm, i) + (lane >> 4) * 4 + r;
                    if (m < p.M && n < p.N) atomicAdd(C + (size_t)m * p.ldc + n, acc[i][j][r] * p.alpha);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + Cfg::moff(wm, i) + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + Cfg::noff(wn, j) + (lane >> 4) * 4;
            if (n >= p.N) continue;          // N is a multiple of 4 (checked on the host)
            store_quad<EPI>(p, Cb, m, n, acc[i][j]);
        }
    }
}

// Small-tile persistent kernel: gridDim.x workgroups walk `total_items` work items round by round.
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
    for (int i = 0; i < Ops::IA; ++i) offA[i] = stage_offset<TA, BM>((i * NW + w) * 64 + lane, p.lda, p.a_heads);
#pragma unroll
    for (int i = 0; i < Ops::IB; ++i) offB[i] = stage_offset<TB, BN>((i * NW + w) * 64 + lane, p.ldb, p.b_heads);
    const uint32_t kstrideA = TA ? (uint32_t)p.lda * 2u : p.a_heads ? (uint32_t)(p.a_heads / 32) : 2u;   // bytes per contraction index
    const uint32_t kstrideB = TB ? (uint32_t)p.ldb * 2u : 2u;

    int L = item_of(b, 0, G, total_items);
    if (L < 0) return;
    WorkItem it = decode_item<Cfg, TA, TB>(p, L, ntm, ntn);
    f32x4 acc[MI][NJ];

    {
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
            epilogue<Cfg, EPI>(p, batch_c<EPI>(p, it.b), it.m0, it.n0, wm, wn, lane, acc);
            L = item_of(b, ++round, G, total_items);
            if (L >= 0) it = decode_item<Cfg, TA, TB>(p, L, ntm, ntn);
        }
    }
}

// Transposed fragment reads are inline assembly here (see read_frag): wait for all LDS reads of the
// phase before its MFMAs, and keep the scheduler from hoisting anything above the wait.
#define PACED_WAIT_LDS_READS()                                \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)

// a raw barrier that the instruction scheduler may not move anything across (MFMAs included)
#define PACED_PINNED_BARRIER()                 \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

// Ring variant of the small-tile kernel for outputs with no more tiles than a few per CU (the decoder's two-token
// rows): there occupancy cannot hide the staging latency, and one buffer with a full wait per K step makes every
// K step cost a memory round trip (≈1 us; 48 of them in a row for K = 3072).  Here a workgroup owns ONE work item
// and NS LDS buffers; NS - 1 K steps are in flight, the wait before step t leaves the NS - 2 younger ones pending.
// Per K step: counted wait | raw barrier | issue step t + NS - 1 into the buffer step t - 1 was read from (every
// wave has passed this barrier, so its fragment reads of step t - 1 have been consumed) | multiply step t.
// No global load with a register destination precedes the loop (the epilogue comes after it and the kernel does
// not loop over items), so the compiler adds no wait of its own ahead of the fragment reads.
template <class Cfg, bool TA, bool TB, int EPI, int NS>
__global__ void __launch_bounds__(Cfg::WGM * Cfg::WGN * 64) gemm_ring_kernel(GemmParams p, int ntm, int ntn, int total_items) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NW = Cfg::WGM * Cfg::WGN;
    constexpr int MI = BM / Cfg::WGM / 16, NJ = BN / Cfg::WGN / 16;
    constexpr bool NATURAL = (EPI == EPI_ATOMIC_F32 || EPI == EPI_TOPK_DENSE || EPI == EPI_TOPK_FILTER);
    using Ops = TileOps<Cfg, TA, TB>;
    constexpr int PER = Ops::IA + Ops::IB;           // LDS-DMA instructions per wave and K step
    static_assert(NS >= 3 && (NS - 2) * PER <= 63, "the pending count must fit the vmcnt field");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / Cfg::WGN, wn = w % Cfg::WGN;

    uint32_t offA[Ops::IA], offB[Ops::IB];
#pragma unroll
    for (int i = 0; i < Ops::IA; ++i) offA[i] = stage_offset<TA, BM>((i * NW + w) * 64 + lane, p.lda, p.a_heads);
#pragma unroll
    for (int i = 0; i < Ops::IB; ++i) offB[i] = stage_offset<TB, BN>((i * NW + w) * 64 + lane, p.ldb, p.b_heads);
    const uint32_t kstrideA = TA ? (uint32_t)p.lda * 2u : p.a_heads ? (uint32_t)(p.a_heads / 32) : 2u;   // bytes per contraction index
    const uint32_t kstrideB = TB ? (uint32_t)p.ldb * 2u : 2u;

    const WorkItem it = decode_item<Cfg, TA, TB>(p, xcd_remap(blockIdx.x, total_items), ntm, ntn);
    const int nsteps = (it.kend - it.kbeg + BK - 1) / BK;
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nsteps)
            Ops::stage(smem, s, (uint32_t)(it.kbeg + s * BK) * kstrideA, (uint32_t)(it.kbeg + s * BK) * kstrideB, w, it.rA, it.rB, offA, offB);
    int buf = 0, nbuf = NS - 1;                      // buffer of step t, buffer of step t + NS - 1
    for (int t = 0; t < nsteps; ++t) {
        if (t + NS - 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PACED_PINNED_BARRIER();
        if (t + NS - 1 < nsteps) {
            const int k0 = it.kbeg + (t + NS - 1) * BK;
            Ops::stage(smem, nbuf, (uint32_t)k0 * kstrideA, (uint32_t)k0 * kstrideB, w, it.rA, it.rB, offA, offB);
        }
        Ops::template multiply<NATURAL, true>(smem, buf, wm, wn, lane, acc);
        buf = buf + 1 == NS ? 0 : buf + 1;
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    }
    epilogue<Cfg, EPI>(p, batch_c<EPI>(p, it.b), it.m0, it.n0, wm, wn, lane, acc);
}

// ---------------------------------------------------------------------------------------------
// Phased 256x256 kernel.
//
// LDS: a ring of ten 16 KiB half-tile slots (all 160 KiB).  The operand stream is the sequence of
// half-tiles  B half 0, A half 0, B half 1, A half 1  of K step 0, then of K step 1, ...; stream item s
// lives in slot s mod 10, and a half is the 128-row tile image of the small configuration (same
// source swizzle, same fragment reads).  K step T runs as four phases g = 4T .. 4T+3; every phase is
//     ds_read the fragments it needs | issue the LDS-DMA of stream item g+8 | s_waitcnt vmcnt(12) |
//     s_barrier | 16 MFMAs (one 64x32 quadrant x 64 deep) | s_barrier
// (A half h is tile rows 128h..128h+127 and a wave owns 64 of them; B half h is the h-th 32-column
// block of every 64 columns and a wave owns one such block per half, i.e. 64 consecutive columns.)
//   phase 1: read B half 0 (4) and A half 0 (8);  quadrant (0,0)
//   phase 2: read B half 1 (4);                   quadrant (0,1)
//   phase 3: read A half 1 (8);                   quadrant (1,1)
//   phase 4: nothing to read;                     quadrant (1,0)
// so six half-tiles (96 KiB per workgroup) stay in flight across the raw barriers and a half-tile has
// six phases (1.5 K steps) to land.  Ordering rules this relies on (LDS-DMA data is ordered for a
// ds_read only by the issuing wave's vmcnt wait followed by a barrier the reader has passed; the two
// wave groups run one barrier apart):
//   * item s is read in phase s-1 at the earliest; the wait of phase g retires items <= g+2, so an
//     item is read no earlier than the phase AFTER the wait that retires it;
//   * item s is issued in phase s-8 into the slot of item s-10, whose last read was in phase s-10 at
//     the latest: a slot is overwritten at least two phases after its last read.
// The stream runs across work items (persistent grid), so the next item's first K steps are in flight
// during an item's last phases and its epilogue.  Past the last item the stream issues the same DMAs
// through a zero-length descriptor (they write zeros nobody reads): the vmcnt arithmetic stays
// uniform and needs no tail case.
struct TileSrc {
    __amdgpu_buffer_rsrc_t rA, rB;
    uint32_t sA, sB;       // scalar byte offsets of the K step inside the descriptors
};

template <bool TA, bool TB>
struct StepStream {
    WorkItem wi;
    int k, round;
    bool live;
    __device__ __forceinline__ TileSrc take(const GemmParams& p, int b, int G, int ntm, int ntn, int total_items,
                                            uint32_t kstrideA, uint32_t kstrideB) {
        TileSrc ts;
        if (live) {
            ts.rA = wi.rA; ts.rB = wi.rB;
            ts.sA = (uint32_t)k * kstrideA; ts.sB = (uint32_t)k * kstrideB;
            k += BK;
            if (k >= wi.kend) {
                const int Ln = item_of(b, ++round, G, total_items);
                if (Ln >= 0) { wi = decode_item<CfgPhased, TA, TB>(p, Ln, ntm, ntn); k = wi.kbeg; }
                else live = false;
            }
        } else {
            ts.rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
            ts.rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0, 0x00020000);
            ts.sA = 0; ts.sB = 0;
        }
        return ts;
    }
};

#define PACED_STAGE_HALF(rsrc, dst, off, soff)                                                           \
    do {                                                                                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), LDS_PTR((dst)), 16, (off)[0], (soff), 0, 0);         \
        __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), LDS_PTR((dst) + 8192), 16, (off)[1], (soff), 0, 0);  \
    } while (0)

// Empty volatile asm that "rewrites" the eight accumulators of a quadrant: MFMAs are pure register
// operations, which instruction selection otherwise floats across the barriers into the load sections.
__device__ __forceinline__ void pin_quadrant(f32x4 (&acc)[8][4], int ha, int hb) {
    asm volatile("" : "+v"(acc[ha * 4 + 0][hb * 2 + 0]), "+v"(acc[ha * 4 + 0][hb * 2 + 1]),
                      "+v"(acc[ha * 4 + 1][hb * 2 + 0]), "+v"(acc[ha * 4 + 1][hb * 2 + 1]),
                      "+v"(acc[ha * 4 + 2][hb * 2 + 0]), "+v"(acc[ha * 4 + 2][hb * 2 + 1]),
                      "+v"(acc[ha * 4 + 3][hb * 2 + 0]), "+v"(acc[ha * 4 + 3][hb * 2 + 1]));
}

template <bool NATURAL>
__device__ __forceinline__ void quadrant_mfma(const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2], f32x4 (&acc)[8][4], int ha, int hb) {
    pin_quadrant(acc, ha, hb);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4& c = acc[ha * 4 + i][hb * 2 + j];
                if (NATURAL) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][kk], fb[j][kk], c, 0, 0, 0);
                else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], c, 0, 0, 0);
            }
    __builtin_amdgcn_s_setprio(0);
    pin_quadrant(acc, ha, hb);
}

__device__ __forceinline__ int ring_wrap(int x) { return x >= 10 ? x - 10 : x; }

// Epilogue of the phased kernel for bf16 outputs.  The MFMA layout gives a lane 4 consecutive columns
// of one row (8-byte stores, 32-byte runs per row); instead every 16-row block of the wave's 128 x 64
// result goes through 2 KiB of LDS as bf16 and comes back with 8 lanes covering a whole 128-byte row,
// so residual/mask loads and the stores move full lines, 16 bytes per lane.  The staging LDS is the
// wave's own 1 KiB pieces of the two ring slots that are dead at this point (the pieces the wave
// itself overwrites with its next two LDS-DMAs), so no barrier is involved.  Values are rounded to
// bf16 before the transposition: C = bf16(aux + bf16(dropout(acc))) for the residual form.
// The residual / mask operand is loaded by inline-asm statements: next to LDS-DMA the compiler answers any
// register-destination load it can see with `s_waitcnt vmcnt(0)` ahead of every LDS read of the kernel, the
// K loop's included, which drains the DMA ring once per K step.  Hidden loads are counted by hand instead.
// A lane consumes 16 pieces (one per 8-row half-block, n = 0..15), eight of them in flight at a time: loads 0-7 go
// out first, and load n + 8 is issued right after the store of half-block n into the registers that n has freed.
// FULL (the tile lies inside the matrix; every store is issued): when piece n is consumed the operations younger
// than its load are loads n+1 .. min(n+7, 15) and the stores issued since, n of them for n < 8 and 7 after that.
// Edge tiles clamp the load addresses into the matrix, predicate the stores and wait for each batch of eight whole.
// bf16 output rows of the phased kernel are stored non-temporally: an output (214-856 MB at the benchmark shape) is read
// again only by a later kernel and far exceeds the caches; stored the ordinary way it evicts the weight panels and
// activations that this launch and the launches of the other streams re-read (step 117.7 -> 115.5 ms, alternating in one job;
// -DGEMM_NT_STORE=0 restores ordinary stores).
#ifndef GEMM_NT_BITS
#define GEMM_NT_BITS 1
#endif
#ifndef GEMM_NT_AUX_LOAD
#define GEMM_NT_AUX_LOAD 1
#endif
#ifndef GEMM_NT_STORE
#define GEMM_NT_STORE 1
#endif
__device__ __forceinline__ void st_out(bf16x8* p, bf16x8 v) {
    if (GEMM_NT_STORE) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <class T>
__device__ __forceinline__ void wait_hidden_load(int n, T& x) {
#define PACED_WAIT_CASE(N, CNT) if (n == N) asm volatile("s_waitcnt vmcnt(" #CNT ") ; hidden-load wait" : "+v"(x) :: "memory");
    PACED_WAIT_CASE(0, 7) PACED_WAIT_CASE(1, 8) PACED_WAIT_CASE(2, 9) PACED_WAIT_CASE(3, 10)
    PACED_WAIT_CASE(4, 11) PACED_WAIT_CASE(5, 12) PACED_WAIT_CASE(6, 13) PACED_WAIT_CASE(7, 14)
    PACED_WAIT_CASE(8, 14) PACED_WAIT_CASE(9, 13) PACED_WAIT_CASE(10, 12) PACED_WAIT_CASE(11, 11)
    PACED_WAIT_CASE(12, 10) PACED_WAIT_CASE(13, 9) PACED_WAIT_CASE(14, 8) PACED_WAIT_CASE(15, 7)
#undef PACED_WAIT_CASE
}
template <class T>
__device__ __forceinline__ void wait_hidden_batch(T (&x)[8]) {
    asm volatile("s_waitcnt vmcnt(0) ; hidden-load wait" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) :: "memory");
}

template <int EPI, bool FULL, bool TILED>
__device__ __forceinline__ void epilogue_rows_impl(const GemmParams& p, void* Cb, int m0, int n0, int wr, int wc, int lane,
                                                   f32x4 (&acc)[8][4], char* pc0, char* pc1, char* pc2, char* pc3,
                                                   u32x4& tbits, unsigned char* tile_bits) {
    // TILED: the bit matrix is kept in tile order, 16 bytes per lane and tile (byte n = the lane's piece n), at tile_bits;
    // a reader's 16 bytes arrive in tbits (requested before the tile's K loop), a writer collects them there.
    constexpr bool LOADS = (EPI == EPI_RES_DROP || EPI == EPI_MASK || (EPI == EPI_MASK_BITS && !TILED));
    if (TILED && EPI == EPI_RELU_DROP) tbits = u32x4{0u, 0u, 0u, 0u};
    const int rr = lane >> 3, c = lane & 7;
    const int ncol = n0 + wc * 64 + c * 8;
    const bool col_ok = FULL || ncol < p.N;           // N is a multiple of 8 on this path
    uint32_t mbits[8];
    bf16x8 aux[8];
    auto load_piece = [&](int n) {
        const int m = m0 + CfgPhased::moff(wr, n >> 1) + (n & 1) * 8 + rr;
        const int mc = FULL ? m : min(m, p.M - 1);
        const int nc = col_ok ? ncol : 0;
        if (EPI == EPI_MASK_BITS) {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(p.aux) + (size_t)mc * p.ldaux + (nc >> 3);
            asm volatile("global_load_ubyte %0, %1, off" : "=v"(mbits[n & 7]) : "v"(src) : "memory");
        } else {
            const bf16* src = p.aux + (size_t)mc * p.ldaux + nc;
            // (the residual / mask operand is read once per launch: non-temporal, it need not stay in the L2s)
            if (GEMM_NT_AUX_LOAD) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(aux[n & 7]) : "v"(src) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(aux[n & 7]) : "v"(src) : "memory");
        }
    };
    auto wait_batch = [&]() {
        if (EPI == EPI_MASK_BITS) wait_hidden_batch(mbits);
        else wait_hidden_batch(aux);
    };
    if (LOADS) {
#pragma unroll
        for (int n = 0; n < 8; ++n) load_piece(n);
        if (!FULL) wait_batch();
    }
    const int mw = lane & 15, g = lane >> 4;
    const int wsw = ((mw & 7) << 1) | (mw >> 3);
    const int woff = (mw & 7) * 128;
    const int roff = rr * 128 + ((c ^ rr) << 4);
    const bool has_drop = (EPI == EPI_RES_DROP || EPI == EPI_RELU_DROP) && p.drop.thresh != 0;
    const float scale = (EPI == EPI_MASK || EPI == EPI_MASK_BITS) ? p.alpha * p.mask_scale : p.alpha;

    // 16-row block i as bf16 in the MFMA layout (4 consecutive columns of one row per lane and 16-column group)
    auto convert = [&](int i, bf16x4 (&o)[4]) {
        const uint64_t drow = (uint64_t)(m0 + CfgPhased::moff(wr, i) + mw) * (uint64_t)p.N + (uint64_t)(n0 + wc * 64 + g * 4);
        const uint32_t qrow = (uint32_t)(drow >> 2);          // quad number: the host sends outputs of more than 2^34 elements elsewhere
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[i][j] * scale;
            if (EPI == EPI_RELU_DROP) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
            if (has_drop) {
                float f[4];
                drop_factor4_on32(p.drop, qrow + j * 4, f);
                v *= f32x4{f[0], f[1], f[2], f[3]};
            }
            o[j] = __builtin_convertvector(v, bf16x4);
        }
    };
    // ... written to the wave's staging pieces (rows 0-7 to one, 8-15 to the other, 8-byte units swizzled)
    auto stage = [&](int i, const bf16x4 (&o)[4]) {
        char* pw = (mw < 8 ? ((i & 1) ? pc2 : pc0) : ((i & 1) ? pc3 : pc1)) + woff;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x4*>(pw + (((j * 4 + g) ^ wsw) << 3)) = o[j];
    };
    // ... and read back with 8 lanes covering a 128-byte row: lane (rr, c) gets columns 8c..8c+7 of rows rr and 8+rr
    auto finish = [&](int i, int q, bf16x8 t) {
        if (q) t = __builtin_shufflevector(t, t, 4, 5, 6, 7, 0, 1, 2, 3);      // rows 8..15 were stored with their 8-byte halves swapped
        const int n = i * 2 + q;
        if (LOADS && FULL) {
            if (EPI == EPI_MASK_BITS) wait_hidden_load(n, mbits[n & 7]);
            else wait_hidden_load(n, aux[n & 7]);
        }
        if (EPI == EPI_RES_DROP) {
            const bf16x8 x = aux[n & 7];
            const bf16x4 lo = __builtin_convertvector(__builtin_convertvector(__builtin_shufflevector(x, x, 0, 1, 2, 3), f32x4) +
                                                      __builtin_convertvector(__builtin_shufflevector(t, t, 0, 1, 2, 3), f32x4), bf16x4);
            const bf16x4 hi = __builtin_convertvector(__builtin_convertvector(__builtin_shufflevector(x, x, 4, 5, 6, 7), f32x4) +
                                                      __builtin_convertvector(__builtin_shufflevector(t, t, 4, 5, 6, 7), f32x4), bf16x4);
            t = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else if (EPI == EPI_MASK) {
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = bf2f(aux[n & 7][e]) > 0.f ? t[e] : (bf16)0.f;
        } else if (EPI == EPI_MASK_BITS) {
            // bit e of the byte keeps element e: every pair of elements is one register, masked by (sign-extended bit 2j,
            // sign-extended bit 2j + 1) in its halves; a masked element becomes +0, as a select would make it
            const uint32_t byte = TILED ? tbits[n >> 2] >> ((n & 3) * 8) : mbits[n & 7];
            u32x4 tw = __builtin_bit_cast(u32x4, t);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = (uint32_t)((int32_t)(byte << (31 - 2 * j)) >> 31);
                const uint32_t hi = (uint32_t)((int32_t)(byte << (30 - 2 * j)) >> 31);
                tw[j] &= (lo & 0xffffu) | (hi & 0xffff0000u);
            }
            t = __builtin_bit_cast(bf16x8, tw);
        }
        const int m = m0 + CfgPhased::moff(wr, i) + q * 8 + rr;
        const bool ok = FULL || (m < p.M && col_ok);
        if (EPI == EPI_RELU_DROP && p.aux && (TILED || ok)) {
            unsigned int byte = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) byte |= (bf2f(t[e]) > 0.f ? 1u : 0u) << e;
            if (TILED) tbits[n >> 2] |= byte << ((n & 3) * 8);
            else const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.aux))[(size_t)m * p.ldaux + (ncol >> 3)] = (unsigned char)byte;
        }
        if (EPI == EPI_BF16 && p.c_heads) {        // a wave's 64 columns are one head: rows of 128 bytes, 128 bytes apart
            if (ok) st_out(reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(Cb) + (size_t)(ncol >> 6) * p.c_heads + (size_t)m * 64 + (ncol & 63)), t);
        } else if (ok) st_out(reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(Cb) + (size_t)m * p.ldc + ncol), t);
        if (LOADS && n < 8) {
            if (FULL) load_piece(n + 8);
            else if (n == 7) {
#pragma unroll
                for (int k = 8; k < 16; ++k) load_piece(k);
                wait_batch();
            }
        }
    };

    // Pipelined over the eight blocks: the LDS reads of block i are in flight while block i+1 is converted and written
    // to the other pair of pieces; only then are the reads waited for.
    bf16x4 o[4];
    convert(0, o);
    stage(0, o);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bf16x8 t0 = *reinterpret_cast<const bf16x8*>(((i & 1) ? pc2 : pc0) + roff);
        const bf16x8 t1 = *reinterpret_cast<const bf16x8*>(((i & 1) ? pc3 : pc1) + roff);
        if (i + 1 < 8) {
            convert(i + 1, o);
            stage(i + 1, o);
        }
        finish(i, 0, t0);
        finish(i, 1, t1);
    }
    if (TILED && EPI == EPI_RELU_DROP) {          // read again by the backward pass only
        if (GEMM_NT_BITS) __builtin_nontemporal_store(tbits, reinterpret_cast<u32x4*>(tile_bits));
        else *reinterpret_cast<u32x4*>(tile_bits) = tbits;
    }
}

template <int EPI>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, void* Cb, int m0, int n0, int wr, int wc, int lane,
                                              f32x4 (&acc)[8][4], char* pc0, char* pc1, char* pc2, char* pc3,
                                              u32x4& tbits, unsigned char* tile_bits) {
    const bool full = m0 + 256 <= p.M && n0 + 256 <= p.N;
    if ((EPI == EPI_RELU_DROP || EPI == EPI_MASK_BITS) && tile_bits) {
        if (full) epilogue_rows_impl<EPI, true, true>(p, Cb, m0, n0, wr, wc, lane, acc, pc0, pc1, pc2, pc3, tbits, tile_bits);
        else epilogue_rows_impl<EPI, false, true>(p, Cb, m0, n0, wr, wc, lane, acc, pc0, pc1, pc2, pc3, tbits, tile_bits);
    } else if (full) epilogue_rows_impl<EPI, true, false>(p, Cb, m0, n0, wr, wc, lane, acc, pc0, pc1, pc2, pc3, tbits, tile_bits);
    else epilogue_rows_impl<EPI, false, false>(p, Cb, m0, n0, wr, wc, lane, acc, pc0, pc1, pc2, pc3, tbits, tile_bits);
}

template <bool TA, bool TB, int EPI>
__global__ void __launch_bounds__(512) gemm_phased_kernel(GemmParams p, int ntm, int ntn, int total_items) {
    constexpr bool NATURAL = (EPI == EPI_ATOMIC_F32 || EPI == EPI_TOPK_DENSE || EPI == EPI_TOPK_FILTER);
    constexpr int HALF = 16384;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int G = gridDim.x, b = blockIdx.x;

    uint32_t offA[2], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offA[i] = stage_offset<TA, 128>((i * 8 + w) * 64 + lane, p.lda, p.a_heads);
        offB[i] = stage_offset<TB, 128, true>((i * 8 + w) * 64 + lane, p.ldb, p.b_heads);
    }
    const uint32_t kstrideA = TA ? (uint32_t)p.lda * 2u : p.a_heads ? (uint32_t)(p.a_heads / 32) : 2u;   // bytes per contraction index
    const uint32_t kstrideB = TB ? (uint32_t)p.ldb * 2u : 2u;
    // bytes from half 0 to half 1 (128 rows; of a transposed head-major A: two heads on)
    const uint32_t halfA = TA ? (p.a_heads ? (uint32_t)(p.a_heads * 4) : 256u) : 256u * (uint32_t)p.lda;
    const uint32_t halfB = TB ? 64u : 64u * (uint32_t)p.ldb;          // B half 1 starts 32 rows after half 0
    char* const sw = smem + w * 1024;                                // this wave's slice of a half-tile image

    int L = item_of(b, 0, G, total_items);
    if (L < 0) return;
    StepStream<TA, TB> st;
    st.wi = decode_item<CfgPhased, TA, TB>(p, L, ntm, ntn);
    st.k = st.wi.kbeg; st.round = 0; st.live = true;
    int m0 = st.wi.m0, n0 = st.wi.n0, bz = st.wi.b;
    int nsteps = (st.wi.kend - st.wi.kbeg + BK - 1) / BK;

    {
        const TileSrc c0 = st.take(p, b, G, ntm, ntn, total_items, kstrideA, kstrideB);
        const TileSrc c1 = st.take(p, b, G, ntm, ntn, total_items, kstrideA, kstrideB);
        PACED_STAGE_HALF(c0.rB, sw + 0 * HALF, offB, c0.sB);
        PACED_STAGE_HALF(c0.rA, sw + 1 * HALF, offA, c0.sA);
        PACED_STAGE_HALF(c0.rB, sw + 2 * HALF, offB, c0.sB + halfB);
        PACED_STAGE_HALF(c0.rA, sw + 3 * HALF, offA, c0.sA + halfA);
        PACED_STAGE_HALF(c1.rB, sw + 4 * HALF, offB, c1.sB);
        PACED_STAGE_HALF(c1.rA, sw + 5 * HALF, offA, c1.sA);
        PACED_STAGE_HALF(c1.rB, sw + 6 * HALF, offB, c1.sB + halfB);
        PACED_STAGE_HALF(c1.rA, sw + 7 * HALF, offA, c1.sA + halfA);
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();       // second wave group runs one barrier behind
    TileSrc c2 = st.take(p, b, G, ntm, ntn, total_items, kstrideA, kstrideB);

    f32x4 acc[8][4];
    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    int base = 0;                                     // ring slot of this K step's first half-tile
    for (int round = 0;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A tile-ordered bit matrix (ldaux < 0): this lane's 16 bytes of the tile at ((tile * 512 + thread) * 16.  The mask
        // is requested here, ahead of the tile's whole K loop, by a load the compiler cannot see (see epilogue_rows_impl);
        // it is older than every operand load of the loop, whose counted waits therefore cover it from the second K step on.
        u32x4 tbits = {0u, 0u, 0u, 0u};
        unsigned char* tile_bits = nullptr;
        if ((EPI == EPI_RELU_DROP || EPI == EPI_MASK_BITS) && p.aux && p.ldaux < 0) {
            tile_bits = const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.aux)) +
                        ((size_t)((m0 >> 8) * ntn + (n0 >> 8)) * 512 + tid) * 16;
            if (EPI == EPI_MASK_BITS) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(tbits) : "v"(tile_bits) : "memory");     // read once
        }
        for (int t = 0; t < nsteps; ++t) {
            const int sB0 = base, sA0 = ring_wrap(base + 1), sB1 = ring_wrap(base + 2), sA1 = ring_wrap(base + 3);
            const int d1 = ring_wrap(base + 8), d2 = ring_wrap(base + 9);      // phases 3 and 4 reuse sB0 and sA0
            // ---- phase 1
            {
                const char* lB = smem + sB0 * HALF;
                const char* lA = smem + sA0 * HALF;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) fb0[j][kk] = read_frag<TB, 128, true>(lB, wc * 32 + j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) fa[i][kk] = read_frag<TA, 128, true>(lA, wr * 64 + i * 16, kk, lane);
            }
            PACED_STAGE_HALF(c2.rB, sw + d1 * HALF, offB, c2.sB);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            PACED_PINNED_BARRIER();
            PACED_WAIT_LDS_READS();
            quadrant_mfma<NATURAL>(fa, fb0, acc, 0, 0);
            PACED_PINNED_BARRIER();
            // ---- phase 2
            {
                const char* lB = smem + sB1 * HALF;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) fb1[j][kk] = read_frag<TB, 128, true>(lB, wc * 32 + j * 16, kk, lane);
            }
            PACED_STAGE_HALF(c2.rA, sw + d2 * HALF, offA, c2.sA);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            PACED_PINNED_BARRIER();
            PACED_WAIT_LDS_READS();
            quadrant_mfma<NATURAL>(fa, fb1, acc, 0, 1);
            PACED_PINNED_BARRIER();
            // ---- phase 3
            {
                const char* lA = smem + sA1 * HALF;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) fa[i][kk] = read_frag<TA, 128, true>(lA, wr * 64 + i * 16, kk, lane);
            }
            PACED_STAGE_HALF(c2.rB, sw + sB0 * HALF, offB, c2.sB + halfB);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            PACED_PINNED_BARRIER();
            PACED_WAIT_LDS_READS();
            quadrant_mfma<NATURAL>(fa, fb1, acc, 1, 1);
            PACED_PINNED_BARRIER();
            // ---- phase 4
            PACED_STAGE_HALF(c2.rA, sw + sA0 * HALF, offA, c2.sA + halfA);
            // the scalar bookkeeping of the next K step runs here, under the partner group's MFMAs
            c2 = st.take(p, b, G, ntm, ntn, total_items, kstrideA, kstrideB);
            base = ring_wrap(base + 4);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            PACED_PINNED_BARRIER();
            quadrant_mfma<NATURAL>(fa, fb0, acc, 1, 0);
            PACED_PINNED_BARRIER();
        }
        // The second wave group still owes the barrier that ends its last phase.  Join it here, so that both groups
        // run their epilogues side by side; otherwise that barrier is met by this group's first barrier of the next
        // item, and each group sits idle through the other's whole epilogue.
        if (wr == 0) __builtin_amdgcn_s_barrier();
        if (EPI == EPI_BF16 || EPI == EPI_RES_DROP || EPI == EPI_RELU_DROP || EPI == EPI_MASK || EPI == EPI_MASK_BITS) {
            char* s1 = sw + ring_wrap(base + 8) * HALF;      // the finished K step's B half 1 and A half 1 slots
            char* s2 = sw + ring_wrap(base + 9) * HALF;
            // (the host refuses a tile-ordered mask with fewer than two K steps, so the loop's waits have covered the load)
            if (EPI == EPI_MASK_BITS) asm volatile("; tile mask bits arrived" : "+v"(tbits) :: "memory");
            epilogue_rows<EPI>(p, batch_c<EPI>(p, bz), m0, n0, wr, wc, lane, acc, s1, s1 + 8192, s2, s2 + 8192, tbits, tile_bits);
        } else {
            epilogue<CfgPhased, EPI>(p, batch_c<EPI>(p, bz), m0, n0, wr, wc, lane, acc);
        }
        L = item_of(b, ++round, G, total_items);
        if (L < 0) break;
        {
            // position and K-step count of the next item (the stream decoded it already for its loads)
            const int ntiles = ntm * ntn;
            const int nsplit = (p.K + p.k_per_split - 1) / p.k_per_split;
            const int per_batch = ntiles * nsplit;
            bz = L / per_batch;
            const int rr = L - bz * per_batch;
            const int split = rr / ntiles, tt = rr - split * ntiles;
            int tm, tn;
            if (ntm >= ntn) { tm = tt / ntn; tn = tt - tm * ntn; } else { tn = tt / ntm; tm = tt - tn * ntm; }
            m0 = tm * 256; n0 = tn * 256;
            const int kb = split * p.k_per_split, ke = min(p.K, kb + p.k_per_split);
            nsteps = (ke - kb + BK - 1) / BK;
        }
        if (wr == 1) __builtin_amdgcn_s_barrier();       // fall one barrier behind again
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// Pool scoring for FEW queries (Q <= 128; taken by default up to 64): the regime where the pool's bytes bind (Q FLOP per pool byte against a
// ridge of ~312), served by a kernel built around the HBM stream instead of the matrix cores.
//   * The queries never touch LDS: wave w keeps the MFMA B fragments of queries 16w .. 16w+15 for the whole
//     contraction (D/32 fragments of 4 registers) in registers for the whole launch.
//   * All of a workgroup's LDS bandwidth and capacity serve the pool rows: a ring of NS tiles of 128 rows x 64
//     columns (16 KiB each, the K-contiguous image of the small-tile kernels), filled by LDS-DMA; NS - 1 tiles stay in
//     flight per compute unit and the stream runs on across the workgroup's row blocks, so the only drain is at the
//     end of the launch (or at a flush, below).
//   * Scores that pass their query's threshold (read once per launch: a threshold that is stale is still a valid lower
//     bound) are collected in an LDS list and flushed to the candidate lists in bulk -- one returning global atomic per
//     query and flush -- because a returning atomic per row block would have to wait for every older LDS-DMA, i.e.
//     drain the ring once per block.
// NW = ceil(Q / 16) waves; a row block of 128 pool rows is D/64 K steps of {counted wait | barrier | issue the tile NS-1
// steps ahead | 16 fragment reads, 16 MFMAs}.  Up to 64 queries: 4 waves, 4 ring slots, TWO workgroups per compute unit,
// so that one workgroup's fragment reads and MFMAs run under the other's waits (one workgroup alone serialises its K step
// and reaches 4.1 TB/s); up to 128: 8 waves, 8 slots, one workgroup.
constexpr int PS_TILE = 16384;              // bytes per ring slot
#ifndef PS_AUX
#define PS_AUX 2                            // cache-policy bits of the pool stream's LDS-DMA: 2 = nt, the rows are read once (0: 2 % slower)
#endif
// (entries of the LDS survivor list, a template parameter LIST: at least 4 rows x the launch's queries, which is what is
// appended between two checks of its fill)
constexpr int PS_MAXF = 24;                 // D <= 768: fragments per wave
constexpr int PS_MAXQ = 128;                // queries per launch

template <int LIST>
struct PoolStreamLds {
    static constexpr int kList = LIST;
    unsigned long long key[LIST];
    unsigned short qry[LIST];
    unsigned int blk_count[32];             // passing scores of the current row block, per group of 4 rows sixteen apart
    unsigned int list_count;
    unsigned int qcount[PS_MAXQ];
    unsigned int qbase[PS_MAXQ];
};

// LDS-DMA of stream position s (row block s / KS of this workgroup, K step s % KS) into its ring slot; positions past the
// end issue the same instructions through an empty descriptor (zeros nobody reads), so that wait counts stay uniform.
// (bi, ki) = (s / KS, s % KS), kept by the caller: stream positions are issued in order, so no division is needed here.
template <int NW, int NS>
__device__ __forceinline__ void pool_stream_issue(const GemmParams& p, char* smem, int s, int total_steps, int bi, int ki, int b, int G, int w,
                                                  const uint32_t (&offA)[16 / NW]) {
    __amdgpu_buffer_rsrc_t r;
    uint32_t soff = 0;
    if (s < total_steps) {
        const int blk = b + bi * G, m0 = blk * 128;
        const int rows = min(128, p.M - m0);
        r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)m0 * p.lda), 0, (uint32_t)(((size_t)(rows - 1) * p.lda + p.K) * 2), 0x00020000);
        soff = (uint32_t)ki * (BK * 2);
    } else {
        r = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
    }
    char* dst = smem + (s % NS) * PS_TILE;
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(dst + (i * NW + w) * 1024), 16, offA[i], soff, 0, PS_AUX);
}

// Survivors of the LDS list to the candidate lists: histogram per query, one returning atomic per query with survivors
// (inline assembly with its own wait: this is the one place that drains the ring), scatter.  Called by every thread.
template <int NW, class Lds>
__device__ __forceinline__ void pool_stream_flush(const GemmParams& p, Lds& L, int tid) {
    __syncthreads();
    const unsigned int cnt = min(L.list_count, (unsigned int)Lds::kList);
    for (unsigned int e = tid; e < cnt; e += NW * 64) atomicAdd(&L.qcount[L.qry[e]], 1u);
    __syncthreads();
    for (int q = tid; q < p.N; q += NW * 64) {
        const unsigned int c = L.qcount[q];
        if (c) {
            unsigned int base;
            unsigned int* slot = p.tk_cnt + q;
            asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(base) : "v"(slot), "v"(c) : "memory");
            L.qbase[q] = base;
            L.qcount[q] = 0;
        }
    }
    __syncthreads();
    for (unsigned int e = tid; e < cnt; e += NW * 64) {
        const unsigned int q = L.qry[e];
        const unsigned int at = L.qbase[q] + atomicAdd(&L.qcount[q], 1u);
        if (at < (unsigned int)p.tk_cap) p.tk_cand[(size_t)q * p.tk_cap + at] = L.key[e];
    }
    __syncthreads();
    for (int q = tid; q < p.N; q += NW * 64) L.qcount[q] = 0;
    if (tid == 0) L.list_count = 0;
    __syncthreads();
}

template <int NW, int NS, int LIST>
__global__ void __launch_bounds__(NW * 64) pool_stream_kernel(GemmParams p, int nblocks) {
    constexpr int IA = 16 / NW;              // LDS-DMA instructions per wave and tile
    static_assert(16 % NW == 0 && NS >= 3 && (NS - 2) * IA <= 63 && NW * 16 <= PS_MAXQ && LIST >= 4 * NW * 16, "tile split over the waves; pending count must fit vmcnt");
    using Lds = PoolStreamLds<LIST>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds& L = *reinterpret_cast<Lds*>(smem + NS * PS_TILE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, b = blockIdx.x;
    const int KS = p.K / BK;                 // K steps per row block
    const int nf = p.K / 32;                 // B fragments per wave

    // ---- queries of this wave into registers (hidden loads: see epilogue_rows_impl for why the compiler must not see them)
    const int n = w * 16 + (lane & 15);
    const int nc = min(n, p.N - 1);          // lanes past Q score a copy of the last query; their scores are never kept
    bf16x8 fb[PS_MAXF];
#pragma unroll
    for (int f = 0; f < PS_MAXF; ++f) {
        const bf16* src = p.B + (size_t)nc * p.ldb + min(f, nf - 1) * 32 + (lane >> 4) * 8;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fb[f]) : "v"(src) : "memory");
    }
    float thr;
    {
        const float* src = p.tk_thr + nc;
        asm volatile("global_load_dword %0, %1, off" : "=v"(thr) : "v"(src) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) ; hidden-load wait" : "+v"(thr) :: "memory");
#pragma unroll
    for (int f = 0; f < PS_MAXF; ++f) asm volatile("" : "+v"(fb[f]));
    if (n >= p.N) thr = INFINITY;
    if (tid < 32) L.blk_count[tid] = 0;
    if (tid == 0) L.list_count = 0;
    for (int i = tid; i < PS_MAXQ; i += NW * 64) L.qcount[i] = 0;

    uint32_t offA[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i) offA[i] = stage_offset<false, 128>((i * NW + w) * 64 + lane, p.lda);
    const int my_blocks = b < nblocks ? (nblocks - b + G - 1) / G : 0;
    const int total_steps = my_blocks * KS;

    // stream position s = (block index of this workgroup) * KS + K step
    int is_b = 0, is_k = 0;                 // row block and K step of the next position to be issued (positions go out in order)
    auto issue = [&](int s) {
        pool_stream_issue<NW, NS>(p, smem, s, total_steps, is_b, is_k, b, G, w, offA);
        if (++is_k == KS) { is_k = 0; ++is_b; }
    };

    auto flush = [&]() { pool_stream_flush<NW>(p, L, tid); };
    constexpr unsigned int PS_LIST = LIST;

    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // One workgroup per CU (eight waves): its waves all stand at the same barrier, nobody fills the wait, so two K steps
    // share one (two ring slots are free at a time, NS - 2 tiles in flight).  Two workgroups per CU: one K step per barrier.
    constexpr bool PAIR = NW == 8;
    static_assert(!PAIR || NS >= 6, "pairs of K steps need a ring of six slots or more");
#pragma unroll
    for (int s = 0; s < NS - 1 - (PAIR ? 1 : 0); ++s) issue(s);
    unsigned int pending = 0;                        // entries in the LDS list (same value in every thread)
    bool clear_counts = false;
    for (int blk_i = 0, s = 0; blk_i < my_blocks; ++blk_i) {
        // The K steps of a row block, unrolled by hand (a macro, not a loop: with the barrier inside, the compiler keeps a
        // loop rolled, indexes the B fragments dynamically and parks them in scratch memory): the B fragment of a step
        // is then a fixed register.
#define PS_KSTEP(ks)                                                                                                   \
        if ((ks) < KS) {                                                                                               \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IA) : "memory");                                        \
            PACED_PINNED_BARRIER();                                                                                    \
            if (clear_counts) { if (tid < 32) L.blk_count[tid] = 0; clear_counts = false; }                            \
            issue(s + NS - 1);                                                                                         \
            const char* tile = smem + (s % NS) * PS_TILE;                                                              \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                         \
                bf16x8 fa[8];                                                                                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) fa[i] = read_frag<false, 128>(tile, i * 16, kk, lane);   \
                _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                          \
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[2 * (ks) + kk], acc[i], 0, 0, 0);       \
            }                                                                                                          \
            ++s;                                                                                                       \
        }
#define PS_TILE_MFMA(ks, tile)                                                                                         \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                         \
                bf16x8 fa[8];                                                                                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) fa[i] = read_frag<false, 128>(tile, i * 16, kk, lane);   \
                _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                          \
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[2 * (ks) + kk], acc[i], 0, 0, 0);       \
            }
        // K steps ks and ks + 1 behind one barrier (the second only if the row block has it).  The row fragments go through
        // two register sets of four by hand, each read four MFMAs before it is used: left to itself the compiler
        // keeps two fragments in flight, and every MFMA then waits an LDS latency for its operand.
#define PS_RD4(set, tile, kk, half) _Pragma("unroll") for (int i = 0; i < 4; ++i) set[i] = read_frag<false, 128>(tile, (4 * (half) + i) * 16, kk, lane); __builtin_amdgcn_sched_barrier(0);
#define PS_MM4(set, f, half) _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[4 * (half) + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(set[i], fb[f], acc[4 * (half) + i], 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
#define PS_KPAIR(ks)                                                                                                   \
        if ((ks) < KS) {                                                                                               \
            const bool two = (ks) + 1 < KS;                                                                            \
            if (two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 4) * IA) : "memory");                               \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * IA) : "memory");                                   \
            PACED_PINNED_BARRIER();                                                                                    \
            if (clear_counts) { if (tid < 32) L.blk_count[tid] = 0; clear_counts = false; }                            \
            issue(s + NS - 2);                                                                                         \
            if (two) issue(s + NS - 1);                                                                                \
            const char* tile0 = smem + (s % NS) * PS_TILE;                                                             \
            const char* tile1 = smem + ((s + 1) % NS) * PS_TILE;                                                       \
            constexpr int f0 = 2 * (ks), f1 = 2 * ((ks) + 1 < PS_MAXF / 2 ? (ks) + 1 : (ks));                          \
            bf16x8 s0[4], s1[4];                    /* two sets of four fragments: four to eight reads stay ahead of the MFMAs */ \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            PS_RD4(s0, tile0, 0, 0) PS_RD4(s1, tile0, 0, 1)                                                             \
            PS_MM4(s0, f0, 0) PS_RD4(s0, tile0, 1, 0)                                                                   \
            PS_MM4(s1, f0, 1) PS_RD4(s1, tile0, 1, 1)                                                                   \
            if (two) {                                                                                                 \
                PS_MM4(s0, f0 + 1, 0) PS_RD4(s0, tile1, 0, 0)                                                           \
                PS_MM4(s1, f0 + 1, 1) PS_RD4(s1, tile1, 0, 1)                                                           \
                PS_MM4(s0, f1, 0) PS_RD4(s0, tile1, 1, 0)                                                               \
                PS_MM4(s1, f1, 1) PS_RD4(s1, tile1, 1, 1)                                                               \
                PS_MM4(s0, f1 + 1, 0) PS_MM4(s1, f1 + 1, 1)                                                             \
            } else {                                                                                                   \
                PS_MM4(s0, f0 + 1, 0) PS_MM4(s1, f0 + 1, 1)                                                             \
            }                                                                                                          \
            s += two ? 2 : 1;                                                                                          \
        }
        if (PAIR) {
            PS_KPAIR(0) PS_KPAIR(2) PS_KPAIR(4) PS_KPAIR(6) PS_KPAIR(8) PS_KPAIR(10)
        } else {
            PS_KSTEP(0) PS_KSTEP(1) PS_KSTEP(2) PS_KSTEP(3) PS_KSTEP(4) PS_KSTEP(5)
            PS_KSTEP(6) PS_KSTEP(7) PS_KSTEP(8) PS_KSTEP(9) PS_KSTEP(10) PS_KSTEP(11)
        }
#undef PS_KPAIR
#undef PS_RD4
#undef PS_MM4
#undef PS_TILE_MFMA
#undef PS_KSTEP
        static_assert(PS_MAXF == 24, "twelve K steps are written out above");
        // ---- row block done: keep what passes.  lane -> query, registers -> 4 consecutive pool rows per 16-row sub-block
        const int m0 = (b + blk_i * G) * 128;
        unsigned int pass = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + (lane >> 4) * 4 + r;
                pass |= (acc[i][r] >= thr && m < p.M ? 1u : 0u) << (i * 4 + r);
            }
        // Counts per register position (bit = 4 * sub-block + r: 4 lane groups x Q queries <= 1024 scores each), so that
        // what is appended between two checks of the list's fill never exceeds the list.
        if (pass) {
#pragma unroll
            for (int bit = 0; bit < 32; ++bit)
                if ((pass >> bit) & 1u) atomicAdd(&L.blk_count[bit], 1u);
        }
        __syncthreads();
        unsigned int total = 0;
        u32x4 c32[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c32[i] = *reinterpret_cast<const u32x4*>(&L.blk_count[4 * i]);
            total += c32[i][0] + c32[i][1] + c32[i][2] + c32[i][3];
        }
        if (total) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned int c = c32[i][r];
                    if (c == 0) continue;
                    if (pending + c > (unsigned int)PS_LIST) { flush(); pending = 0; }
                    if ((pass >> (i * 4 + r)) & 1u) {
                        const unsigned int at = atomicAdd(&L.list_count, 1u);
                        const int m = m0 + i * 16 + (lane >> 4) * 4 + r;
                        if (at < (unsigned int)PS_LIST) { L.key[at] = topk_key(acc[i][r], p.tk_row_base + (unsigned int)m); L.qry[at] = (unsigned short)n; }
                    }
                    pending += c;
                }
            clear_counts = true;                     // after the next barrier: every thread has read the counts by then
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (pending) flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

namespace {

// Process-wide defaults (benchmarks and tests set them through paced_gemm_set_*) and the options of the call under way on
// this thread (paced_gemm_bf16_ex; a field < 0 = take the process default).  Launch code reads the eff_* functions only.
int g_force_tile = 0;    // 0 = heuristic, 128 or 256 = forced (paced_gemm_set_tile, for benchmarking)
int g_grid_limit = 0;    // 0 = one phased workgroup per CU; n > 0: at most n (paced_gemm_set_grid; PACED_GEMM_GRID at load)
bool g_grid_env_read = false;
struct CallOpts { int grid_limit = -1, tile = -1, ring = -1; };
thread_local CallOpts t_call;
thread_local int t_last_kernel = 0;   // tile of this thread's most recent launch (64, 128, 256), +1 when the ring variant took it

int eff_tile() { return t_call.tile >= 0 ? t_call.tile : g_force_tile; }
int eff_grid_limit() {
    if (t_call.grid_limit >= 0) return t_call.grid_limit;
    if (!g_grid_env_read) { const char* e = getenv("PACED_GEMM_GRID"); if (e && g_grid_limit == 0) g_grid_limit = atoi(e); g_grid_env_read = true; }
    return g_grid_limit;
}

int cu_count() { return paced_cu_count(); }

int g_ring = -1;         // -1 = not set yet (PACED_GEMM_RING read on first use), 0 off, 1 on (paced_gemm_set_ring)
bool ring_enabled() {
    if (t_call.ring >= 0) return t_call.ring != 0;
    if (g_ring < 0) { const char* e = getenv("PACED_GEMM_RING"); g_ring = (e && atoi(e) == 0) ? 0 : 1; }
    return g_ring != 0;
}

template <class Cfg, bool TA, bool TB, int EPI, int NS>
int launch_ring(const GemmParams& p, int ntm, int ntn, int total, hipStream_t stream) {
    constexpr int smem = NS * (Cfg::BM + Cfg::BN) * BK * 2;
    if (smem > 65536) {
        static PerDeviceOnce once;
        bool& attr_set = once.flag();
        if (!attr_set) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring_kernel<Cfg, TA, TB, EPI, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) { paced_set_error("hipFuncSetAttribute(%d bytes LDS) failed: %s", smem, hipGetErrorString(e)); return PACED_ERR_HIP; }
            attr_set = true;
        }
    }
    hipLaunchKernelGGL((gemm_ring_kernel<Cfg, TA, TB, EPI, NS>), dim3(total), dim3(Cfg::WGM * Cfg::WGN * 64), smem, stream, p, ntm, ntn, total);
    t_last_kernel = Cfg::BM + 1;
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

template <class Cfg, bool TA, bool TB, int EPI>
int launch_cfg(GemmParams p, int splits, hipStream_t stream) {
    const int ntm = (p.M + Cfg::BM - 1) / Cfg::BM, ntn = (p.N + Cfg::BN - 1) / Cfg::BN;
    const int ktiles = (p.K + BK - 1) / BK;
    constexpr bool PHASED = std::is_same<Cfg, CfgPhased>::value;
    int slots = cu_count() * (PHASED ? 1 : std::is_same<Cfg, CfgTiny>::value ? 6 : 3);     // workgroups resident per CU
    if (PHASED) {
        // At most g_grid_limit workgroups: the compute units the grid leaves out stay free for the kernels of other
        // streams for as long as this launch runs (a phased workgroup holds all of a unit's LDS and registers until
        // the whole launch is over, so nothing else starts on a unit it has taken).
        const int limit = eff_grid_limit();
        if (limit > 0 && limit < slots) slots = limit;
    }
    constexpr bool RINGABLE = !PHASED && EPI != EPI_TOPK_DENSE && EPI != EPI_TOPK_FILTER;
    const bool ring_on = ring_enabled();
    if (splits <= 0) {          // auto: fill the grid once, keep >= 4 K steps per slice
        splits = 1;
        // one workgroup per CU, or no split at all up to the ring kernel's three per CU, when it will take the launch;
        // else one per resident slot
        const int tiles = ntm * ntn * p.batch;
        const bool tiny_cfg = std::is_same<Cfg, CfgTiny>::value;
        const int fill = !(RINGABLE && ring_on) ? slots : tiles <= cu_count() ? cu_count() : (tiny_cfg && tiles <= 3 * cu_count()) ? tiles : slots;
        if (EPI == EPI_ATOMIC_F32 && ntm * ntn * p.batch < fill) splits = fill / (ntm * ntn * p.batch);
        if (splits > ktiles / 4) splits = ktiles / 4 > 0 ? ktiles / 4 : 1;
    }
    if (splits > ktiles) splits = ktiles;
    const int tiles_per_split = (ktiles + splits - 1) / splits;
    p.k_per_split = tiles_per_split * BK;
    splits = (ktiles + tiles_per_split - 1) / tiles_per_split;
    const int total = ntm * ntn * splits * p.batch;
    const int grid = total < slots ? total : slots;
    if constexpr (RINGABLE) {
        // few tiles per CU: latency is hidden inside the workgroup (ring of K steps) instead of by occupancy
        const int per_cu = (total + cu_count() - 1) / cu_count();
        if (ring_on && tiles_per_split >= 3) {
            if constexpr (std::is_same<Cfg, CfgTiny>::value) {
                if (per_cu == 1) return launch_ring<Cfg, TA, TB, EPI, 8>(p, ntm, ntn, total, stream);
                if (per_cu == 2) return launch_ring<Cfg, TA, TB, EPI, 4>(p, ntm, ntn, total, stream);
                if (per_cu == 3) return launch_ring<Cfg, TA, TB, EPI, 3>(p, ntm, ntn, total, stream);
            } else {
                if (per_cu == 1) return launch_ring<Cfg, TA, TB, EPI, 4>(p, ntm, ntn, total, stream);
            }
        }
    }
    constexpr int smem = PHASED ? 163840 : (Cfg::BM + Cfg::BN) * BK * 2;
    if (smem > 65536) {
        static PerDeviceOnce once;        // one per template instantiation
        bool& attr_set = once.flag();
        if (!attr_set) {
            hipError_t e;
            if constexpr (PHASED) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_phased_kernel<TA, TB, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            else e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<Cfg, TA, TB, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) { paced_set_error("hipFuncSetAttribute(%d bytes LDS) failed: %s", smem, hipGetErrorString(e)); return PACED_ERR_HIP; }
            attr_set = true;
        }
    }
    if constexpr (PHASED) hipLaunchKernelGGL((gemm_phased_kernel<TA, TB, EPI>), dim3(grid), dim3(512), smem, stream, p, ntm, ntn, total);
    else hipLaunchKernelGGL((gemm_bf16_kernel<Cfg, TA, TB, EPI>), dim3(grid), dim3(Cfg::WGM * Cfg::WGN * 64), smem, stream, p, ntm, ntn, total);
    t_last_kernel = Cfg::BM;
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

template <bool TA, bool TB, int EPI>
int launch(const GemmParams& p, int splits, hipStream_t stream) {
    const int force_tile = eff_tile();
    if (EPI == EPI_ATOMIC_F32 && splits <= 0 && force_tile == 0 && ring_enabled()) {
        // Short contraction into a mid-sized output (the decoder's weight gradients): the launch is bound by the
        // output, every split adds the whole output once more in atomics, and a big tile leaves CUs empty; take
        // 64x64 tiles, no more splits than it takes to give every CU a workgroup, and the ring kernel.
        const long long t64 = (long long)((p.M + 63) / 64) * ((p.N + 63) / 64) * p.batch;
        const int ktiles = (p.K + BK - 1) / BK;
        if (ktiles >= 3 && ktiles <= 32 && t64 <= 3LL * cu_count()) return launch_cfg<CfgTiny, TA, TB, EPI>(p, 0, stream);
    }
    bool large;
    if (force_tile == 128) large = false;
    else if (force_tile == 256) large = true;
    else {
        // the 256x256 tile pays off once there is about one tile per CU (split slices count for atomics)
        const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256) * p.batch;
        large = p.M >= 256 && p.N >= 256 && (t256 >= 200 || (EPI == EPI_ATOMIC_F32 && t256 >= 9));
    }
    // the phased kernel moves bf16 outputs (and their residual / mask operand) in 16-byte row pieces
    const bool rows16 = (EPI == EPI_F32 || EPI == EPI_ATOMIC_F32 || EPI == EPI_TOPK_DENSE || EPI == EPI_TOPK_FILTER) ||
                        (p.N % 8 == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 &&
                         (!(EPI == EPI_RES_DROP || EPI == EPI_MASK) || (p.ldaux % 8 == 0 && ((uintptr_t)p.aux & 15) == 0)));
    // ... and numbers dropout's quads with 32 bits
    const bool quads32 = !(EPI == EPI_RES_DROP || EPI == EPI_RELU_DROP) || p.drop.thresh == 0 || p.drop_narrow;
    if (large && rows16 && quads32) return launch_cfg<CfgPhased, TA, TB, EPI>(p, splits, stream);
    if ((EPI == EPI_RELU_DROP || EPI == EPI_MASK_BITS) && p.aux && p.ldaux < 0) {
        paced_set_error("paced_gemm_bf16: a tile-ordered bit matrix (ldaux = -1) needs a shape that takes the 256x256 kernel; ask paced_gemm_tile_bits_bytes first");
        return PACED_ERR_ARG;
    }
    // fewer 128x128 tiles than CUs and no K split to make up for it: quarter-size tiles
    const long long t128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.batch;
    // (with the ring kernel half a grid of 128x128 tiles beats more than two 64x64 tiles per CU: half the operand traffic)
    const int tiny_below = ring_enabled() ? cu_count() / 2 : cu_count();
    const bool tiny = force_tile == 64 || (force_tile == 0 && t128 < tiny_below && (EPI != EPI_ATOMIC_F32 || splits == 1));
    if (tiny) return launch_cfg<CfgTiny, TA, TB, EPI>(p, splits, stream);
    return launch_cfg<CfgSmall, TA, TB, EPI>(p, splits, stream);
}

template <bool TA, bool TB>
int dispatch_epi(const GemmParams& p, int epi, int splits, hipStream_t s) {
    switch (epi) {
        case EPI_BF16: return launch<TA, TB, EPI_BF16>(p, splits, s);
        case EPI_RES_DROP: return launch<TA, TB, EPI_RES_DROP>(p, splits, s);
        case EPI_RELU_DROP: return launch<TA, TB, EPI_RELU_DROP>(p, splits, s);
        case EPI_MASK: return launch<TA, TB, EPI_MASK>(p, splits, s);
        case EPI_MASK_BITS: return launch<TA, TB, EPI_MASK_BITS>(p, splits, s);
        case EPI_F32: return launch<TA, TB, EPI_F32>(p, splits, s);
        case EPI_ATOMIC_F32: return launch<TA, TB, EPI_ATOMIC_F32>(p, splits, s);
        case EPI_TOPK_DENSE: if (!TA && !TB) return launch<false, false, EPI_TOPK_DENSE>(p, splits, s); break;
        case EPI_TOPK_FILTER: if (!TA && !TB) return launch<false, false, EPI_TOPK_FILTER>(p, splits, s); break;
    }
    return PACED_ERR_ARG;
}

}  // namespace

extern "C" int paced_gemm_set_tile(int tile) {
    PACED_CHECK_ARG(tile == 0 || tile == 64 || tile == 128 || tile == 256, "paced_gemm_set_tile: tile must be 0 (auto), 64, 128 or 256");
    g_force_tile = tile;
    return PACED_OK;
}

extern "C" int paced_gemm_last_kernel(void) { return t_last_kernel; }

extern "C" long long paced_gemm_tile_bits_bytes(int M, int N) {
    if (M < 256 || N < 256 || N % 8 != 0 || g_force_tile == 128 || g_force_tile == 64) return 0;
    if ((unsigned long long)M * (unsigned long long)N > (1ull << 34)) return 0;      // dropout would send the writer elsewhere
    const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
    if (g_force_tile != 256 && t256 < 200) return 0;
    return t256 * 8192;
}

extern "C" int paced_gemm_set_grid(int workgroups) {
    PACED_CHECK_ARG(workgroups >= 0, "paced_gemm_set_grid: the limit must be >= 0 (0 = one workgroup per compute unit)");
    g_grid_limit = workgroups;
    g_grid_env_read = true;
    return PACED_OK;
}

extern "C" int paced_gemm_get_grid(void) { return g_grid_limit; }

extern "C" int paced_gemm_set_ring(int on) {
    g_ring = on ? 1 : 0;
    return PACED_OK;
}

// Scores pool rows [M, K] (p.A) against p.N <= 128 queries (p.B) and appends what passes the thresholds (EPI_TOPK_FILTER's
// contract) with the streaming kernel above.  Returns PACED_ERR_ARG when the shape is outside what that kernel covers.
template <int NW, int NS, int LIST, int PER_CU>
static int launch_pool_stream(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = NS * PS_TILE + (int)sizeof(PoolStreamLds<LIST>);
    static_assert(PER_CU * smem <= 163840, "rings and survivor lists must fit the 160 KiB of a compute unit");
    static PerDeviceOnce once;               // one per instantiation
    bool& attr_set = once.flag();
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_stream_kernel<NW, NS, LIST>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) { paced_set_error("hipFuncSetAttribute(%d bytes LDS) failed: %s", smem, hipGetErrorString(e)); return PACED_ERR_HIP; }
        attr_set = true;
    }
    const int nblocks = (p.M + 127) / 128;
    const int grid = nblocks < PER_CU * cu_count() ? nblocks : PER_CU * cu_count();
    hipLaunchKernelGGL((pool_stream_kernel<NW, NS, LIST>), dim3(grid), dim3(NW * 64), smem, stream, p, nblocks);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

int paced_pool_stream_launch(const GemmParams& p, hipStream_t stream) {
    if (p.N > PS_MAXQ || p.K > PS_MAXF * 32 || p.K % BK != 0 || p.M <= 0) return PACED_ERR_ARG;
    // (three workgroups of three slots per unit: 3.95 ms per 10 M-row sweep against 3.86 for two of four)
    if (p.N <= 64) return launch_pool_stream<4, 4, 1024, 2>(p, stream);
    return launch_pool_stream<8, 8, 1024, 1>(p, stream);
}

int paced_gemm_launch(const GemmParams& p, bool trans_a, bool trans_b, int epilogue, int splits, hipStream_t s) {
    if (!trans_a && !trans_b) return dispatch_epi<false, false>(p, epilogue, splits, s);
    if (!trans_a && trans_b) return dispatch_epi<false, true>(p, epilogue, splits, s);
    return dispatch_epi<true, true>(p, epilogue, splits, s);
}

static int gemm_entry(const void* A, const void* B, void* C, const void* aux,
                      int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
                      int trans_a, int trans_b, int epilogue, float alpha, float mask_scale,
                      float drop_p, uint32_t drop_seed, int splits, void* stream,
                      int batch, long long strideA, long long strideB, long long strideC,
                      long long a_heads = 0, long long b_heads = 0, long long c_heads = 0) {
    PACED_CHECK_ARG(A && B && C, "paced_gemm_bf16: null operand");
    if (a_heads || b_heads || c_heads) {
        PACED_CHECK_ARG(batch == 1 && a_heads >= 0 && b_heads >= 0 && c_heads >= 0 && a_heads % 64 == 0 && b_heads % 64 == 0 && c_heads % 8 == 0,
                        "paced_gemm_bf16_heads: head strides must be non-negative multiples of 64 (A, B) / 8 (C); no batches");
        const long long a_rows = trans_a ? K : M, a_feat = trans_a ? M : K;
        if (a_heads) PACED_CHECK_ARG(lda == 64 && a_feat % 64 == 0 && a_heads >= a_rows * 64 && (size_t)(a_feat / 64) * a_heads * 2 < 0xFFFFFFF0ull,
                                     "paced_gemm_bf16_heads: a head-major A needs lda = 64, a feature count that is a multiple of 64 and heads >= rows * 64 elements apart (below 4 GB in all)");
        if (b_heads) PACED_CHECK_ARG(trans_b && ldb == 64 && N % 64 == 0 && b_heads >= (long long)K * 64 && (size_t)(N / 64) * b_heads * 2 < 0xFFFFFFF0ull && !(a_heads && trans_a),
                                     "paced_gemm_bf16_heads: a head-major B must be the transposed operand (ldb = 64, N %% 64 == 0, heads >= K * 64 apart), and not beside a transposed head-major A");
        if (c_heads) PACED_CHECK_ARG(epilogue == EPI_BF16 && N % 64 == 0 && c_heads >= (long long)M * 64,
                                     "paced_gemm_bf16_heads: a head-major C needs the plain bf16 epilogue, N %% 64 == 0 and heads >= M * 64 elements apart");
    }
    PACED_CHECK_ARG(batch >= 1 && batch <= 65536, "paced_gemm_bf16_batched: batch must be in 1..65536 (got %d)", batch);
    if (batch > 1) {
        PACED_CHECK_ARG(epilogue == EPI_BF16 || epilogue == EPI_F32 || epilogue == EPI_ATOMIC_F32,
                        "paced_gemm_bf16_batched: only the plain bf16, fp32 and atomic epilogues are batched");
        PACED_CHECK_ARG(strideA % 8 == 0 && strideB % 8 == 0 && strideC % 8 == 0 && strideA >= 0 && strideB >= 0 && strideC >= 0,
                        "paced_gemm_bf16_batched: strides must be non-negative multiples of 8 elements");
    }
    PACED_CHECK_ARG(M > 0 && N > 0 && K > 0, "paced_gemm_bf16: M, N, K must be positive (got %d %d %d)", M, N, K);
    PACED_CHECK_ARG(epilogue >= 0 && epilogue < EPI_COUNT, "paced_gemm_bf16: unknown epilogue %d", epilogue);
    PACED_CHECK_ARG(!(trans_a && !trans_b), "paced_gemm_bf16: (trans_a=1, trans_b=0) is not built");
    PACED_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0, "paced_gemm_bf16: N and ldc must be multiples of 4");
    PACED_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0, "paced_gemm_bf16: lda/ldb must be multiples of 8 (16-byte rows)");
    PACED_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0,
                    "paced_gemm_bf16: operands must be 16-byte aligned");
    if (!trans_a) PACED_CHECK_ARG(K % 64 == 0, "paced_gemm_bf16: K must be a multiple of 64 for a K-contiguous A");
    if (!trans_b) PACED_CHECK_ARG(K % 64 == 0, "paced_gemm_bf16: K must be a multiple of 64 for a K-contiguous B");
    if (trans_a && !a_heads) PACED_CHECK_ARG((size_t)K * lda * 2 < 0xFFFFFFF0ull && lda >= ((M + 7) & ~7),
                                 "paced_gemm_bf16: transposed A too large for 32-bit offsets, or lda below M rounded up to a multiple of 8");
    if (trans_b) PACED_CHECK_ARG((size_t)K * ldb * 2 < 0xFFFFFFF0ull && N % 8 == 0, "paced_gemm_bf16: transposed B too large for 32-bit offsets or N%%8");
    if (!trans_a) PACED_CHECK_ARG((size_t)128 * lda * 2 < 0x7FFFFFFFull, "paced_gemm_bf16: lda too large");
    if (!trans_b) PACED_CHECK_ARG((size_t)128 * ldb * 2 < 0x7FFFFFFFull, "paced_gemm_bf16: ldb too large");
    if (epilogue == EPI_RES_DROP || epilogue == EPI_MASK)
        PACED_CHECK_ARG(aux && ldaux % 4 == 0 && ((uintptr_t)aux & 7) == 0, "paced_gemm_bf16: epilogue %d needs an 8-byte aligned aux with ldaux%%4==0", epilogue);
    if (epilogue == EPI_MASK_BITS || (epilogue == EPI_RELU_DROP && aux))
        PACED_CHECK_ARG(aux && N % 8 == 0 && (ldaux >= N / 8 || (ldaux == -1 && batch == 1 && ((uintptr_t)aux & 15) == 0 && K >= 2 * BK)),
                        "paced_gemm_bf16: the bit matrix needs N %% 8 == 0 and ldaux >= N/8 bytes per row, or ldaux = -1 (tile order, 16-byte aligned, K >= 128)");
    PACED_CHECK_ARG(splits >= 0, "paced_gemm_bf16: splits must be >= 0 (0 = choose automatically)");
    PACED_CHECK_ARG(splits <= 1 || epilogue == EPI_ATOMIC_F32, "paced_gemm_bf16: split contraction needs the atomic epilogue");
    if (epilogue != EPI_ATOMIC_F32) splits = 1;

    GemmParams p;
    p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C; p.aux = (const bf16*)aux;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
    p.k_per_split = K; p.a_heads = a_heads; p.b_heads = b_heads; p.c_heads = c_heads;
    p.alpha = alpha; p.mask_scale = mask_scale;
    p.drop = make_drop(drop_p, drop_seed);
    p.drop_narrow = (unsigned long long)M * (unsigned long long)N <= (1ull << 34);
    p.tk_cand = nullptr; p.tk_cnt = nullptr; p.tk_thr = nullptr; p.tk_cap = 0; p.tk_row_base = 0;
    p.batch = batch; p.strideA = strideA; p.strideB = strideB; p.strideC = strideC;
    return paced_gemm_launch(p, trans_a != 0, trans_b != 0, epilogue, splits, (hipStream_t)stream);
}

extern "C" int paced_gemm_bf16(const void* A, const void* B, void* C, const void* aux,
                               int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
                               int trans_a, int trans_b, int epilogue, float alpha, float mask_scale,
                               float drop_p, uint32_t drop_seed, int splits, void* stream) {
    return gemm_entry(A, B, C, aux, M, N, K, lda, ldb, ldc, ldaux, trans_a, trans_b, epilogue, alpha, mask_scale,
                      drop_p, drop_seed, splits, stream, 1, 0, 0, 0);
}

extern "C" int paced_gemm_bf16_heads(const void* A, const void* B, void* C, const void* aux,
                                     int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
                                     int trans_a, int trans_b, int epilogue, float alpha, float mask_scale,
                                     float drop_p, uint32_t drop_seed, int splits,
                                     long long a_heads, long long b_heads, long long c_heads, void* stream) {
    return gemm_entry(A, B, C, aux, M, N, K, lda, ldb, ldc, ldaux, trans_a, trans_b, epilogue, alpha, mask_scale,
                      drop_p, drop_seed, splits, stream, 1, 0, 0, 0, a_heads, b_heads, c_heads);
}

// Mirrors `struct paced_gemm_opts` of include/paced_hip.h.
struct paced_gemm_opts { int grid_limit, tile, ring, kernel; };

extern "C" int paced_gemm_bf16_ex(const void* A, const void* B, void* C, const void* aux,
                                  int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
                                  int trans_a, int trans_b, int epilogue, float alpha, float mask_scale,
                                  float drop_p, uint32_t drop_seed, int splits,
                                  long long a_heads, long long b_heads, long long c_heads,
                                  int batch, long long strideA, long long strideB, long long strideC,
                                  paced_gemm_opts* opts, void* stream) {
    if (opts) {
        PACED_CHECK_ARG(opts->tile < 0 || opts->tile == 0 || opts->tile == 64 || opts->tile == 128 || opts->tile == 256,
                        "paced_gemm_bf16_ex: opts.tile must be negative (process default), 0 (heuristic), 64, 128 or 256");
        t_call.grid_limit = opts->grid_limit; t_call.tile = opts->tile; t_call.ring = opts->ring;
    }
    const int rc = gemm_entry(A, B, C, aux, M, N, K, lda, ldb, ldc, ldaux, trans_a, trans_b, epilogue, alpha, mask_scale,
                              drop_p, drop_seed, splits, stream, batch, strideA, strideB, strideC, a_heads, b_heads, c_heads);
    t_call = CallOpts();
    if (opts) opts->kernel = rc == PACED_OK ? t_last_kernel : 0;
    return rc;
}

extern "C" int paced_gemm_bf16_batched(const void* A, const void* B, void* C,
                                       int M, int N, int K, int lda, int ldb, int ldc,
                                       int trans_a, int trans_b, int epilogue, float alpha, int splits,
                                       int batch, long long strideA, long long strideB, long long strideC, void* stream) {
    return gemm_entry(A, B, C, nullptr, M, N, K, lda, ldb, ldc, 0, trans_a, trans_b, epilogue, alpha, 1.f,
                      0.f, 0, splits, stream, batch, strideA, strideB, strideC);
}

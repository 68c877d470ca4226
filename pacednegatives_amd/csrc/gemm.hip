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

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;   // 16 KiB per operand tile

// Per-lane source offset (bytes, relative to the tile's first row / column) of the 16-byte chunk
// that LDS-DMA instruction `i` of wave `w` writes at lane-linear position g = (i*4+w)*64 + lane.
template <bool TRANS>
__device__ __forceinline__ uint32_t stage_offset(int g, int ld) {
    if (!TRANS) {
        const int r = g >> 3, cs = g & 7;             // 128-byte rows, 8 chunks
        const int c = cs ^ ((r >> 1) & 7);
        return (uint32_t)(r * ld + c * 8) * 2u;
    } else {
        const int k = g >> 4, cs = g & 15;            // 256-byte rows, 16 chunks (8 units of 32 B)
        const int hk = (k & 3) | (((k >> 3) & 1) << 2);
        const int c = (((cs >> 1) ^ hk) << 1) | (cs & 1);
        return (uint32_t)(k * ld + c * 8) * 2u;
    }
}

// 16 rows x 32 k fragment (one v_mfma_f32_16x16x32_bf16 operand) for rows rb..rb+15, k-step kk.
template <bool TRANS>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int rb, int kk, int lane) {
    if (!TRANS) {
        const int row = rb + (lane & 15);
        const int c = kk * 4 + (lane >> 4);
        const int addr = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
        return *reinterpret_cast<const bf16x8*>(tile + addr);
    } else {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int u = rb >> 4;
        const int hk = q | ((g & 1) << 2);
        const int k0 = kk * 32 + 8 * g + q;
        const int addr0 = k0 * 256 + ((u ^ hk) << 5) + p * 8;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + addr0));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + addr0 + 4 * 256));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <bool TA, bool TB, int EPI>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lA = smem;
    char* lB = smem + TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;

    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int t = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    // Buffer descriptors: base at the tile's first row (or column), range = exactly the valid bytes,
    // so rows >= M/N and contraction rows >= K read as zero.
    const int rowsA = min(BM, p.M - m0), rowsB = min(BN, p.N - n0);
    const bf16* baseA = TA ? p.A + m0 : p.A + (size_t)m0 * p.lda;
    const bf16* baseB = TB ? p.B + n0 : p.B + (size_t)n0 * p.ldb;
    const uint32_t recA = TA ? (uint32_t)(((size_t)(p.K - 1) * p.lda + rowsA) * 2) : (uint32_t)(((size_t)(rowsA - 1) * p.lda + p.K) * 2);
    const uint32_t recB = TB ? (uint32_t)(((size_t)(p.K - 1) * p.ldb + rowsB) * 2) : (uint32_t)(((size_t)(rowsB - 1) * p.ldb + p.K) * 2);
    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, recA, 0x00020000);
    __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)baseB, 0, recB, 0x00020000);

    uint32_t offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = (i * 4 + w) * 64 + lane;
        offA[i] = stage_offset<TA>(g, p.lda);
        offB[i] = stage_offset<TB>(g, p.ldb);
    }
    const uint32_t kstrideA = TA ? (uint32_t)p.lda * 2u : 2u;
    const uint32_t kstrideB = TB ? (uint32_t)p.ldb * 2u : 2u;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const uint32_t sA = (uint32_t)k0 * kstrideA, sB = (uint32_t)k0 * kstrideB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(lA + (i * 4 + w) * 1024), 16, offA[i], sA, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LDS_PTR(lB + (i * 4 + w) * 1024), 16, offB[i], sB, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = read_frag<TA>(lA, wm * 64 + i * 16, kk, lane);
                fb[i] = read_frag<TB>(lB, wn * 64 + i * 16, kk, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (EPI == EPI_ATOMIC_F32 || EPI == EPI_TOPK_DENSE || EPI == EPI_TOPK_FILTER)   // natural: lane -> n, registers -> m
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    else                          // swapped: lane -> m, registers -> 4 consecutive n
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // ---- epilogue
    if (EPI == EPI_TOPK_DENSE || EPI == EPI_TOPK_FILTER) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane & 15);
            if (n >= p.N) continue;
            const float thr = (EPI == EPI_TOPK_FILTER) ? p.tk_thr[n] : 0.f;
            unsigned long long* cand = p.tk_cand + (size_t)n * p.tk_cap;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
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
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                    if (m < p.M && n < p.N) atomicAdd(C + (size_t)m * p.ldc + n, acc[i][j][r] * p.alpha);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;          // N is a multiple of 4 (checked on the host)
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha;
            const size_t ci = (size_t)m * p.ldc + n;
            const uint64_t di = (uint64_t)m * (uint64_t)p.N + n;   // dropout counter: logical (row, col)
            if (EPI == EPI_F32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + ci) = f32x4{v[0], v[1], v[2], v[3]};
                continue;
            }
            if (EPI == EPI_RES_DROP || EPI == EPI_MASK) {
                const bf16x4 a = *reinterpret_cast<const bf16x4*>(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (EPI == EPI_RES_DROP) v[r] = bf2f(a[r]) + v[r] * drop_factor(p.drop, di + r);
                    else v[r] = bf2f(a[r]) > 0.f ? v[r] * p.mask_scale : 0.f;
                }
            } else if (EPI == EPI_RELU_DROP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f) * drop_factor(p.drop, di + r);
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + ci) = o;
        }
    }
}

template <bool TA, bool TB, int EPI>
int launch(const GemmParams& p, int splits, hipStream_t stream) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    dim3 grid(ntm * ntn, splits);
    hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, EPI>), grid, dim3(256), 2 * TILE_BYTES, stream, p);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
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
    PACED_CHECK_ARG(splits >= 1, "paced_gemm_bf16: splits must be >= 1");
    PACED_CHECK_ARG(splits == 1 || epilogue == EPI_ATOMIC_F32, "paced_gemm_bf16: split contraction needs the atomic epilogue");

    GemmParams p;
    p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C; p.aux = (const bf16*)aux;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
    const int ktiles = (K + BK - 1) / BK;
    const int tiles_per_split = (ktiles + splits - 1) / splits;
    p.k_per_split = tiles_per_split * BK;
    splits = (ktiles + tiles_per_split - 1) / tiles_per_split;
    p.alpha = alpha; p.mask_scale = mask_scale;
    p.drop = make_drop(drop_p, drop_seed);
    p.tk_cand = nullptr; p.tk_cnt = nullptr; p.tk_thr = nullptr; p.tk_cap = 0; p.tk_row_base = 0;
    return paced_gemm_launch(p, trans_a != 0, trans_b != 0, epilogue, splits, (hipStream_t)stream);
}

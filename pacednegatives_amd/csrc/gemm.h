// Internal interface of the tiled bf16 MFMA contraction kernel (gemm.hip).
#pragma once
#include "common.h"

enum Epi : int {
    EPI_BF16 = 0,        // C = acc
    EPI_RES_DROP = 1,    // C = aux + dropout(acc)
    EPI_RELU_DROP = 2,   // C = dropout(relu(acc)); if aux: bit (m, n) of the byte matrix aux = (C > 0)
    EPI_MASK = 3,        // C = acc * (aux > 0 ? mask_scale : 0)
    EPI_F32 = 4,         // C(f32) = acc
    EPI_ATOMIC_F32 = 5,  // C(f32) += acc   (atomic, split-K allowed)
    EPI_MASK_BITS = 6,   // C = acc * (bit (m, n) of the byte matrix aux ? mask_scale : 0)
    EPI_COUNT,           // ---- epilogues below are internal to the pool scorer (mining.hip)
    EPI_TOPK_DENSE = 7,  // cand[n][m] = key(acc, row_base + m)
    EPI_TOPK_FILTER = 8  // if acc >= thr[n]: cand[n][cnt[n]++] = key(acc, row_base + m)
};

struct GemmParams {
    const bf16* A;
    const bf16* B;
    void* C;
    const bf16* aux;
    int M, N, K;
    int lda, ldb, ldc, ldaux;
    int k_per_split;   // multiple of BK
    // Head-major operands (> 0: elements between heads; the operand's leading dimension is then 64): the operand's feature
    // columns [64h, 64h + 64) are the contiguous [rows, 64] matrix at base + h * stride instead of a column block of rows of
    // ld elements.  A: features = K (not transposed) or M (transposed); B: transposed only, features = N; C: plain bf16
    // epilogue only, features = N.
    long long a_heads, b_heads, c_heads;
    int batch;         // independent problems of the same shape; operand b starts strideX * b elements after operand 0
    long long strideA, strideB, strideC;
    float alpha;       // scales acc before the epilogue
    float mask_scale;
    DropCfg drop;
    int drop_narrow;   // the output has at most 2^34 elements: dropout's quad numbers fit 32 bits
    // pool scorer only
    unsigned long long* tk_cand;
    unsigned int* tk_cnt;
    const float* tk_thr;
    int tk_cap;
    unsigned int tk_row_base;
};

// Order-preserving 64-bit key: larger key <=> (higher score, then lower pool row).
__host__ __device__ __forceinline__ unsigned long long topk_key(float score, unsigned int row) {
    score += 0.0f;                                   // -0 -> +0
    unsigned int u = __builtin_bit_cast(unsigned int, score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(~row);
}
__host__ __device__ __forceinline__ float topk_key_score(unsigned long long key) {
    unsigned int u = (unsigned int)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, u);
}
__host__ __device__ __forceinline__ unsigned int topk_key_row(unsigned long long key) { return ~(unsigned int)key; }

// Launches the tiled kernel for (trans_a, trans_b, epilogue); arguments are already validated.
int paced_gemm_launch(const GemmParams& p, bool trans_a, bool trans_b, int epilogue, int splits, hipStream_t stream);



// Pool scoring for at most 128 queries with the HBM-streaming kernel (EPI_TOPK_FILTER's contract: p.A pool rows [M, K],
// p.B queries [N, K], p.tk_*); PACED_ERR_ARG when the shape is outside what it covers (the caller then takes the tiled kernel).
int paced_pool_stream_launch(const GemmParams& p, hipStream_t stream);

// Shared device helpers for the gfx950 kernels of the lceT5 hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---------------------------------------------------------------------------------------------
// Error reporting shared by every extern "C" entry point (defined in api.hip).
extern "C" const char* paced_last_error(void);
void paced_set_error(const char* fmt, ...);
#define PACED_OK 0
#define PACED_ERR_ARG (-1)
#define PACED_ERR_HIP (-2)
#define PACED_CHECK_ARG(cond, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            paced_set_error(__VA_ARGS__);     \
            return PACED_ERR_ARG;             \
        }                                     \
    } while (0)
#define PACED_CHECK_LAUNCH()                                               \
    do {                                                                   \
        hipError_t e_ = hipGetLastError();                                 \
        if (e_ != hipSuccess) {                                            \
            paced_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return PACED_ERR_HIP;                                          \
        }                                                                  \
    } while (0)

// ---------------------------------------------------------------------------------------------
// bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }   // v_cvt_pk_bf16_f32: RNE, NaN kept

// ---------------------------------------------------------------------------------------------
// Counter-based dropout.  One hash evaluation serves FOUR neighbouring elements: element idx takes the
// (idx & 3)-th 16-bit field of a 64-bit value derived from the quad number idx >> 2, and is KEPT when
// its field >= thresh16 = round(p * 65536); so the realised rate is p quantised to 1/65536, and kept
// values are scaled by 1/(1 - thresh16/65536).  Forward and backward kernels recompute the same
// decision from the same (seed, index), so no mask is ever stored.
__host__ __device__ __forceinline__ uint32_t paced_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// the two 32-bit words (fields 0,1 and fields 2,3) of quad number `quad`
__host__ __device__ __forceinline__ void paced_drop_words(uint32_t seed, uint64_t quad, uint32_t& h0, uint32_t& h1) {
    const uint32_t hi = (uint32_t)(quad >> 32) & 0xffffffu;
    h0 = paced_mix32(((uint32_t)quad ^ seed) + hi * 0x9e3779u);     // 24 x 24-bit product: a full-rate multiply
    h1 = (h0 ^ (h0 >> 13)) * 0xc2b2ae35u;
    h1 ^= h1 >> 15;
}
__host__ __device__ __forceinline__ uint32_t paced_drop_bits(uint32_t seed, uint64_t idx) {
    uint32_t h0, h1;
    paced_drop_words(seed, idx >> 2, h0, h1);
    const uint32_t h = (idx & 2) ? h1 : h0;
    return (idx & 1) ? (h >> 16) : (h & 0xffffu);
}
struct DropCfg {
    uint32_t seed;
    uint32_t thresh;   // 16-bit threshold; 0 => dropout disabled
    float scale;       // 1/(1-p) for the quantised p
};
__host__ __forceinline__ DropCfg make_drop(float p, uint32_t seed) {
    DropCfg d;
    d.seed = seed;
    if (p <= 0.f) { d.thresh = 0; d.scale = 1.f; return d; }
    double t = (double)p * 65536.0 + 0.5;
    d.thresh = t >= 65535.0 ? 65535u : (uint32_t)t;
    if (d.thresh == 0) d.thresh = 1;
    d.scale = (float)(1.0 / (1.0 - (double)d.thresh / 65536.0));
    return d;
}
__device__ __forceinline__ float drop_factor(const DropCfg& d, uint64_t idx) {
    if (d.thresh == 0) return 1.f;
    return paced_drop_bits(d.seed, idx) >= d.thresh ? d.scale : 0.f;
}
// factors of the aligned quad (idx4 a multiple of 4 .. idx4 + 3) from ONE hash evaluation
__device__ __forceinline__ void drop_factor4(const DropCfg& d, uint64_t idx4, float (&f)[4]) {
    if (d.thresh == 0) { f[0] = f[1] = f[2] = f[3] = 1.f; return; }
    uint32_t h0, h1;
    paced_drop_words(d.seed, idx4 >> 2, h0, h1);
    f[0] = (h0 & 0xffffu) >= d.thresh ? d.scale : 0.f;
    f[1] = (h0 >> 16) >= d.thresh ? d.scale : 0.f;
    f[2] = (h1 & 0xffffu) >= d.thresh ? d.scale : 0.f;
    f[3] = (h1 >> 16) >= d.thresh ? d.scale : 0.f;
}
// same with dropout known to be on (no uniform branch per call)
__device__ __forceinline__ void drop_factor4_on(const DropCfg& d, uint64_t idx4, float (&f)[4]) {
    uint32_t h0, h1;
    paced_drop_words(d.seed, idx4 >> 2, h0, h1);
    f[0] = (h0 & 0xffffu) >= d.thresh ? d.scale : 0.f;
    f[1] = (h0 >> 16) >= d.thresh ? d.scale : 0.f;
    f[2] = (h1 & 0xffffu) >= d.thresh ? d.scale : 0.f;
    f[3] = (h1 >> 16) >= d.thresh ? d.scale : 0.f;
}
// keep decisions only (dropout known to be on): for callers that apply the 1/(1-p) scale elsewhere
__device__ __forceinline__ void drop_keep4_on(const DropCfg& d, uint64_t idx4, bool (&k)[4]) {
    uint32_t h0, h1;
    paced_drop_words(d.seed, idx4 >> 2, h0, h1);
    k[0] = (h0 & 0xffffu) >= d.thresh;
    k[1] = (h0 >> 16) >= d.thresh;
    k[2] = (h1 & 0xffffu) >= d.thresh;
    k[3] = (h1 >> 16) >= d.thresh;
}
// factors of 8 consecutive elements starting at a multiple of 4
__device__ __forceinline__ void drop_factor8(const DropCfg& d, uint64_t idx4, float (&f)[8]) {
    float a[4], b[4];
    drop_factor4(d, idx4, a);
    drop_factor4(d, idx4 + 4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware remap of a linear workgroup id: workgroups are dealt round-robin over the 8
// XCDs, so ids with equal (id % 8) share an L2; give each of those groups a contiguous range.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + i;
}

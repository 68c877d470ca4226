// Shared device helpers for the gfx950 kernels of the lceT5 hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
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
// Counter-based dropout.  One 32-bit hash serves TWO neighbouring elements (indices 2i and 2i+1 take
// its low and high 16 bits); an element is KEPT when its 16-bit value >= thresh16 = round(p * 65536),
// so the realised rate is p quantised to 1/65536, and kept values are scaled by 1/(1 - thresh16/65536).
// Forward and backward kernels recompute the same decision from the same (seed, index), so no mask
// is ever stored.
__host__ __device__ __forceinline__ uint32_t paced_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t paced_drop_bits(uint32_t seed, uint64_t idx) {
    const uint32_t pair = (uint32_t)(idx >> 1), hi = (uint32_t)(idx >> 33);
    const uint32_t h = paced_mix32((pair ^ seed) + hi * 0x9e3779b9u);
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
// factors of the aligned pair (idx even, idx+1) from ONE hash
__device__ __forceinline__ void drop_factor2(const DropCfg& d, uint64_t idx_even, float& f0, float& f1) {
    if (d.thresh == 0) { f0 = 1.f; f1 = 1.f; return; }
    const uint32_t pair = (uint32_t)(idx_even >> 1), hi = (uint32_t)(idx_even >> 33);
    const uint32_t h = paced_mix32((pair ^ d.seed) + hi * 0x9e3779b9u);
    f0 = (h & 0xffffu) >= d.thresh ? d.scale : 0.f;
    f1 = (h >> 16) >= d.thresh ? d.scale : 0.f;
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

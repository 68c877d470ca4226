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
// Counter-based dropout.  One 32-bit hash of (seed, 64-bit element index); an element is KEPT when
// hash >= thresh where thresh = p * 2^32, and kept values are scaled by 1/(1-p).  Forward and
// backward kernels recompute the same decision from the same (seed, index), so no mask is stored.
__host__ __device__ __forceinline__ uint32_t paced_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t paced_drop_hash(uint32_t seed, uint64_t idx) {
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t h = paced_mix32(lo ^ seed);
    h = paced_mix32(h + 0x9e3779b9u * (hi + 1u) + (seed << 7) + (seed >> 3));
    return h;
}
struct DropCfg {
    uint32_t seed;
    uint32_t thresh;   // 0 => dropout disabled
    float scale;       // 1/(1-p)
};
__host__ __forceinline__ DropCfg make_drop(float p, uint32_t seed) {
    DropCfg d;
    d.seed = seed;
    if (p <= 0.f) { d.thresh = 0; d.scale = 1.f; return d; }
    double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    d.scale = 1.f / (1.f - p);
    return d;
}
__device__ __forceinline__ float drop_factor(const DropCfg& d, uint64_t idx) {
    if (d.thresh == 0) return 1.f;
    return paced_drop_hash(d.seed, idx) >= d.thresh ? d.scale : 0.f;
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

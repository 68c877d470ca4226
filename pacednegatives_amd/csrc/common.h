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

// hipFuncSetAttribute acts on the CURRENT device, so "already done" is kept per device (a process that drives a second
// GPU would otherwise launch there without the attribute).  Racing threads at worst set the attribute twice.
struct PerDeviceOnce {
    bool done[64] = {};
    bool& flag() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
        return done[dev];
    }
};

// compute units of the current device (256 when it cannot be asked)
inline int paced_cu_count() {
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

// ---------------------------------------------------------------------------------------------
// bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }   // v_cvt_pk_bf16_f32: RNE, NaN kept

// ---------------------------------------------------------------------------------------------
// Counter-based dropout.  One hash state serves FOUR neighbouring elements: element idx takes the
// (idx & 3)-th of four 32-bit values derived from the quad number idx >> 2 and is KEPT when the top
// 16 bits of its value are >= thresh16 = round(p * 65536), i.e. when value >= thresh16 << 16; so the
// realised rate is p quantised to 1/65536, and kept values are scaled by 1/(1 - thresh16/65536).
// Forward and backward kernels recompute the same decision from the same (seed, index), so no mask
// is ever stored.  Every multiply is 24 x 24 bits (v_mad_u32_u24, a full-rate instruction; a 32-bit
// integer multiply issues at a quarter of that): two multiply / xor-shift rounds make the state, one
// multiply-add per element makes its value.  tools/dropout_hash_stats.py holds the statistics the
// constants were chosen by (keep rates, pair correlations at distances 1 .. 2^19 and between the four
// values, dependence on index bits, neighbouring seeds, byte histograms).
__host__ __device__ __forceinline__ uint32_t paced_mad24(uint32_t a, uint32_t c, uint32_t add) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, c) + add;
#else
    return (a & 0xffffffu) * (c & 0xffffffu) + add;
#endif
}
// the state shared by the four elements of quad number `quad`
__host__ __device__ __forceinline__ uint32_t paced_drop_mix(uint32_t x) {
    x ^= x >> 16; x = paced_mad24(x, 0xeb352du, x);
    x ^= x >> 13; x = paced_mad24(x, 0x2c1b3du, x);
    x ^= x >> 12;
    return x;
}
__host__ __device__ __forceinline__ uint32_t paced_drop_state(uint32_t seed, uint64_t quad) {
    const uint32_t hi = (uint32_t)(quad >> 32) & 0xffffffu;
    return paced_drop_mix(paced_mad24(hi, 0x9e3779u, (uint32_t)quad ^ seed));
}
// the same state for a quad number below 2^32 (a tensor of fewer than 2^34 elements): its high word contributes nothing
__host__ __device__ __forceinline__ uint32_t paced_drop_state32(uint32_t seed, uint32_t quad) {
    return paced_drop_mix(quad ^ seed);
}
// value of element e (0..3) of a quad with state y: kept iff value >= DropCfg::limit
template <int E> __host__ __device__ __forceinline__ uint32_t paced_drop_value(uint32_t y) {
    constexpr uint32_t C = E == 0 ? 0x1b873bu : E == 1 ? 0x6ca68bu : E == 2 ? 0xb2ae35u : 0xd4e12du;
    return paced_mad24(y, C, y);
}
__host__ __device__ __forceinline__ uint32_t paced_drop_multiplier(int e) {
    return e == 0 ? 0x1b873bu : e == 1 ? 0x6ca68bu : e == 2 ? 0xb2ae35u : 0xd4e12du;
}
__host__ __device__ __forceinline__ uint32_t paced_drop_bits(uint32_t seed, uint64_t idx) {   // 16-bit field of one element
    const uint32_t y = paced_drop_state(seed, idx >> 2);
    return paced_mad24(y, paced_drop_multiplier((int)(idx & 3)), y) >> 16;
}
struct DropCfg {
    uint32_t seed;
    uint32_t thresh;   // 16-bit threshold; 0 => dropout disabled
    uint32_t limit;    // thresh << 16: an element is kept iff its 32-bit value >= limit
    float scale;       // 1/(1-p) for the quantised p
};
__host__ __forceinline__ DropCfg make_drop(float p, uint32_t seed) {
    DropCfg d;
    d.seed = seed;
    if (p <= 0.f) { d.thresh = 0; d.limit = 0; d.scale = 1.f; return d; }
    double t = (double)p * 65536.0 + 0.5;
    d.thresh = t >= 65535.0 ? 65535u : (uint32_t)t;
    if (d.thresh == 0) d.thresh = 1;
    d.limit = d.thresh << 16;
    d.scale = (float)(1.0 / (1.0 - (double)d.thresh / 65536.0));
    return d;
}
__device__ __forceinline__ float drop_factor(const DropCfg& d, uint64_t idx) {
    if (d.thresh == 0) return 1.f;
    return paced_drop_bits(d.seed, idx) >= d.thresh ? d.scale : 0.f;
}
// keep decisions of the aligned quad (idx4 a multiple of 4 .. idx4 + 3) from ONE state (dropout known to be on)
// ... from the quad's state
__device__ __forceinline__ void drop_keep4_of_state(const DropCfg& d, uint32_t y, bool (&k)[4]) {
    k[0] = paced_drop_value<0>(y) >= d.limit;
    k[1] = paced_drop_value<1>(y) >= d.limit;
    k[2] = paced_drop_value<2>(y) >= d.limit;
    k[3] = paced_drop_value<3>(y) >= d.limit;
}
__device__ __forceinline__ void drop_keep4_on(const DropCfg& d, uint64_t idx4, bool (&k)[4]) {
    drop_keep4_of_state(d, paced_drop_state(d.seed, idx4 >> 2), k);
}
// factors of the aligned quad, dropout known to be on (no uniform branch per call)
__device__ __forceinline__ void drop_factor4_on(const DropCfg& d, uint64_t idx4, float (&f)[4]) {
    bool k[4];
    drop_keep4_on(d, idx4, k);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = k[e] ? d.scale : 0.f;
}
// factors of quad number `quad` < 2^32, dropout known to be on
__device__ __forceinline__ void drop_factor4_on32(const DropCfg& d, uint32_t quad, float (&f)[4]) {
    const uint32_t y = paced_drop_state32(d.seed, quad);
    f[0] = paced_drop_value<0>(y) >= d.limit ? d.scale : 0.f;
    f[1] = paced_drop_value<1>(y) >= d.limit ? d.scale : 0.f;
    f[2] = paced_drop_value<2>(y) >= d.limit ? d.scale : 0.f;
    f[3] = paced_drop_value<3>(y) >= d.limit ? d.scale : 0.f;
}
__device__ __forceinline__ void drop_factor4(const DropCfg& d, uint64_t idx4, float (&f)[4]) {
    if (d.thresh == 0) { f[0] = f[1] = f[2] = f[3] = 1.f; return; }
    drop_factor4_on(d, idx4, f);
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

// Memory-bound satellites of the T5 step: RMSNorm, embedding, dropout, cross-entropy, the LCE
// loss grouping and difficulty weights, relative-position-bias tables, AdamW.  One wave per row
// for the row-wise ops, 16-byte vector accesses throughout.
#include "common.h"

namespace {

constexpr int ROW_BLOCK = 256;                 // 4 waves, one row per wave per iteration
constexpr int MAX_ROW_GRID = 4096;

__host__ int row_grid(int rows) {
    int g = (rows + 3) / 4;
    return g < 1 ? 1 : (g > MAX_ROW_GRID ? MAX_ROW_GRID : g);
}

// ------------------------------------------------------------------------------------------ RMSNorm
// y = bf16(x * rsqrt(mean(x^2) + eps) * w * out_scale) [* dropout]; rstd saved in fp32.
// Streamed operands (read or written once, far larger than the caches): NORM_NT bit 0 = non-temporal loads, bit 1 =
// non-temporal stores.
#ifndef NORM_NT
#define NORM_NT 3           // in the training step: -0.3 .. -0.6 ms against 0 (the streams stop evicting what other kernels re-read)
#endif
template <class T> __device__ __forceinline__ T ld_stream(const T* p) { return (NORM_NT & 1) ? __builtin_nontemporal_load(p) : *p; }
template <class T> __device__ __forceinline__ void st_stream(T* p, T v) { if (NORM_NT & 2) __builtin_nontemporal_store(v, p); else *p = v; }

constexpr int NORM_MAX_CHUNKS = 4;                       // d <= 2048
#ifndef NORM_FWD_NT_STORE
#define NORM_FWD_NT_STORE (NORM_NT & 2)
#endif
// One wave per row; the row is read twice (the second read hits the cache: holding it in registers between the two passes
// measured 10 us slower on 139264 x 768, the loads of the second pass then sit behind the row sum).
__global__ void __launch_bounds__(ROW_BLOCK) rmsnorm_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                               bf16* __restrict__ y, float* __restrict__ rstd,
                                                               int M, int d, float eps, float out_scale, DropCfg drop) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nchunk = d >> 3;
    for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
        const bf16* xr = x + (size_t)row * d;
        float ss = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = bf2f(v[j]); ss += f * f; }
        }
        ss = wave_sum(ss);
        const float rs = rsqrtf(ss / (float)d + eps);
        if (lane == 0 && rstd) rstd[row] = rs;
        for (int c = lane; c < nchunk; c += 64) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c * 8);
            const bf16x8 wv8 = *reinterpret_cast<const bf16x8*>(w + c * 8);
            bf16x8 o;
            float df[8];
            drop_factor8(drop, (uint64_t)row * d + c * 8, df);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(v[j]) * rs * bf2f(wv8[j]) * out_scale * df[j]);
            if (NORM_FWD_NT_STORE) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(y + (size_t)row * d + c * 8));
            else *reinterpret_cast<bf16x8*>(y + (size_t)row * d + c * 8) = o;
        }
    }
}

// Backward of the above, fused with the residual-stream add and (optionally) the dropout mask of
// the sub-layer that precedes this norm in the forward pass:
//   dyo = dy * dropmask(drop)                          (dropout on the norm output, final norms only)
//   g   = dyo * w * out_scale ;  xh = x * rstd
//   dx  = rstd * (g - xh * mean(g * xh)) + dres
//   dw += sum_rows dyo * xh * out_scale                 (fp32 atomics, one per column per workgroup)
//   dx2 = dx * dropmask(drop2)                          (optional second output)
__global__ void __launch_bounds__(ROW_BLOCK) rmsnorm_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                               const bf16* __restrict__ w, const float* __restrict__ rstd,
                                                               const bf16* __restrict__ dres, bf16* __restrict__ dx,
                                                               bf16* __restrict__ dx2, float* __restrict__ dw,
                                                               int M, int d, float out_scale, DropCfg drop, DropCfg drop2) {
    __shared__ float red[4][2048];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nchunk = d >> 3;
    float dwacc[NORM_MAX_CHUNKS][8];
#pragma unroll
    for (int i = 0; i < NORM_MAX_CHUNKS; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) dwacc[i][j] = 0.f;

    for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
        const float rs = rstd[row];
        float g[NORM_MAX_CHUNKS][8], xh[NORM_MAX_CHUNKS][8];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) {
                const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dy + (size_t)row * d + c * 8);
                const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + (size_t)row * d + c * 8);
                const bf16x8 wv8 = *reinterpret_cast<const bf16x8*>(w + c * 8);
                float df[8];
                drop_factor8(drop, (uint64_t)row * d + c * 8, df);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float dyo = bf2f(dv[j]) * df[j];
                    xh[i][j] = bf2f(xv[j]) * rs;
                    g[i][j] = dyo * bf2f(wv8[j]) * out_scale;
                    dwacc[i][j] += dyo * xh[i][j] * out_scale;
                    dot += g[i][j] * xh[i][j];
                }
            }
        }
        dot = wave_sum(dot) / (float)d;
#pragma unroll
        for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) {
                bf16x8 r8;
                if (dres) r8 = *reinterpret_cast<const bf16x8*>(dres + (size_t)row * d + c * 8);
                bf16x8 o, o2;
                float df2[8];
                drop_factor8(drop2, (uint64_t)row * d + c * 8, df2);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = rs * (g[i][j] - xh[i][j] * dot);
                    if (dres) v += bf2f(r8[j]);
                    o[j] = f2bf(v);
                    o2[j] = f2bf(bf2f(o[j]) * df2[j]);
                }
                *reinterpret_cast<bf16x8*>(dx + (size_t)row * d + c * 8) = o;
                if (dx2) *reinterpret_cast<bf16x8*>(dx2 + (size_t)row * d + c * 8) = o2;
            }
        }
    }
    if (!dw) return;
#pragma unroll
    for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wv][c * 8 + j] = dwacc[i][j];
        }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < d; col += ROW_BLOCK) {
        const float s = red[0][col] + red[1][col] + red[2][col] + red[3][col];
        atomicAdd(dw + col, s);
    }
}

// Row-group variant of the backward kernel for d = LPR * PER * 8: LPR lanes share a row, so a wave works on 64 / LPR
// rows at once, every lane carries PER 16-byte pieces of each operand, and all loads of a row group are issued before
// the first use.  With one row per wave (above) d = 768 leaves a third of the lanes idle and too few bytes in flight
// to cover HBM latency: 16-22 % less time on 139264 x 768 (5.4 TB/s with every operand present, cold caches).  The
// forward kernel gains nothing from the same layout.  Same arithmetic per element as above; only the order of the
// row sums differs.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// operands stay packed (two bf16 per register) until the element is used
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t bf_pack(float a, float b) {
    const bf16x2 v = __builtin_convertvector((__attribute__((ext_vector_type(2))) float){a, b}, bf16x2);
    return *reinterpret_cast<const uint32_t*>(&v);
}
// Called inside a loop on loop-invariant packed operands: the compiler would otherwise unpack them to one register per
// element ahead of the loop and spill.
template <int N>
__device__ __forceinline__ void keep_packed(u32x4 (&r)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(r[i][j]));
}

// DROP: dropout on the norm output (the two final norms only)
template <int LPR, int PER, bool DROP>
__global__ void __launch_bounds__(ROW_BLOCK) __attribute__((amdgpu_waves_per_eu(2)))
rmsnorm_bwd_rows_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                                    const bf16* __restrict__ w, const float* __restrict__ rstd,
                                                                    const bf16* __restrict__ dres, bf16* __restrict__ dx,
                                                                    bf16* __restrict__ dx2, float* __restrict__ dw,
                                                                    int M, float out_scale, DropCfg drop, DropCfg drop2) {
    constexpr int d = LPR * PER * 8, RPW = 64 / LPR;
    __shared__ float red[4][d];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ll = lane % LPR, sub = lane / LPR;
    u32x4 wr[PER];
    float dwacc[PER][8];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        wr[i] = *reinterpret_cast<const u32x4*>(w + (ll + i * LPR) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) dwacc[i][j] = 0.f;
    }
    for (int row0 = (blockIdx.x * 4 + wv) * RPW; row0 < M; row0 += gridDim.x * 4 * RPW) {
        const int row = row0 + sub;
        const bool ok = row < M;
        const int rr = ok ? row : M - 1;
        const size_t base = (size_t)rr * d;
        u32x4 dv[PER], xv[PER], rv[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            dv[i] = ld_stream(reinterpret_cast<const u32x4*>(dy + base + (ll + i * LPR) * 8));
            xv[i] = ld_stream(reinterpret_cast<const u32x4*>(x + base + (ll + i * LPR) * 8));
        }
#pragma unroll
        for (int i = 0; i < PER; ++i)
            rv[i] = dres ? ld_stream(reinterpret_cast<const u32x4*>(dres + base + (ll + i * LPR) * 8)) : u32x4{0u, 0u, 0u, 0u};
        const float rs = rstd[rr];
        const float okf = ok ? out_scale : 0.f;
        keep_packed(wr);
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            float df[8];
            if (DROP) drop_factor8(drop, (uint64_t)base + (ll + i * LPR) * 8, df);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t du = dv[i][j >> 1], xu = xv[i][j >> 1], wu = wr[i][j >> 1];
                const float dyo = ((j & 1) ? bf_hi(du) : bf_lo(du)) * (DROP ? df[j] : 1.f);
                const float xh = ((j & 1) ? bf_hi(xu) : bf_lo(xu)) * rs;
                const float g = dyo * ((j & 1) ? bf_hi(wu) : bf_lo(wu)) * out_scale;
                dot += g * xh;
            }
            __builtin_amdgcn_sched_barrier(0);      // one piece at a time: the scheduler otherwise unpacks every piece first and spills
        }
        dot = row_sum<LPR>(dot) / (float)d;
        keep_packed(dv);
        keep_packed(xv);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = ll + i * LPR;
            float df[8], df2[8], v[8];
            if (DROP) drop_factor8(drop, (uint64_t)base + c * 8, df);
            drop_factor8(drop2, (uint64_t)base + c * 8, df2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t du = dv[i][j >> 1], xu = xv[i][j >> 1], wu = wr[i][j >> 1], ru = rv[i][j >> 1];
                const float dyo = ((j & 1) ? bf_hi(du) : bf_lo(du)) * (DROP ? df[j] : 1.f);
                const float xh = ((j & 1) ? bf_hi(xu) : bf_lo(xu)) * rs;
                const float g = dyo * ((j & 1) ? bf_hi(wu) : bf_lo(wu)) * out_scale;
                dwacc[i][j] += dyo * xh * okf;
                v[j] = rs * (g - xh * dot);
                if (dres) v[j] += (j & 1) ? bf_hi(ru) : bf_lo(ru);
            }
            u32x4 o, o2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = bf_pack(v[2 * j], v[2 * j + 1]);
                o2[j] = bf_pack(bf_lo(o[j]) * df2[2 * j], bf_hi(o[j]) * df2[2 * j + 1]);
            }
            if (ok) {
                st_stream(reinterpret_cast<u32x4*>(dx + base + c * 8), o);
                if (dx2) st_stream(reinterpret_cast<u32x4*>(dx2 + base + c * 8), o2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!dw) return;
#pragma unroll
    for (int i = 0; i < PER; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = dwacc[i][j];
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
            if (sub == 0) red[wv][(ll + i * LPR) * 8 + j] = s;
        }
    __syncthreads();
    for (int col = threadIdx.x; col < d; col += ROW_BLOCK) {
        const float s = red[0][col] + red[1][col] + red[2][col] + red[3][col];
        atomicAdd(dw + col, s);
    }
}

// Grid for a grid-stride kernel over `groups` equal work units: as many workgroups as fit on the device at once, then
// trimmed so that every workgroup takes the same number of units (no second, partly filled round).
template <class Kernel>
__host__ int balanced_grid(Kernel kernel, int& slots, int groups, int cap) {
    if (slots == 0) {
        int occ = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, ROW_BLOCK, 0) != hipSuccess || occ <= 0) occ = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = occ * cus;
    }
    const int s = slots < cap ? slots : cap;
    const int iters = (groups + s - 1) / s;
    const int g = (groups + iters - 1) / iters;
    return g < 1 ? 1 : g;
}

template <int LPR, int PER, bool DROP>
__host__ void launch_rmsnorm_bwd_rows_drop(const bf16* dy, const bf16* x, const bf16* w, const float* rstd, const bf16* dres, bf16* dx,
                                           bf16* dx2, float* dw, int M, float out_scale, DropCfg drop, DropCfg drop2, hipStream_t stream) {
    static int slots = 0;
    const int groups = (M + 4 * (64 / LPR) - 1) / (4 * (64 / LPR));
    const int grid = balanced_grid(rmsnorm_bwd_rows_kernel<LPR, PER, DROP>, slots, groups, 2048);   // the cap bounds the per-column atomics on dw
    hipLaunchKernelGGL((rmsnorm_bwd_rows_kernel<LPR, PER, DROP>), dim3(grid), dim3(ROW_BLOCK), 0, stream, dy, x, w, rstd, dres, dx, dx2, dw,
                       M, out_scale, drop, drop2);
}
template <int LPR, int PER>
__host__ void launch_rmsnorm_bwd_rows(const bf16* dy, const bf16* x, const bf16* w, const float* rstd, const bf16* dres, bf16* dx,
                                      bf16* dx2, float* dw, int M, float out_scale, DropCfg drop, DropCfg drop2, hipStream_t stream) {
    if (drop.thresh) launch_rmsnorm_bwd_rows_drop<LPR, PER, true>(dy, x, w, rstd, dres, dx, dx2, dw, M, out_scale, drop, drop2, stream);
    else launch_rmsnorm_bwd_rows_drop<LPR, PER, false>(dy, x, w, rstd, dres, dx, dx2, dw, M, out_scale, drop, drop2, stream);
}

// ---------------------------------------------------------------------------------------- embedding
__global__ void __launch_bounds__(ROW_BLOCK) embedding_fwd_kernel(const long long* __restrict__ ids, const bf16* __restrict__ table,
                                                                 bf16* __restrict__ out, int M, int d, int vocab, DropCfg drop) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nchunk = d >> 3;
    for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
        long long id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);     // ids are validated on the host; clamp keeps the load in range
        for (int c = lane; c < nchunk; c += 64) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(table + (size_t)id * d + c * 8);
            if (drop.thresh) {
                float df[8];
                drop_factor8(drop, (uint64_t)row * d + c * 8, df);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * df[j]);
            }
            *reinterpret_cast<bf16x8*>(out + (size_t)row * d + c * 8) = v;
        }
    }
}

__global__ void __launch_bounds__(ROW_BLOCK) embedding_bwd_kernel(const long long* __restrict__ ids, const bf16* __restrict__ dout,
                                                                 float* __restrict__ dtable, int M, int d, int vocab, DropCfg drop) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
        long long id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        // one dword per lane, 256 contiguous bytes per wave instruction: the shape float atomics run fastest in
        for (int col = lane; col < d; col += 64) {
            const float v = bf2f(dout[(size_t)row * d + col]) * drop_factor(drop, (uint64_t)row * d + col);
            atomicAdd(dtable + (size_t)id * d + col, v);
        }
    }
}

// ------------------------------------------------------------------------------------------ dropout
__global__ void __launch_bounds__(256) dropout_apply_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t n8, DropCfg drop) {
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < n8; c += (size_t)gridDim.x * 256) {
        bf16x8 v = *reinterpret_cast<const bf16x8*>(x + c * 8);
        float df[8];
        drop_factor8(drop, c * 8, df);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * df[j]);
        *reinterpret_cast<bf16x8*>(y + c * 8) = v;
    }
}

__global__ void __launch_bounds__(256) dropout_mask_kernel(float* __restrict__ out, size_t n, DropCfg drop) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = drop_factor(drop, i);
}

// -------------------------------------------------------------------------------------------- casts
__global__ void __launch_bounds__(256) cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, size_t n) {
    const size_t n4 = n >> 2;
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < n4; c += (size_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + c * 4);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
        *reinterpret_cast<bf16x4*>(y + c * 4) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = f2bf(x[n4 * 4 + threadIdx.x]);
}

// ------------------------------------------------------------------------------------ cross-entropy
// One workgroup per logits row (fp32, V columns).  loss = logsumexp(row) - row[label]; rows whose
// label equals ignore_index give loss 0 and a zero gradient.  dlogits (optional, bf16) =
// (softmax - onehot) * gscale.
// exp(a - b) for a <= b where both may be -inf (an empty partial): the empty side contributes nothing
__device__ __forceinline__ float exp_diff(float a, float b) { return a == b ? 1.f : __expf(a - b); }

__device__ __forceinline__ void block_reduce_max_sum(float& m, float& s, float* sh) {
    // combine (max, sum-exp) pairs: wave level then across the 4 waves; threads without elements hold (-inf, 0)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        const float mn = fmaxf(m, m2);
        s = s * exp_diff(m, mn) + s2 * exp_diff(m2, mn);
        m = mn;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[wv * 2] = m; sh[wv * 2 + 1] = s; }
    __syncthreads();
    float M = sh[0], S = sh[1];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const float m2 = sh[i * 2], s2 = sh[i * 2 + 1];
        const float mn = fmaxf(M, m2);
        S = S * exp_diff(M, mn) + s2 * exp_diff(m2, mn);
        M = mn;
    }
    m = M; s = S;
    __syncthreads();
}

__global__ void __launch_bounds__(256) ce_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                float* __restrict__ loss, bf16* __restrict__ dlogits,
                                                int V, int ldl, long long ignore_index, float gscale) {
    __shared__ float sh[8];
    const int row = blockIdx.x;
    const float* lr = logits + (size_t)row * ldl;
    const long long label = labels[row];
    float m = -INFINITY, s = 0.f;
    const int n4 = V >> 2;
    for (int c = threadIdx.x; c < n4; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lr + c * 4);
        const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        const float mn = fmaxf(m, mx);
        s = s * __expf(m - mn) + __expf(v[0] - mn) + __expf(v[1] - mn) + __expf(v[2] - mn) + __expf(v[3] - mn);
        m = mn;
    }
    block_reduce_max_sum(m, s, sh);
    const float lse = m + __logf(s);
    const bool ignored = (label == ignore_index) || label < 0 || label >= V;
    if (threadIdx.x == 0) loss[row] = ignored ? 0.f : lse - lr[label];
    if (!dlogits) return;
    bf16* dr = dlogits + (size_t)row * ldl;
    const float g = ignored ? 0.f : gscale;
    for (int c = threadIdx.x; c < n4; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lr + c * 4);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float p = __expf(v[j] - lse);
            if ((long long)(c * 4 + j) == label) p -= 1.f;
            o[j] = f2bf(p * g);
        }
        *reinterpret_cast<bf16x4*>(dr + c * 4) = o;
    }
}

// ----------------------------------------------------------------------- LCE grouping and weights
// ce[r] = pce[r] + (sum | mean) nce[r*n .. (r+1)*n)        r in [0, R)
__global__ void lce_group_kernel(const float* __restrict__ pce, const float* __restrict__ nce, float* __restrict__ ce,
                                 int R, int n, int use_mean) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float s = 0.f;
    for (int j = 0; j < n; ++j) s += nce[(size_t)r * n + j];
    if (use_mean) s /= (float)n;
    ce[r] = pce[r] + s;
}

// w_i = loss_i/eta if loss_i <= eta else 0 ; stats = {mean w, mean w*loss, d mean(w*loss)/d eta, mean loss}
__global__ void __launch_bounds__(256) lce_weights_kernel(const float* __restrict__ loss, const float* __restrict__ eta_p,
                                                         float* __restrict__ w, float* __restrict__ stats, int R) {
    __shared__ float sh[4][4];
    const float eta = eta_p[0];
    float sw = 0.f, swl = 0.f, sg = 0.f, sl = 0.f;
    for (int i = threadIdx.x; i < R; i += 256) {
        const float l = loss[i];
        const bool hard = l > eta;
        const float wi = hard ? 0.f : l / eta;
        w[i] = wi;
        sw += wi; swl += wi * l; sl += l;
        if (!hard) sg -= l * l / (eta * eta);
    }
    sw = wave_sum(sw); swl = wave_sum(swl); sg = wave_sum(sg); sl = wave_sum(sl);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[wv][0] = sw; sh[wv][1] = swl; sh[wv][2] = sg; sh[wv][3] = sl; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float t = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        stats[threadIdx.x] = t / (float)R;
    }
}

// Threshold weights of pacednegatives/weights.py with the slope in eta next to them:
//   form 0 (LCEWeights, :47-78):  w = l <= eta ? l / eta     : l * mask * eta
//   form 1 (EtaWeights, :10-45):  w = l <= eta ? 1 - l / eta : l * mask * eta
__device__ __forceinline__ void eta_weight(float l, float eta, float mask, int form, float& w, float& dw) {
    if (l > eta) { w = l * mask * eta; dw = l * mask; }
    else if (form == 0) { w = l / eta; dw = -l / (eta * eta); }
    else { w = 1.f - l / eta; dw = l / (eta * eta); }
}
__global__ void eta_weights_kernel(const float* __restrict__ loss, const float* __restrict__ eta_p, float mask, int form,
                                   float* __restrict__ w, float* __restrict__ dw, int R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    float wi, di;
    eta_weight(loss[i], eta_p[0], mask, form, wi, di);
    w[i] = wi;
    if (dw) dw[i] = di;
}
// Meta objective of the eta-bound schedule (old/eta_bound.py:73-78): ce = (pce + nce) / 2, v = EtaWeights(ce),
// J = mean(pce v) + mean(nce v) - sum(v).  out = { J, dJ/d eta, mean(v), mean(ce < eta) }.
__global__ void __launch_bounds__(256) eta_bound_meta_kernel(const float* __restrict__ pce, const float* __restrict__ nce,
                                                            const float* __restrict__ eta_p, float mask, float* __restrict__ v,
                                                            float* __restrict__ out, int R) {
    __shared__ float sh[4][4];
    const float eta = eta_p[0], inv = 1.f / (float)R;
    float sj = 0.f, sg = 0.f, sv = 0.f, ss = 0.f;
    for (int i = threadIdx.x; i < R; i += 256) {
        const float p = pce[i], n = nce[i], ce = (p + n) * 0.5f;
        float vi, di;
        eta_weight(ce, eta, mask, 1, vi, di);
        v[i] = vi;
        const float coef = (p + n) * inv - 1.f;          // dJ / dv_i
        sj += coef * vi; sg += coef * di; sv += vi; ss += ce < eta ? 1.f : 0.f;
    }
    sj = wave_sum(sj); sg = wave_sum(sg); sv = wave_sum(sv); ss = wave_sum(ss);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[wv][0] = sj; sh[wv][1] = sg; sh[wv][2] = sv; sh[wv][3] = ss; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float t = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        out[threadIdx.x] = threadIdx.x < 2 ? t : t * inv;
    }
}

// ------------------------------------------------------------------- relative-position-bias tables
// tbl[h][i] = rel_bias[bucket[i]][h] for i in [0, L): the additive bias as a function of
// (key - query) only; the bucket of each offset is computed on the host.
__global__ void relbias_table_fwd_kernel(const bf16* __restrict__ rel_bias, const int* __restrict__ bucket,
                                         float* __restrict__ tbl, int H, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * L) return;
    const int h = i / L, j = i % L;
    tbl[i] = bf2f(rel_bias[bucket[j] * H + h]);
}
__global__ void relbias_table_bwd_kernel(const float* __restrict__ dtbl, const int* __restrict__ bucket,
                                         float* __restrict__ drel_bias, int H, int L, int replicas) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * L) return;
    const int h = i / L, j = i % L;
    float s = 0.f;
    for (int r = 0; r < replicas; ++r) s += dtbl[(size_t)r * H * L + i];
    atomicAdd(drel_bias + bucket[j] * H + h, s);
}

// -------------------------------------------------------------------------------------------- AdamW
// Hugging Face AdamW (the optimiser the reference constructs): m = b1 m + (1-b1) g;
// v = b2 v + (1-b2) g^2; p -= step_size * m / (sqrt(v) + eps); then p -= lr*wd*p.  step_size
// (= lr * sqrt(1-b2^t)/(1-b1^t)) is computed on the host.  Also refreshes the bf16 shadow copy.
// CLEAR: the gradient is set to zero once it has been read (the next backward pass accumulates into it), which saves the
// separate fill pass over the buffer.
template <bool CLEAR>
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16* __restrict__ pb, size_t n,
                                                   float b1, float b2, float eps, float step_size, float decay, float gscale) {
    const size_t n4 = n >> 2;
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < n4; c += (size_t)gridDim.x * 256) {
        f32x4 pv = *reinterpret_cast<f32x4*>(p + c * 4);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + c * 4);
        f32x4 mv = *reinterpret_cast<f32x4*>(m + c * 4);
        f32x4 vv = *reinterpret_cast<f32x4*>(v + c * 4);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = gv[j] * gscale;
            mv[j] = mv[j] * b1 + gj * (1.f - b1);
            vv[j] = vv[j] * b2 + gj * gj * (1.f - b2);
            pv[j] = pv[j] - step_size * (mv[j] / (sqrtf(vv[j]) + eps));
            pv[j] = pv[j] - decay * pv[j];
            o[j] = f2bf(pv[j]);
        }
        *reinterpret_cast<f32x4*>(p + c * 4) = pv;
        *reinterpret_cast<f32x4*>(m + c * 4) = mv;
        *reinterpret_cast<f32x4*>(v + c * 4) = vv;
        if (pb) *reinterpret_cast<bf16x4*>(pb + c * 4) = o;
        if (CLEAR) *reinterpret_cast<f32x4*>(g + c * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        const float gj = g[i] * gscale;
        const float mj = m[i] * b1 + gj * (1.f - b1);
        const float vj = v[i] * b2 + gj * gj * (1.f - b2);
        float pj = p[i] - step_size * (mj / (sqrtf(vj) + eps));
        pj -= decay * pj;
        p[i] = pj; m[i] = mj; v[i] = vj;
        if (pb) pb[i] = f2bf(pj);
        if (CLEAR) g[i] = 0.f;
    }
}

int flat_grid(size_t work_items) {
    size_t g = (work_items + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

}  // namespace

extern "C" int paced_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int d, float eps,
                                 float out_scale, float drop_p, uint32_t drop_seed, void* stream) {
    PACED_CHECK_ARG(x && w && y, "paced_rmsnorm_fwd: null operand");
    PACED_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0, "paced_rmsnorm_fwd: need M > 0 and d a positive multiple of 8 (got %d, %d)", M, d);
    hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3(row_grid(M)), dim3(ROW_BLOCK), 0, (hipStream_t)stream,
                       (const bf16*)x, (const bf16*)w, (bf16*)y, rstd, M, d, eps, out_scale, make_drop(drop_p, drop_seed));
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres,
                                 void* dx, void* dx2, float* dw, int M, int d, float out_scale,
                                 float drop_p, uint32_t drop_seed, float drop2_p, uint32_t drop2_seed, void* stream) {
    PACED_CHECK_ARG(dy && x && w && rstd && dx, "paced_rmsnorm_bwd: null operand");
    PACED_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0 && d <= 2048, "paced_rmsnorm_bwd: need M > 0 and d a multiple of 8 up to 2048 (got %d, %d)", M, d);
    const DropCfg drop = make_drop(drop_p, drop_seed), drop2 = make_drop(drop2_p, drop2_seed);
#define BWD_ROWS(L, P) launch_rmsnorm_bwd_rows<L, P>((const bf16*)dy, (const bf16*)x, (const bf16*)w, rstd, (const bf16*)dres, (bf16*)dx, \
                                                    (bf16*)dx2, dw, M, out_scale, drop, drop2, (hipStream_t)stream)
    if (d == 768) BWD_ROWS(16, 6);
    else if (d == 512) BWD_ROWS(16, 4);
    else {
        int grid = row_grid(M);
        if (grid > 1024) grid = 1024;   // bounds the per-column atomics on dw
        hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(grid), dim3(ROW_BLOCK), 0, (hipStream_t)stream,
                           (const bf16*)dy, (const bf16*)x, (const bf16*)w, rstd, (const bf16*)dres, (bf16*)dx, (bf16*)dx2, dw,
                           M, d, out_scale, drop, drop2);
    }
#undef BWD_ROWS
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_embedding_fwd(const long long* ids, const void* table, void* out, int M, int d, int vocab,
                                   float drop_p, uint32_t drop_seed, void* stream) {
    PACED_CHECK_ARG(ids && table && out, "paced_embedding_fwd: null operand");
    PACED_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0 && vocab > 0, "paced_embedding_fwd: bad shape M=%d d=%d vocab=%d", M, d, vocab);
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(row_grid(M)), dim3(ROW_BLOCK), 0, (hipStream_t)stream,
                       ids, (const bf16*)table, (bf16*)out, M, d, vocab, make_drop(drop_p, drop_seed));
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_embedding_bwd(const long long* ids, const void* dout, float* dtable, int M, int d, int vocab,
                                   float drop_p, uint32_t drop_seed, void* stream) {
    PACED_CHECK_ARG(ids && dout && dtable, "paced_embedding_bwd: null operand");
    PACED_CHECK_ARG(M > 0 && d > 0 && vocab > 0, "paced_embedding_bwd: bad shape M=%d d=%d vocab=%d", M, d, vocab);
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(row_grid(M)), dim3(ROW_BLOCK), 0, (hipStream_t)stream,
                       ids, (const bf16*)dout, dtable, M, d, vocab, make_drop(drop_p, drop_seed));
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_dropout_apply(const void* x, void* y, size_t n, float drop_p, uint32_t drop_seed, void* stream) {
    PACED_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "paced_dropout_apply: need n a positive multiple of 8");
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(flat_grid(n / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, (bf16*)y, n / 8, make_drop(drop_p, drop_seed));
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_dropout_mask(float* out, size_t n, float drop_p, uint32_t drop_seed, void* stream) {
    PACED_CHECK_ARG(out && n > 0, "paced_dropout_mask: bad arguments");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, out, n, make_drop(drop_p, drop_seed));
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_cast_f32_bf16(const float* x, void* y, size_t n, void* stream) {
    PACED_CHECK_ARG(x && y && n > 0, "paced_cast_f32_bf16: bad arguments");
    PACED_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0, "paced_cast_f32_bf16: misaligned");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(flat_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)y, n);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_cross_entropy(const float* logits, const long long* labels, float* loss, void* dlogits,
                                   int rows, int V, int ldl, long long ignore_index, float gscale, void* stream) {
    PACED_CHECK_ARG(logits && labels && loss, "paced_cross_entropy: null operand");
    PACED_CHECK_ARG(rows > 0 && V > 0 && V % 4 == 0 && ldl % 4 == 0 && ldl >= V, "paced_cross_entropy: need V, ldl multiples of 4 (rows=%d V=%d ldl=%d)", rows, V, ldl);
    hipLaunchKernelGGL(ce_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, labels, loss, (bf16*)dlogits, V, ldl, ignore_index, gscale);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_lce_group(const float* pce, const float* nce, float* ce, int R, int n, int use_mean, void* stream) {
    PACED_CHECK_ARG(pce && nce && ce && R > 0 && n > 0, "paced_lce_group: bad arguments");
    hipLaunchKernelGGL(lce_group_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, pce, nce, ce, R, n, use_mean);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_eta_weights(const float* loss, const float* eta, float mask, int form, float* w, float* dw, int R, void* stream) {
    PACED_CHECK_ARG(loss && eta && w && R > 0 && (form == 0 || form == 1), "paced_eta_weights: bad arguments");
    hipLaunchKernelGGL(eta_weights_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, loss, eta, mask, form, w, dw, R);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}
extern "C" int paced_eta_bound_meta(const float* pce, const float* nce, const float* eta, float mask, float* v, float* out, int R, void* stream) {
    PACED_CHECK_ARG(pce && nce && eta && v && out && R > 0, "paced_eta_bound_meta: bad arguments");
    hipLaunchKernelGGL(eta_bound_meta_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pce, nce, eta, mask, v, out, R);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}
extern "C" int paced_lce_weights(const float* loss, const float* eta, float* w, float* stats, int R, void* stream) {
    PACED_CHECK_ARG(loss && eta && w && stats && R > 0, "paced_lce_weights: bad arguments");
    hipLaunchKernelGGL(lce_weights_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss, eta, w, stats, R);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_relbias_table_fwd(const void* rel_bias, const int* bucket, float* tbl, int H, int L, void* stream) {
    PACED_CHECK_ARG(rel_bias && bucket && tbl && H > 0 && L > 0, "paced_relbias_table_fwd: bad arguments");
    hipLaunchKernelGGL(relbias_table_fwd_kernel, dim3((H * L + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)rel_bias, bucket, tbl, H, L);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_relbias_table_bwd(const float* dtbl, int replicas, const int* bucket, float* drel_bias, int H, int L, void* stream) {
    PACED_CHECK_ARG(dtbl && bucket && drel_bias && H > 0 && L > 0 && replicas >= 1, "paced_relbias_table_bwd: bad arguments");
    hipLaunchKernelGGL(relbias_table_bwd_kernel, dim3((H * L + 255) / 256), dim3(256), 0, (hipStream_t)stream, dtbl, bucket, drel_bias, H, L, replicas);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

extern "C" int paced_adamw(float* p, float* g, float* m, float* v, void* p_bf16, size_t n,
                           float beta1, float beta2, float eps, float step_size, float decay, float grad_scale, int clear_grad,
                           void* stream) {
    PACED_CHECK_ARG(p && g && m && v && n > 0, "paced_adamw: bad arguments");
    PACED_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)p_bf16 & 7) == 0, "paced_adamw: misaligned buffers");
    if (clear_grad)
        hipLaunchKernelGGL(adamw_kernel<true>, dim3(flat_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream,
                           p, g, m, v, (bf16*)p_bf16, n, beta1, beta2, eps, step_size, decay, grad_scale);
    else
        hipLaunchKernelGGL(adamw_kernel<false>, dim3(flat_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream,
                           p, g, m, v, (bf16*)p_bf16, n, beta1, beta2, eps, step_size, decay, grad_scale);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

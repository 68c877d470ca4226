// LDS read rates on gfx950 for the access patterns of the attention kernels: bytes per clock and CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_rate.hip -o gpurun_out/lds_rate && gpurun_out/lds_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ int row_img_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ int tr_img_off(int r, int c) { return r * 128 + ((((c >> 2) ^ ((r >> 1) & 1))) << 6) + ((c & 3) << 4); }

// MODE 0: ds_read_b128, row image, lane = row (frag_rows).  1: ds_read_b64_tr_b16, tr image (frag_cols).
// 2: ds_read2_b32 of consecutive floats (bias table).  3: ds_read_b128 all lanes of a half the same address (LSE).
// 4: ds_read_b128 linear (lane * 16).  5: ds_read_b64 linear.  6: ds_swizzle.  7: ds_write_b128 row image.
template <int MODE>
__global__ void __launch_bounds__(256) lds_kernel(uint32_t* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384 / 4; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ks = u & 3, rb = (u >> 2) * 32;
            if (MODE == 0) {
                const int r = rb + (lane & 31);
                u32x4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(row_img_off(r, 2 * ks + (lane >> 5))));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc ^= v.x;
            } else if (MODE == 1) {
                const int h = lane >> 5, g1 = (lane >> 4) & 1, i = lane & 15, q4 = i >> 2, p4 = i & 3;
                const int col = (u & 1) * 32 + 16 * g1 + 4 * p4, c = col >> 3, within = (col & 7) * 2;
                const int r0 = rb + ((u >> 1) & 1) * 16 + 4 * h + q4;
                u32x2 v;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(tr_img_off(r0, c) + within));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc ^= v.x;
            } else if (MODE == 2) {
                u32x2 v;
                asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(v) : "v"((lane + 8 * u) * 4));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc ^= v.x;
            } else if (MODE == 3) {
                u32x4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((lane >> 5) * 16 + u * 32));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc ^= v.x;
            } else if (MODE == 4) {
                u32x4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lane * 16 + u * 1024));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc ^= v.x;
            } else if (MODE == 5) {
                u32x2 v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(lane * 8 + u * 512));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc ^= v.x;
            } else if (MODE == 6) {
                uint32_t v;
                asm volatile("ds_swizzle_b32 %0, %1 offset:0xC0A0" : "=v"(v) : "v"(acc + u));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc ^= v;
            } else if (MODE == 7) {
                const int r = rb + (lane & 31);
                u32x4 v = {acc, acc, acc, acc};
                asm volatile("ds_write_b128 %0, %1" :: "v"(row_img_off(r, 2 * ks + (lane >> 5)) + w * 0), "v"(v) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE> void run(const char* name, int bytes_per_lane) {
    uint32_t* out; hipMalloc(&out, 2048 * 256 * 4);
    int dev = 0, mhz = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, dev);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {
        const int iters = 20000, grid = 256 * per_cu;
        lds_kernel<MODE><<<grid, 256, 16384>>>(out, 100);
        hipEventRecord(e0);
        lds_kernel<MODE><<<grid, 256, 16384>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_cu = (double)iters * 8 * 4 * per_cu;      // wave instructions per CU
        const double ns_per = ms * 1e6 / instr_per_cu;
        printf("%-34s waves/CU %2d: %.2f ns per wave instruction per CU = %.1f bytes/ns per CU (%.0f B/clk at 2.4 GHz)\n", name, 4 * per_cu, ns_per,
               64.0 * bytes_per_lane / ns_per, 64.0 * bytes_per_lane / ns_per / 2.4);
    }
    hipFree(out);
}
int main() {
    run<0>("ds_read_b128 row image (frag_rows)", 16);
    run<1>("ds_read_b64_tr_b16 tr image", 8);
    run<2>("ds_read2_b32 consecutive", 8);
    run<3>("ds_read_b128 broadcast", 16);
    run<4>("ds_read_b128 linear", 16);
    run<5>("ds_read_b64 linear", 8);
    run<6>("ds_swizzle_b32", 4);
    run<7>("ds_write_b128 row image", 16);
    return 0;
}

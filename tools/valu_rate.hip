// Issue rate of a few integer VALU instructions on gfx950: cycles per wave-instruction on one SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(uint32_t* out, uint32_t c, int iters, long long* cyc) {
    uint32_t x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 9 + i + c;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (OP == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 5) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(x[i]));
            if (OP == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t*)&x[i & 6]) : "v"(x[7]), "v"(c) : "vcc");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name) {
    uint32_t* out; long long* cyc; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 2; ++waves) {    // workgroups per CU: 1 -> one wave per SIMD, 2 -> two
        int grid = 256 * waves;
        rate_kernel<OP><<<grid, 256>>>(out, 0x9e3779b1u, 16, cyc);
        hipEventRecord(e0);
        rate_kernel<OP><<<grid, 256>>>(out, 0x9e3779b1u, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-16s waves/SIMD %d: %.2f ms, wave0 %.1f cycle-counter ticks per instr, %.2f ns per instr per SIMD\n", name, waves, ms, (double)c / (iters * 8.0),
               ms * 1e6 / (iters * 8.0 * waves));
    }
}
int main() {
    run<2>("v_xor_b32"); run<0>("v_mul_lo_u32"); run<1>("v_mul_u32_u24"); run<3>("v_mad_u32_u24"); run<4>("v_mul_hi_u32"); run<5>("v_alignbit_b32"); run<6>("v_mad_u64_u32");
    return 0;
}

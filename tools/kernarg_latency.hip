// How long a workgroup waits for its kernel arguments on gfx950: cycles from entry to the first use of an argument,
// for arguments read with s_load (a struct by value) and for arguments preloaded into SGPRs by the dispatcher
// (plain scalars; build with -mllvm -amdgpu-kernarg-preload-count=16).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/kernarg_latency.hip -o ab/kernarg_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
struct Args { long long* out; const float* a; const float* b; int n; int pad[20]; };
__global__ void __launch_bounds__(256) by_struct(Args x) {
    const long long t0 = __builtin_readcyclecounter();
    const int n = x.n;                                   // first use of an argument: waits for the s_load
    asm volatile("" :: "s"(n));
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += x.a[i] * x.b[i];
    if (acc == 12345.f) x.out[0] = 1;
    if (threadIdx.x == 0) x.out[blockIdx.x] = t1 - t0;
}
__global__ void __launch_bounds__(256) by_scalars(long long* out, const float* a, const float* b, int n) {
    const long long t0 = __builtin_readcyclecounter();
    asm volatile("" :: "s"(n));
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += a[i] * b[i];
    if (acc == 12345.f) out[0] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
static void report(const char* name, long long* d_out, int G) {
    std::vector<long long> h(G);
    (void)hipMemcpy(h.data(), d_out, G * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double s = 0; for (auto v : h) s += v;
    printf("%-12s workgroups %d: cycles entry -> argument usable: mean %.0f  median %lld  p90 %lld  max %lld\n", name, G, s / G, h[G / 2], h[G * 9 / 10], h[G - 1]);
}
int main() {
    const int G = 13056, n = 4096;
    long long* out; float *a, *b;
    (void)hipMalloc(&out, G * 8); (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4);
    (void)hipMemset(a, 0, n * 4); (void)hipMemset(b, 0, n * 4);
    Args x = {}; x.out = out; x.a = a; x.b = b; x.n = n;
    for (int r = 0; r < 2; ++r) {
        by_struct<<<G, 256>>>(x); (void)hipDeviceSynchronize(); report("struct", out, G);
        by_scalars<<<G, 256>>>(out, a, b, n); (void)hipDeviceSynchronize(); report("scalars", out, G);
    }
    return 0;
}

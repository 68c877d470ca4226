// Stand-in for a communication kernel that keeps compute units busy beside the training step: `wgs` workgroups that
// each wait `usec` microseconds on the constant 100 MHz clock and leave.  `lds_bytes` of dynamic LDS per workgroup decide
// how many fit a compute unit (more than half of the 160 KB: one per unit, so `wgs` units are held; 0: the dispatcher may
// stack them).  Built by tools/cu_holder_step.py into tools/libcu_holder.so.
#include <hip/hip_runtime.h>
extern "C" __global__ void __launch_bounds__(256) hold_kernel(long long ticks, int* sink) {
    extern __shared__ int lds[];
    const long long t0 = wall_clock64();
    int spins = 0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(64); ++spins; }
    if (sink && spins < 0) lds[threadIdx.x] = sink[0] = spins;      // never taken: keeps the LDS allocation and the loop alive
}
extern "C" int hold_cus(int wgs, int lds_bytes, long long usec, void* stream) {
    if (wgs <= 0) return 0;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)hold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(hold_kernel, dim3(wgs), dim3(256), lds_bytes, (hipStream_t)stream, usec * 100, (int*)nullptr);
    return (int)hipGetLastError();
}

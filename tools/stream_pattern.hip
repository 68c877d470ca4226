// What HBM delivers to LDS-DMA streams of a [rows][768] bf16 matrix, by the shape of a step: 128 rows x 64 columns (128-byte
// pieces at a 1536-byte stride: the tile of the GEMM kernels) against 16 whole rows (1536-byte pieces), and by workgroups
// per compute unit.  No arithmetic: issue, counted wait, barrier.   build: hipcc --offload-arch=gfx950 -O3 tools/stream_pattern.hip -o /tmp/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE, int NS, int AUX, int IA>
__device__ __forceinline__ void issue_step(const char* base, char* smem, long long s, long long total, int b, int G, int w, const unsigned (&off)[IA]) {
    constexpr int STEP = MODE == 0 ? 16384 : 24576;
    __amdgpu_buffer_rsrc_t r;
    unsigned soff = 0;
    if (s < total) {
        const long long unit = b + (MODE == 0 ? s / 12 : s) * G;
        const char* p = base + unit * (MODE == 0 ? 128LL * 1536 : 16LL * 1536);
        r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, MODE == 0 ? 128 * 1536 : 16 * 1536, 0x00020000);
        soff = MODE == 0 ? (unsigned)(s % 12) * 128 : 0;
    } else r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0, 0x00020000);
    char* dst = smem + (s % NS) * STEP;
#pragma unroll
    for (int i = 0; i < IA; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(dst + (i * 4 + w) * 1024), 16, off[i], soff, 0, AUX);
}

// MODE 0: step = 128 rows x 64 cols (16 KiB, 12 steps per 128-row block);  MODE 1: step = 16 rows x 768 cols (24 KiB)
template <int MODE, int NS, int AUX>
__global__ void __launch_bounds__(256) k(const char* base, long long rows, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STEP = MODE == 0 ? 16384 : 24576;
    constexpr int IA = STEP / 1024 / 4;                  // DMA instructions per wave and step
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, G = gridDim.x, b = blockIdx.x;
    const long long units = MODE == 0 ? rows / 128 : rows / 16;      // blocks of rows
    const long long mine = b < units ? (units - b + G - 1) / G : 0;
    const long long total = MODE == 0 ? mine * 12 : mine;
    unsigned off[IA];
    for (int i = 0; i < IA; ++i) {
        const int g = (i * 4 + w) * 64 + lane;           // 16-byte unit inside the step's LDS image
        if (MODE == 0) { const int r = g >> 3, c = g & 7; off[i] = r * 1536 + c * 16; }
        else { const int r = g / 96, c = g % 96; off[i] = r * 1536 + c * 16; }
    }
    auto issue = [&](long long s) { issue_step<MODE, NS, AUX, IA>(base, smem, s, total, b, G, w, off); };
    for (int s = 0; s < NS - 1; ++s) issue(s);
    int acc = 0;
    for (long long s = 0; s < total; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IA) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(s + NS - 1);
        acc += *reinterpret_cast<int*>(smem + (s % NS) * STEP + threadIdx.x * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678) sink[0] = acc;
}

template <int MODE, int NS, int AUX>
void run(const char* name, const char* buf, long long rows, int* sink, int per_cu) {
    constexpr int STEP = MODE == 0 ? 16384 : 24576;
    const int smem = NS * STEP;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, NS, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<MODE, NS, AUX>), dim3(256 * per_cu), dim3(256), smem, 0, buf, rows, sink);
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<MODE, NS, AUX>), dim3(256 * per_cu), dim3(256), smem, 0, buf, rows, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %d wg/cu, %d slots of %d KiB: %7.1f us  %6.2f TB/s\n", name, per_cu, NS, STEP / 1024, ms / 5 * 1e3, rows * 1536.0 / (ms / 5 * 1e-3) / 1e12);
}

int main() {
    const long long rows = 1 << 21;          // 3.2 GB
    char* buf; int* sink;
    hipMalloc(&buf, rows * 1536); hipMalloc(&sink, 4);
    hipMemset(buf, 1, rows * 1536);
    run<0, 4, 0>("128 rows x 64 cols per step", buf, rows, sink, 2);
    run<0, 4, 2>("128 rows x 64 cols per step, nt", buf, rows, sink, 2);
    run<0, 8, 2>("128 rows x 64 cols per step, nt", buf, rows, sink, 1);
    run<1, 3, 0>("16 whole rows per step", buf, rows, sink, 2);
    run<1, 3, 2>("16 whole rows per step, nt", buf, rows, sink, 2);
    run<1, 6, 2>("16 whole rows per step, nt", buf, rows, sink, 1);
    run<1, 4, 2>("16 whole rows per step, nt", buf, rows, sink, 1);
    run<1, 2, 2>("16 whole rows per step, nt", buf, rows, sink, 3);
    hipDeviceSynchronize();
    return 0;
}

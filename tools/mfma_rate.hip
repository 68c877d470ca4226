// Sustained MFMA rate under power limits: 16x16x32 vs 32x32x16 bf16, random operands, registers only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ void __launch_bounds__(512) k(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
    float sum = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i * 2 + kk], b[j * 2 + kk], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][7];
    }
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main() {
    bf16x8* in; float* out;
    hipMalloc(&in, 4096 * 16); hipMalloc(&out, 256 * 512 * 4 * 4);
    unsigned short* h = (unsigned short*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
    hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {16, 32}) {
            hipEventRecord(e0);
            if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, in, out, iters);
            else hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per wave per iteration: 16 MFMA 16x16x32 (16384 flop each) or 8 MFMA 32x32x16 (32768 flop each) = 262144 flop
            double fl = 256.0 * 8 * iters * 262144.0;
            printf("shape %2d: %8.2f ms  %8.1f TFLOP/s\n", shape, ms, fl / ms / 1e9);
        }
    return 0;
}

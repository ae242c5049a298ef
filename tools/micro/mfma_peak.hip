// calibration: back-to-back v_mfma_f32_32x32x2_f32 with 4 independent accumulators, no memory traffic
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *out; hipMalloc(&out, 256 * 1024 * 4 * 4);
    for (int wpb : {4, 16}) {
        const int iters = 2000, blocks = 256 * (wpb == 4 ? 4 : 1);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * wpb), 0, 0, out, 10, 1.0f, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * wpb), 0, 0, out, iters, 1.0f, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * wpb * iters * 64.0 * 4096.0;
        printf("waves/block %d: %.3f ms, %.1f TFLOP/s\n", wpb, ms, flops / ms / 1e9);
    }
    return 0;
}

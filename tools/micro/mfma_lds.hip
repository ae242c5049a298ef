// calibration 2: the inner-loop pattern of mlp.hip -- one 16-byte LDS read feeding four MFMAs -- without anything else
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float s_w[128 * 128];
    for (int i = threadIdx.x; i < 128 * 128; i += blockDim.x) s_w[i] = 1e-3f * (i & 255);
    __syncthreads();
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const int lane = threadIdx.x & 63;
    const float *wrow = s_w + ((lane >> 5) * 16 * 32 + (lane & 31)) * 4;
    float a = a0 + threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float4 t;
            if (MODE == 0) t = make_float4(a, a, a, a);
            else t = *reinterpret_cast<const float4 *>(wrow + ((s + (MODE == 2 ? (i & 3) * 32 : 0)) & 127) * 128);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t.w, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(float *out) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms, %.1f TFLOP/s\n", MODE, ms, (double)blocks * 16 * iters * 64.0 * 4096.0 / ms / 1e9);
}
int main() {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    run<0>(out); run<1>(out); run<2>(out);
    return 0;
}

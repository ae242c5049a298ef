// calibration 3: the memory access pattern of mlp.hip's linear_kernel (K = N = 128) WITHOUT the MFMAs: cooperative chunk
// loads through a wavefront-private LDS buffer, epilogue stores in the 32x32 C/D layout.  How long does the traffic alone take?
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kThreads = 1024, kRows = 512, kPitch = 36;
template <int MODE>   // 0: loads + stores, 1: loads only, 2: stores only
__global__ __launch_bounds__(kThreads) void k(int P, const float *__restrict__ X, float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
    float *s_x = smem + wave * 32 * kPitch;
    const int ntiles = (P + kRows - 1) / kRows;
    float sum = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * kRows + wave * 32;
        float a[16];
        for (int kc = 0; kc < 128; kc += 32) {
            if (MODE != 2) {
                float4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = row0 + 8 * j + (lane >> 3);
                    v[j] = r < P ? *reinterpret_cast<const float4 *>(X + (size_t)r * 128 + kc + 4 * (lane & 7)) : make_float4(0, 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(s_x + (8 * j + (lane >> 3)) * kPitch + 4 * (lane & 7)) = v[j];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 t = *reinterpret_cast<const float4 *>(s_x + l32 * kPitch + 16 * half + 4 * j);
                    a[4 * j] = t.x; a[4 * j + 1] = t.y; a[4 * j + 2] = t.z; a[4 * j + 3] = t.w;
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) sum += a[s];
            }
        }
        if (MODE != 1) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (rr < P) Y[(size_t)rr * 128 + l32 + 32 * nb] = sum + r;
                }
        }
    }
    if (MODE == 1 && sum == 12345.678f) Y[0] = sum;
}
template <int MODE> void run(int P, const float *X, float *Y, const char *what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 16 * 32 * kPitch * 4 + 65536;      // same LDS footprint as the real kernel: one workgroup per CU
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(kThreads), lds, 0, P, X, Y);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(kThreads), lds, 0, P, X, Y);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-16s %.1f us per launch\n", what, ms / 20 * 1e3);
}
int main() {
    const int P = 500000;
    float *X, *Y; hipMalloc(&X, (size_t)P * 128 * 4); hipMalloc(&Y, (size_t)P * 128 * 4);
    hipMemset(X, 0, (size_t)P * 128 * 4);
    run<0>(P, X, Y, "loads + stores"); run<1>(P, X, Y, "loads only"); run<2>(P, X, Y, "stores only");
    return 0;
}

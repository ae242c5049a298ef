// calibration 4: which ingredient of mlp.hip's linear_kernel costs the matrix pipe its second half?
//   mode 0: MFMA stream fed by LDS B reads and by an A chunk re-staged through LDS every 64 MFMAs (no global memory)
//   mode 1: + the epilogue stores (acc -> global, C/D layout)         mode 2: + the chunk loads from global memory
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kPitch = 36;
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(int P, const float *__restrict__ X, float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_w = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
    float *s_x = smem + 128 * 128 + wave * 32 * kPitch;
    for (int i = tid; i < 128 * 128; i += 64 * WAVES) s_w[i] = 1e-3f * (float)((i * 2654435761u >> 20) & 1023) - 0.5f;
    __syncthreads();
    constexpr int kRows = 32 * WAVES;
    const int ntiles = P / kRows;
    float4 nxt[4];
    for (int j = 0; j < 4; ++j) nxt[j] = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * kRows + wave * 32;
        f32x16 acc[4];
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        for (int kc = 0; kc < 128; kc += 32) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(s_x + (8 * j + (lane >> 3)) * kPitch + 4 * (lane & 7)) = nxt[j];
            if (MODE >= 2) {
                const bool same = kc + 32 < 128;
                const int r0 = same ? row0 : row0 + (int)gridDim.x * kRows, kk = same ? kc + 32 : 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int r = r0 + 8 * j + (lane >> 3); r = r < P ? r : P - 1;
                    nxt[j] = *reinterpret_cast<const float4 *>(X + (size_t)r * 128 + kk + 4 * (lane & 7));
                }
            }
            __builtin_amdgcn_wave_barrier();
            float a[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 t = *reinterpret_cast<const float4 *>(s_x + l32 * kPitch + 16 * half + 4 * j);
                a[4 * j] = t.x; a[4 * j + 1] = t.y; a[4 * j + 2] = t.z; a[4 * j + 3] = t.w;
            }
            const float *wrow = s_w + ((kc + 16 * half) * 32 + l32) * 4;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float4 t = *reinterpret_cast<const float4 *>(wrow + s * 128);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], t.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], t.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], t.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], t.w, acc[3], 0, 0, 0);
            }
        }
        if (MODE >= 1) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Y[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * half) * 128 + l32 + 32 * nb] = acc[nb][r];
        } else {
            float sacc = 0.f;
            for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) sacc += acc[q][r];
            if (sacc == 12345.678f) Y[0] = sacc;
        }
    }
}
template <int MODE, int WAVES> void run(int P, const float *X, float *Y) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (128 * 128 + WAVES * 32 * kPitch) * 4;
    hipFuncSetAttribute((const void *)k<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), lds, 0, P, X, Y);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), lds, 0, P, X, Y);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d, %2d waves/CU: %.1f us per launch (%.1f TFLOP/s)\n", MODE, WAVES, ms / 20 * 1e3, 2.0 * P * 128 * 128 / (ms / 20) / 1e9);
}
int main() {
    const int P = 499712;     // 976 x 512
    float *X, *Y; hipMalloc(&X, (size_t)P * 128 * 4); hipMalloc(&Y, (size_t)P * 128 * 4);
    {   // random activations (all-zero data toggles no bits: lower power, optimistic clocks)
        float *h = (float *)malloc((size_t)P * 128 * 4);
        unsigned st = 12345u;
        for (size_t i = 0; i < (size_t)P * 128; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
        hipMemcpy(X, h, (size_t)P * 128 * 4, hipMemcpyHostToDevice);
        free(h);
    }
    run<0, 16>(P, X, Y); run<1, 16>(P, X, Y); run<2, 16>(P, X, Y);
    run<0, 8>(P, X, Y); run<1, 8>(P, X, Y); run<2, 8>(P, X, Y);
    run<0, 4>(P, X, Y); run<2, 4>(P, X, Y);
    return 0;
}

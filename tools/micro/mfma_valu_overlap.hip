// Do MFMA and VALU work of DIFFERENT wavefronts on one SIMD overlap?  One workgroup of 8 wavefronts per CU (2 per SIMD):
// wavefronts 0-3 issue N x v_mfma_f32_32x32x16_bf16 (4 independent accumulators), wavefronts 4-7 issue M dependent-free
// v_fma_f32 (8 chains).  Modes: 1 = MFMA only, 2 = VALU only, 3 = both.  Prints microseconds.
// hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
__global__ __launch_bounds__(512) void k(int mode, int n_mfma, int n_valu, float *out, int small) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            bf16x8_t a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
            if (small) {                                    // 16x16x32: 4 passes, 4 accumulator registers
                f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
                for (int i = 0; i < 2 * n_mfma; i += 4) {
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d1, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d2, 0, 0, 0);
                    d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d3, 0, 0, 0);
                }
                r = d0[0] + d1[1] + d2[2] + d3[3];
            } else {
            f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            for (int i = 0; i < n_mfma; i += 4) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
            r = c0[0] + c1[1] + c2[2] + c3[3];
            }
        }
    } else if (mode & 2) {
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
        const float m = 1.0001f, q = 0.5f;
        for (int i = 0; i < n_valu; i += 8) {
            x0 = __builtin_fmaf(x0, m, q); x1 = __builtin_fmaf(x1, m, q); x2 = __builtin_fmaf(x2, m, q); x3 = __builtin_fmaf(x3, m, q);
            x4 = __builtin_fmaf(x4, m, q); x5 = __builtin_fmaf(x5, m, q); x6 = __builtin_fmaf(x6, m, q); x7 = __builtin_fmaf(x7, m, q);
        }
        r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}
int main() {
    float *out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n_mfma = 4096, n_valu = 32768;      // 4096 x 32 cycles = 131k cycles; 32768 x 4 cycles = 131k cycles
    for (int small : {0, 1})
    for (int mode : {1, 2, 3, 1, 2, 3}) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, n_mfma, n_valu, out, small);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, n_mfma, n_valu, out, small);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s mode %d (%s): %.1f us\n", small ? "16x16x32" : "32x32x16", mode, mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", 1e3f * ms);
    }
    return 0;
}

// mlp.hip -- the dense layer of the reference's field networks (models/mlp.py:39-110:
// DeformationField / CanonicalField / ColorField are all  z -> [Linear(128) + leaky_relu(0.1)] x (1 + n_layers) -> Linear).
//
// One kernel template does every GEMM of the forward and of the input-gradient chain of the backward:
//     Y[r][n] = act_out( sum_k A[r][k] * Wt[k][n] + bias[n] ),        A = X            (forward)
//                                                                     A = dY (.) lrelu'(H)   (backward; A is also stored)
// with exact f32 arithmetic on the matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf chain;
// 157 TFLOP/s peak on MI355X = the f32 vector peak, but no operand shuffling and no VALU slots spent on the products).
//
// Work decomposition: a workgroup (4 wavefronts) owns 128 rows, a wavefront 32 rows x all output columns
// (NB blocks of 32).  The 32x32x2 instruction contracts two k per issue: lanes 0-31 supply k_a, lanes 32-63 k_b.  K is
// split in two halves -- lanes 0-31 walk k = 0..KH-1, lanes 32-63 k = KH..2KH-1 -- so every lane reads ONE contiguous
// run of its row (16-byte loads for K = 128) and keeps it in registers: the A operand never touches LDS.  The weight
// panel Wt (2*KH x 32*NB, zero padded, prepared by the host layer) sits in LDS for the whole persistent workgroup;
// lane l reads Wt[k(l)][ (l & 31) + 32 nb ]: 32 consecutive floats per half-wavefront, conflict free.
// Weight gradients (dW = dPre^T X, a reduction over all rows) are plain library GEMMs in the host layer (hipBLASLt).
#include "d3ga_internal.h"

namespace d3ga {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kMlpRows = 128;        // rows per workgroup
constexpr int kMlpMaxKH = 64;        // K <= 128

template <int NB>
__global__ __launch_bounds__(256) void linear_kernel(int P, int K, int KH, int n_store, const float *__restrict__ X,
                                                     const float *__restrict__ mask, float mask_slope,
                                                     float *__restrict__ a_out, const float *__restrict__ Wt,
                                                     const float *__restrict__ bias, float out_slope,
                                                     float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];        // [2*KH][32*NB]
    constexpr int N32 = 32 * NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    {   // weight panel: contiguous copy, 16 bytes per thread and step
        const int nvec = 2 * KH * N32 / 4;
        for (int v = tid; v < nvec; v += 256) reinterpret_cast<float4 *>(s_w)[v] = reinterpret_cast<const float4 *>(Wt)[v];
    }
    __syncthreads();
    const bool fast = (K == 2 * KH) && (KH % 4 == 0);                  // 16-byte loads of the lane's run (K = 128: KH = 64)
    const int ntiles = (P + kMlpRows - 1) / kMlpRows;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * kMlpRows + wave * 32;
        const int row = row0 + l32;
        const bool in = row < P;
        float a[kMlpMaxKH];
        const int k0 = half * KH;
        if (fast) {
#pragma unroll
            for (int j = 0; j < kMlpMaxKH / 4; ++j) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (in && 4 * j < KH) {
                    const size_t o = (size_t)row * K + k0 + 4 * j;
                    x = *reinterpret_cast<const float4 *>(X + o);
                    if (mask) {
                        const float4 m = *reinterpret_cast<const float4 *>(mask + o);
                        x.x *= m.x > 0.f ? 1.f : mask_slope; x.y *= m.y > 0.f ? 1.f : mask_slope;
                        x.z *= m.z > 0.f ? 1.f : mask_slope; x.w *= m.w > 0.f ? 1.f : mask_slope;
                        if (a_out) *reinterpret_cast<float4 *>(a_out + o) = x;
                    }
                }
                a[4 * j] = x.x; a[4 * j + 1] = x.y; a[4 * j + 2] = x.z; a[4 * j + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < kMlpMaxKH; ++s) {
                float x = 0.f;
                const int k = k0 + s;
                if (in && s < KH && k < K) {
                    const size_t o = (size_t)row * K + k;
                    x = X[o];
                    if (mask) {
                        x *= mask[o] > 0.f ? 1.f : mask_slope;
                        if (a_out) a_out[o] = x;
                    }
                }
                a[s] = x;
            }
        }
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const float *wrow = s_w + (size_t)k0 * N32 + l32;
#pragma unroll
        for (int s = 0; s < kMlpMaxKH; ++s) {
            if (s < KH) {                                              // wave-uniform
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wrow[s * N32 + 32 * nb], acc[nb], 0, 0, 0);
            }
        }
        // epilogue: C/D layout of the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = l32 + 32 * nb;
            const float b = (bias && n < n_store) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float y = acc[nb][r] + b;
                y = y > 0.f ? y : out_slope * y;
                if (rr < P && n < n_store) Y[(size_t)rr * n_store + n] = y;
            }
        }
    }
}

}  // namespace d3ga

using namespace d3ga;

// Y (P, n_out) = act( A (P,K) * Wt + bias ),  A = X, or X (.) lrelu'(mask) when mask != NULL (A then also written to a_out).
// Wt: (2*ceil(K/2), 32*ceil(n_out/32)) row-major, zero padded: Wt[k][n] = weight of input k for output n.
extern "C" int d3ga_mlp_linear(int32_t P, int32_t K, int32_t n_out, const float *X, const float *mask, float mask_slope,
                               float *a_out, const float *Wt, const float *bias, float out_slope, float *Y,
                               d3ga_stream_t stream) {
    if (P < 0 || K < 1 || K > 2 * kMlpMaxKH || n_out < 1 || n_out > 128) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!X || !Wt || !Y) return D3GA_E_NULL;
    if (a_out && !mask) return D3GA_E_CONFIG;
    if ((((uintptr_t)X | (uintptr_t)Wt | (uintptr_t)mask | (uintptr_t)a_out) & 15) != 0) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const int KH = (K + 1) / 2, NB = (n_out + 31) / 32;
    const size_t lds = (size_t)2 * KH * 32 * NB * sizeof(float);
    const int ntiles = (P + kMlpRows - 1) / kMlpRows;
    const int grid = ntiles < 512 ? ntiles : 512;                     // persistent: the weight panel is staged once
#define D3GA_MLP_LAUNCH(NBV)                                                                                          \
    do {                                                                                                              \
        static bool attr[64] = {};                                                                                    \
        int dev = 0;                                                                                                  \
        D3GA_HIP(hipGetDevice(&dev));                                                                                 \
        if (dev >= 0 && dev < 64 && !attr[dev]) {                                                                     \
            D3GA_HIP(hipFuncSetAttribute((const void *)linear_kernel<NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         2 * kMlpMaxKH * 32 * NBV * (int)sizeof(float)));                             \
            attr[dev] = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL(linear_kernel<NBV>, dim3(grid), dim3(256), lds, s, P, K, KH, n_out, X, mask, mask_slope,    \
                           a_out, Wt, bias, out_slope, Y);                                                            \
    } while (0)
    switch (NB) {
        case 1: D3GA_MLP_LAUNCH(1); break;
        case 2: D3GA_MLP_LAUNCH(2); break;
        case 3: D3GA_MLP_LAUNCH(3); break;
        default: D3GA_MLP_LAUNCH(4); break;
    }
#undef D3GA_MLP_LAUNCH
    return check_launch(s, 0);
}

// loss.hip -- fused L1 image loss ("loss tail"): mean |a - b| and its gradient.
// Replaces utils/loss_utils.py:29 (l1_loss = torch.abs(network_output - gt).mean()), which autograd runs as six
// full-image ATen kernels (sub, abs, mean, expand/div, sign, mul): here one streaming pass each way.
#include "d3ga_internal.h"

namespace d3ga {

__device__ __forceinline__ float wave_sum_loss(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(kBlock) void l1_mean_fwd_kernel(int64_t n4, int64_t n, const float *__restrict__ a,
                                                             const float *__restrict__ b, float inv_n,
                                                             float *__restrict__ out) {
    __shared__ float s_part[kBlock / 64];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {             // two independent 16-byte loads per array in flight
        const float4 x0 = reinterpret_cast<const float4 *>(a)[i], y0 = reinterpret_cast<const float4 *>(b)[i];
        const float4 x1 = reinterpret_cast<const float4 *>(a)[i + stride], y1 = reinterpret_cast<const float4 *>(b)[i + stride];
        acc += fabsf(x0.x - y0.x) + fabsf(x0.y - y0.y) + fabsf(x0.z - y0.z) + fabsf(x0.w - y0.w);
        acc += fabsf(x1.x - y1.x) + fabsf(x1.y - y1.y) + fabsf(x1.z - y1.z) + fabsf(x1.w - y1.w);
    }
    for (; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i], y = reinterpret_cast<const float4 *>(b)[i];
        acc += fabsf(x.x - y.x) + fabsf(x.y - y.y) + fabsf(x.z - y.z) + fabsf(x.w - y.w);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc += fabsf(a[i] - b[i]);
    acc = wave_sum_loss(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) t += s_part[w];
        atomicAdd(out, t * inv_n);
    }
}

__global__ __launch_bounds__(kBlock) void l1_mean_bwd_kernel(int64_t n4, int64_t n, const float *__restrict__ a,
                                                             const float *__restrict__ b, const float *__restrict__ g,
                                                             float inv_n, float *__restrict__ grad_a) {
    const float s = g[0] * inv_n;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i], y = reinterpret_cast<const float4 *>(b)[i];
        reinterpret_cast<float4 *>(grad_a)[i] =
            make_float4(s * sgn(x.x - y.x), s * sgn(x.y - y.y), s * sgn(x.z - y.z), s * sgn(x.w - y.w));
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        grad_a[i] = s * sgn(a[i] - b[i]);
}

}  // namespace d3ga

using namespace d3ga;

static inline int loss_grid(int64_t n4, int cap) {
    const int64_t blocks = (n4 + kBlock - 1) / kBlock;
    return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

extern "C" int d3ga_l1_mean_fwd(int64_t n, const float *a, const float *b, float *out, d3ga_stream_t stream) {
    if (n <= 0) return D3GA_E_SIZE;
    if (!a || !b || !out) return D3GA_E_NULL;
    if (((uintptr_t)a | (uintptr_t)b) & 15) return D3GA_E_CONFIG;       // 16-byte aligned inputs
    hipStream_t s = (hipStream_t)stream;
    D3GA_HIP(hipMemsetAsync(out, 0, sizeof(float), s));
    const int64_t n4 = n / 4;
    // few, fat workgroups: every workgroup ends with ONE float atomic on the same word, and same-address device-scope
    // atomics serialise at ~12 ns each (2048 of them cost more than streaming the two images)
    hipLaunchKernelGGL(l1_mean_fwd_kernel, dim3(loss_grid(n4, 512)), dim3(kBlock), 0, s, n4, n, a, b, 1.0f / (float)n, out);
    return check_launch(s, 0);
}

extern "C" int d3ga_l1_mean_bwd(int64_t n, const float *a, const float *b, const float *g, float *grad_a,
                                d3ga_stream_t stream) {
    if (n <= 0) return D3GA_E_SIZE;
    if (!a || !b || !g || !grad_a) return D3GA_E_NULL;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad_a) & 15) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(l1_mean_bwd_kernel, dim3(loss_grid(n4, 2048)), dim3(kBlock), 0, s, n4, n, a, b, g, 1.0f / (float)n,
                       grad_a);
    return check_launch(s, 0);
}

// loss.hip -- the loss tail next to the render boundary.
//   l1_mean_*   fused L1 image loss: mean |a - b| and its gradient.  Replaces utils/loss_utils.py:29
//               (torch.abs(network_output - gt).mean()), which autograd runs as six full-image ATen kernels.
//   ssim_*      fused 11x11 Gaussian-window SSIM (utils/loss_utils.py:46-86, called at train.py:192): the reference runs
//               five depthwise conv2d (11x11, zero padding) plus ~15 elementwise kernels forward and their autograd
//               backward; here one LDS-tiled separable pass each way.
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


// ---------------------------------------------------------------------------------------------------------
// SSIM.  ssim_map = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)) with mu = w*x, s1 = w*x^2 - mu1^2,
// s12 = w*xy - mu1 mu2 and w the 11x11 window gaussian(11, 1.5) (x) gaussian(11, 1.5) (utils/loss_utils.py:46-57),
// zero padding of 5 (F.conv2d(..., padding=window_size // 2)).  The window is separable: a workgroup owns a 16x16
// output tile of one channel, stages the (16+10)^2 halo of both images in LDS, convolves the five maps
// (x, y, x^2, y^2, xy) horizontally into LDS and vertically into registers.  The forward also stores the three
// partial derivatives the backward needs (d ssim / d(w*x), d(w*x^2), d(w*xy)); the backward convolves those three maps
// with the same (symmetric) window:  dL/dx = w*Dm + 2 x (w*Dq1) + y (w*Dq12).
// HBM traffic per pixel and channel: forward 8 B read + 12 B written, backward 20 B read + 4 B written.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSsimTile = 16, kSsimHalo = 5, kSsimIn = kSsimTile + 2 * kSsimHalo;     // 26
__device__ __constant__ float c_ssim_w[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                              2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                              3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};
constexpr float kSsimC1 = 0.01f * 0.01f, kSsimC2 = 0.03f * 0.03f;

__global__ __launch_bounds__(256) void ssim_fwd_kernel(int C, int H, int W, int tiles_x, int tiles_y,
                                                       const float *__restrict__ img1, const float *__restrict__ img2,
                                                       float inv_n, float *__restrict__ out, float *__restrict__ Dm,
                                                       float *__restrict__ Dq1, float *__restrict__ Dq12,
                                                       float *__restrict__ out_l1) {
    __shared__ float s_x[kSsimIn][kSsimIn + 1], s_y[kSsimIn][kSsimIn + 1];
    __shared__ float s_h[5][kSsimIn][kSsimTile + 1];
    __shared__ float s_part[8];
    const int tid = threadIdx.x, px = tid & 15, py = tid >> 4;
    const int ntiles = C * tiles_x * tiles_y;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) w[k] = c_ssim_w[k];
    float local = 0.f, local_l1 = 0.f;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int c = t / (tiles_x * tiles_y), r = t - c * tiles_x * tiles_y;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const float *p1 = img1 + (size_t)c * H * W, *p2 = img2 + (size_t)c * H * W;
        __syncthreads();                                   // previous tile's LDS reads are done
        for (int idx = tid; idx < kSsimIn * kSsimIn; idx += 256) {
            const int rr = idx / kSsimIn, cc = idx - rr * kSsimIn;
            const int gy = ty * kSsimTile + rr - kSsimHalo, gx = tx * kSsimTile + cc - kSsimHalo;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            s_x[rr][cc] = in ? p1[(size_t)gy * W + gx] : 0.f;
            s_y[rr][cc] = in ? p2[(size_t)gy * W + gx] : 0.f;
        }
        __syncthreads();
        for (int idx = tid; idx < kSsimIn * kSsimTile; idx += 256) {        // horizontal pass
            const int rr = idx >> 4, cx = idx & 15;
            float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float x = s_x[rr][cx + k], y = s_y[rr][cx + k], wk = w[k];
                a += wk * x; b += wk * y; aa += wk * x * x; bb += wk * y * y; ab += wk * x * y;
            }
            s_h[0][rr][cx] = a; s_h[1][rr][cx] = b; s_h[2][rr][cx] = aa; s_h[3][rr][cx] = bb; s_h[4][rr][cx] = ab;
        }
        __syncthreads();
        float mu1 = 0.f, mu2 = 0.f, q1 = 0.f, q2 = 0.f, q12 = 0.f;           // vertical pass
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float wk = w[k];
            mu1 += wk * s_h[0][py + k][px]; mu2 += wk * s_h[1][py + k][px]; q1 += wk * s_h[2][py + k][px];
            q2 += wk * s_h[3][py + k][px]; q12 += wk * s_h[4][py + k][px];
        }
        const int gy = ty * kSsimTile + py, gx = tx * kSsimTile + px;
        if (gy < H && gx < W) {
            const float s1 = q1 - mu1 * mu1, s2 = q2 - mu2 * mu2, s12 = q12 - mu1 * mu2;
            const float A = 2.f * mu1 * mu2 + kSsimC1, B = 2.f * s12 + kSsimC2;
            const float Dd = mu1 * mu1 + mu2 * mu2 + kSsimC1, E = s1 + s2 + kSsimC2;
            const float iD = 1.0f / Dd, iE = 1.0f / E;
            const float val = A * B * iD * iE;
            local += val;
            local_l1 += fabsf(s_x[py + kSsimHalo][px + kSsimHalo] - s_y[py + kSsimHalo][px + kSsimHalo]);   // fused L1
            if (Dm) {
                // partials w.r.t. the five convolved maps (s1, s12 depend on mu1 through -mu1^2, -mu1 mu2)
                const float d_s1 = -val * iE;                              // d/d s1   (= d/d q1)
                const float d_s12 = 2.f * A * iD * iE;                     // d/d s12  (= d/d q12)
                const float d_mu1 = 2.f * mu2 * B * iD * iE - 2.f * mu1 * val * iD;
                const size_t o = (size_t)c * H * W + (size_t)gy * W + gx;
                Dm[o] = d_mu1 - 2.f * mu1 * d_s1 - mu2 * d_s12;
                Dq1[o] = d_s1;
                Dq12[o] = d_s12;
            }
        }
    }
    local = wave_sum_loss(local);
    local_l1 = wave_sum_loss(local_l1);
    if ((tid & 63) == 0) { s_part[tid >> 6] = local; s_part[4 + (tid >> 6)] = local_l1; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(out, (s_part[0] + s_part[1] + s_part[2] + s_part[3]) * inv_n);
        if (out_l1) atomicAdd(out_l1, (s_part[4] + s_part[5] + s_part[6] + s_part[7]) * inv_n);
    }
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(int C, int H, int W, int tiles_x, int tiles_y,
                                                       const float *__restrict__ img1, const float *__restrict__ img2,
                                                       const float *__restrict__ Dm, const float *__restrict__ Dq1,
                                                       const float *__restrict__ Dq12, const float *__restrict__ g,
                                                       const float *__restrict__ g_l1, float inv_n,
                                                       float *__restrict__ grad1) {
    __shared__ float s_in[3][kSsimIn][kSsimIn + 1];
    __shared__ float s_h[3][kSsimIn][kSsimTile + 1];
    const int tid = threadIdx.x, px = tid & 15, py = tid >> 4;
    const int ntiles = C * tiles_x * tiles_y;
    const float scale = g[0] * inv_n;
    const float scale_l1 = g_l1 ? g_l1[0] * inv_n : 0.f;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) w[k] = c_ssim_w[k];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int c = t / (tiles_x * tiles_y), r = t - c * tiles_x * tiles_y;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const size_t plane = (size_t)c * H * W;
        __syncthreads();
        for (int idx = tid; idx < kSsimIn * kSsimIn; idx += 256) {
            const int rr = idx / kSsimIn, cc = idx - rr * kSsimIn;
            const int gy = ty * kSsimTile + rr - kSsimHalo, gx = tx * kSsimTile + cc - kSsimHalo;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = plane + (size_t)gy * W + gx;
            s_in[0][rr][cc] = in ? Dm[o] : 0.f;
            s_in[1][rr][cc] = in ? Dq1[o] : 0.f;
            s_in[2][rr][cc] = in ? Dq12[o] : 0.f;
        }
        __syncthreads();
        for (int idx = tid; idx < kSsimIn * kSsimTile; idx += 256) {
            const int rr = idx >> 4, cx = idx & 15;
            float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float wk = w[k];
                a += wk * s_in[0][rr][cx + k]; b += wk * s_in[1][rr][cx + k]; d += wk * s_in[2][rr][cx + k];
            }
            s_h[0][rr][cx] = a; s_h[1][rr][cx] = b; s_h[2][rr][cx] = d;
        }
        __syncthreads();
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float wk = w[k];
            a += wk * s_h[0][py + k][px]; b += wk * s_h[1][py + k][px]; d += wk * s_h[2][py + k][px];
        }
        const int gy = ty * kSsimTile + py, gx = tx * kSsimTile + px;
        if (gy < H && gx < W) {
            const size_t o = plane + (size_t)gy * W + gx;
            const float x = img1[o], y = img2[o], df = x - y;
            grad1[o] = scale * (a + 2.f * x * b + y * d) + scale_l1 * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
        }
    }
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

static inline int ssim_grid(int ntiles, int cap) { return ntiles < 1 ? 1 : (ntiles > cap ? cap : ntiles); }

extern "C" int d3ga_ssim_l1_fwd(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *out,
                                float *Dm, float *Dq1, float *Dq12, float *out_l1, d3ga_stream_t stream) {
    if (C <= 0 || H <= 0 || W <= 0) return D3GA_E_SIZE;
    if (!img1 || !img2 || !out) return D3GA_E_NULL;
    if ((Dm != nullptr) != (Dq1 != nullptr) || (Dm != nullptr) != (Dq12 != nullptr)) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    D3GA_HIP(hipMemsetAsync(out, 0, sizeof(float), s));
    if (out_l1) D3GA_HIP(hipMemsetAsync(out_l1, 0, sizeof(float), s));
    const int tx = (W + kSsimTile - 1) / kSsimTile, ty = (H + kSsimTile - 1) / kSsimTile;
    // persistent grid: every workgroup ends with ONE atomic on the result word (same-address atomics serialise)
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(ssim_grid(C * tx * ty, 2048)), dim3(256), 0, s, C, H, W, tx, ty, img1, img2,
                       1.0f / ((float)C * (float)H * (float)W), out, Dm, Dq1, Dq12, out_l1);
    return check_launch(s, 0);
}

extern "C" int d3ga_ssim_fwd(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *out,
                             float *Dm, float *Dq1, float *Dq12, d3ga_stream_t stream) {
    return d3ga_ssim_l1_fwd(C, H, W, img1, img2, out, Dm, Dq1, Dq12, nullptr, stream);
}

extern "C" int d3ga_ssim_l1_bwd(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, const float *Dm,
                                const float *Dq1, const float *Dq12, const float *g, const float *g_l1,
                                float *grad_img1, d3ga_stream_t stream) {
    if (C <= 0 || H <= 0 || W <= 0) return D3GA_E_SIZE;
    if (!img1 || !img2 || !Dm || !Dq1 || !Dq12 || !g || !grad_img1) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    const int tx = (W + kSsimTile - 1) / kSsimTile, ty = (H + kSsimTile - 1) / kSsimTile;
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3(ssim_grid(C * tx * ty, 8192)), dim3(256), 0, s, C, H, W, tx, ty, img1, img2,
                       Dm, Dq1, Dq12, g, g_l1, 1.0f / ((float)C * (float)H * (float)W), grad_img1);
    return check_launch(s, 0);
}

extern "C" int d3ga_ssim_bwd(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, const float *Dm,
                             const float *Dq1, const float *Dq12, const float *g, float *grad_img1,
                             d3ga_stream_t stream) {
    return d3ga_ssim_l1_bwd(C, H, W, img1, img2, Dm, Dq1, Dq12, g, nullptr, grad_img1, stream);
}

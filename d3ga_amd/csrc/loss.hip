// loss.hip -- the loss tail next to the render boundary.
//   l1_mean_*   fused L1 image loss: mean |a - b| and its gradient.  Replaces utils/loss_utils.py:29
//               (torch.abs(network_output - gt).mean()), which autograd runs as six full-image ATen kernels.
//   ssim_*      fused 11x11 Gaussian-window SSIM (utils/loss_utils.py:46-86, called at train.py:192): the reference runs
//               five depthwise conv2d (11x11, zero padding) plus ~15 elementwise kernels forward and their autograd
//               backward; here one LDS-tiled separable pass each way.
#include "d3ga_internal.h"

#include <stdlib.h>

namespace d3ga {

__device__ __forceinline__ float wave_sum_loss(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// sum |a - b| of this workgroup's grid-stride share, times inv_n (every thread returns; thread 0 holds the value)
__device__ __forceinline__ float l1_block_sum(int64_t n4, int64_t n, const float *__restrict__ a, const float *__restrict__ b,
                                              float inv_n, float *s_part) {
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
    float t = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) t += s_part[w];
    }
    return t * inv_n;
}

__global__ __launch_bounds__(kBlock) void l1_mean_fwd_kernel(int64_t n4, int64_t n, const float *__restrict__ a,
                                                             const float *__restrict__ b, float inv_n,
                                                             float *__restrict__ out) {
    __shared__ float s_part[kBlock / 64];
    const float t = l1_block_sum(n4, n, a, b, inv_n, s_part);
    if (threadIdx.x == 0) atomicAdd(out, t);
}

// Two-stage form: one partial per workgroup (plain store), then ONE workgroup adds the partials in index order -- no zero
// fill, no same-address atomics (2048 of them serialise for ~25 us), and the result does not depend on arrival order.
// b_cell (optional): the address of `b` is read from device memory -- a captured step is pointed at another resident
// target by rewriting that cell (d3ga_amd/graph.py: TensorSlot), without copying the image into a static buffer
__global__ __launch_bounds__(kBlock) void l1_mean_partial_kernel(int64_t n4, int64_t n, const float *__restrict__ a,
                                                                 const float *__restrict__ b, const float *const *b_cell,
                                                                 float inv_n, float *__restrict__ partials) {
    __shared__ float s_part[kBlock / 64];
    if (b_cell) b = *b_cell;
    const float t = l1_block_sum(n4, n, a, b, inv_n, s_part);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(int np, const float *__restrict__ partials, float *__restrict__ out) {
    __shared__ float s_part[kBlock / 64];
    float acc = 0.f;
    for (int i = threadIdx.x; i < np; i += kBlock) acc += partials[i];
    acc = wave_sum_loss(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) t += s_part[w];
        out[0] = t;
    }
}

// the same for many partials (one per quadrant of the compositing forward: 4 x tiles): 1024 threads, 16-byte loads
__global__ __launch_bounds__(1024) void sum_partials_wide_kernel(int np, const float *__restrict__ partials, float *__restrict__ out) {
    __shared__ float s_part[16];
    float acc = 0.f;
    const int np4 = np >> 2;
    // eight independent 16-byte loads in flight per thread (a 4K frame has 130 k partials: 32 DEPENDENT load -> add trips per
    // thread made this launch 14.8 us at C5; 1080p: 8 trips, 4.7 us = the floor of any launch)
    int i = threadIdx.x;
    for (; i + 7 * 1024 < np4; i += 8 * 1024) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4 *>(partials)[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    for (; i < np4; i += 1024) {
        const float4 v = reinterpret_cast<const float4 *>(partials)[i];
        acc += (v.x + v.y) + (v.z + v.w);
    }
    for (int j = 4 * np4 + threadIdx.x; j < np; j += 1024) acc += partials[j];
    acc = wave_sum_loss(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_part[w];
        out[0] = t;
    }
}
void launch_sum_partials(int np, const float *partials, float *out, hipStream_t s) {
    if (np > 4096) hipLaunchKernelGGL(sum_partials_wide_kernel, dim3(1), dim3(1024), 0, s, np, partials, out);
    else hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, s, np, partials, out);
}

// (Round 3 also tried ONE launch -- the last workgroup to arrive, found through an arrival counter, adds the partials: the
// ~2000 same-address counter atomics serialise at the memory side, 34 us against 10 + 4.6 us for the two launches; removed.)
__global__ __launch_bounds__(kBlock) void l1_mean_bwd_kernel(int64_t n4, int64_t n, const float *__restrict__ a,
                                                             const float *__restrict__ b, const float *const *b_cell,
                                                             const float *__restrict__ g, float inv_n,
                                                             float *__restrict__ grad_a) {
    if (b_cell) b = *b_cell;
    const float s = g[0] * inv_n;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
    auto sgn4 = [&](const float4 &x, const float4 &y) {
        return make_float4(s * sgn(x.x - y.x), s * sgn(x.y - y.y), s * sgn(x.z - y.z), s * sgn(x.w - y.w));
    };
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {             // two independent 16-byte loads per array in flight
        const float4 x0 = reinterpret_cast<const float4 *>(a)[i], y0 = reinterpret_cast<const float4 *>(b)[i];
        const float4 x1 = reinterpret_cast<const float4 *>(a)[i + stride], y1 = reinterpret_cast<const float4 *>(b)[i + stride];
        reinterpret_cast<float4 *>(grad_a)[i] = sgn4(x0, y0);
        reinterpret_cast<float4 *>(grad_a)[i + stride] = sgn4(x1, y1);
    }
    for (; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i], y = reinterpret_cast<const float4 *>(b)[i];
        reinterpret_cast<float4 *>(grad_a)[i] = sgn4(x, y);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        grad_a[i] = s * sgn(a[i] - b[i]);
}


// ---------------------------------------------------------------------------------------------------------
// SSIM.  ssim_map = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)) with mu = w*x, s1 = w*x^2 - mu1^2,
// s12 = w*xy - mu1 mu2 and w the 11x11 window gaussian(11, 1.5) (x) gaussian(11, 1.5) (utils/loss_utils.py:46-57),
// zero padding of 5 (F.conv2d(..., padding=window_size // 2)).  The window is separable: a workgroup owns a 16x32
// output tile of one channel, stages the (16+10)x(32+10) halo of both images in LDS, convolves the five maps
// (x, y, x^2, y^2, xy) horizontally into LDS and vertically into registers.  The forward also stores the three
// partial derivatives the backward needs (d ssim / d(w*x), d(w*x^2), d(w*xy)); the backward convolves those three maps
// with the same (symmetric) window:  dL/dx = w*Dm + 2 x (w*Dq1) + y (w*Dq12).
// HBM traffic per pixel and channel: forward 8 B read + 12 B written, backward 20 B read + 4 B written.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSsimTW = 16, kSsimTH = 32, kSsimHalo = 5;               // output tile 16 wide x 32 high per workgroup
constexpr int kSsimInW = kSsimTW + 2 * kSsimHalo, kSsimInH = kSsimTH + 2 * kSsimHalo;   // 26 x 42 input halo
constexpr int kSsimXP = 28;                                            // floats per input row in LDS (16-byte aligned rows)
constexpr int kSsimHP = 44;                                            // floats per COLUMN of the transposed horizontal result
using f2 = __attribute__((ext_vector_type(2))) float;                  // two pixels per lane: v_pk_fma_f32 / v_pk_mul_f32
__device__ __constant__ float c_ssim_w[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                              2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                              3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};
constexpr float kSsimC1 = 0.01f * 0.01f, kSsimC2 = 0.03f * 0.03f;

// Register tiling.  The first version gave every thread one output pixel and read every tap from LDS (90 ds_read_b32 per
// pixel): the LDS pipe, not the VALU, was the limit.  Now
//   horizontal pass: one thread = FOUR adjacent outputs of a row: 16 input floats per image come in with four 16-byte LDS
//                    reads, the products x^2, y^2, xy are formed once per input, the taps run on registers; the
//                    results are written TRANSPOSED (s_h[map][column][row]);
//   vertical pass:   one thread = TWO vertically adjacent outputs of a column: 12 consecutive rows per map = six 8-byte
//                    LDS reads, both outputs accumulated as one 2-vector (packed f32 instructions).
// LDS instructions per output pixel: 0.45 instead of 3.

// 11-tap window over 14 consecutive samples -> 4 outputs (o = 0..3 uses samples o..o+10)
__device__ __forceinline__ void taps4(const float (&v)[16], const float (&w)[11], float (&o)[4]) {
    f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const f2 wk = {w[k], w[k]};
        a01 += wk * (f2){v[k], v[k + 1]};
        a23 += wk * (f2){v[k + 2], v[k + 3]};
    }
    o[0] = a01.x; o[1] = a01.y; o[2] = a23.x; o[3] = a23.y;
}
__device__ __forceinline__ void load16(const float *row, float (&v)[16]) {      // row: 16-byte aligned
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 t = reinterpret_cast<const float4 *>(row)[j];
        v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    }
}
// two vertically adjacent outputs (rows r, r+1; r even) of one column: samples r..r+11 of the transposed map
__device__ __forceinline__ f2 vtaps2(const float *col, int r, const float (&w)[11]) {
    float v[12];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float2 t = reinterpret_cast<const float2 *>(col + r)[j];
        v[2 * j] = t.x; v[2 * j + 1] = t.y;
    }
    f2 a = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) a += (f2){w[k], w[k]} * (f2){v[k], v[k + 1]};
    return a;
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(int C, int H, int W, int tiles_x, int tiles_y,
                                                       const float *__restrict__ img1, const float *__restrict__ img2,
                                                       float inv_n, float *__restrict__ out, float *__restrict__ Dm,
                                                       float *__restrict__ Dq1, float *__restrict__ Dq12,
                                                       float *__restrict__ out_l1) {
    __shared__ __attribute__((aligned(16))) float s_x[kSsimInH][kSsimXP], s_y[kSsimInH][kSsimXP];
    __shared__ __attribute__((aligned(16))) float s_h[5][kSsimTW][kSsimHP];
    __shared__ float s_part[8];
    const int tid = threadIdx.x;
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
        for (int idx = tid; idx < kSsimInH * kSsimXP; idx += 256) {         // (the two pad columns are zero-filled too)
            const int rr = idx / kSsimXP, cc = idx - rr * kSsimXP;
            const int gy = ty * kSsimTH + rr - kSsimHalo, gx = tx * kSsimTW + cc - kSsimHalo;
            const bool in = cc < kSsimInW && gy >= 0 && gy < H && gx >= 0 && gx < W;
            s_x[rr][cc] = in ? p1[(size_t)gy * W + gx] : 0.f;
            s_y[rr][cc] = in ? p2[(size_t)gy * W + gx] : 0.f;
        }
        __syncthreads();
        if (tid < kSsimInH * (kSsimTW / 4)) {              // horizontal pass: 42 rows x 4 column quads
            const int rr = tid >> 2, c0 = 4 * (tid & 3);
            float x[16], y[16], p[16], o[4];
            load16(&s_x[rr][c0], x);
            load16(&s_y[rr][c0], y);
            taps4(x, w, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) s_h[0][c0 + q][rr] = o[q];
            taps4(y, w, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) s_h[1][c0 + q][rr] = o[q];
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = x[i] * x[i];
            taps4(p, w, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) s_h[2][c0 + q][rr] = o[q];
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = y[i] * y[i];
            taps4(p, w, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) s_h[3][c0 + q][rr] = o[q];
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = x[i] * y[i];
            taps4(p, w, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) s_h[4][c0 + q][rr] = o[q];
        }
        __syncthreads();
        // vertical pass: thread = column px, output rows py2 and py2 + 1
        const int px = tid & 15, py2 = 2 * (tid >> 4);
        const f2 mu1 = vtaps2(s_h[0][px], py2, w), mu2 = vtaps2(s_h[1][px], py2, w);
        const f2 q1 = vtaps2(s_h[2][px], py2, w), q2 = vtaps2(s_h[3][px], py2, w), q12 = vtaps2(s_h[4][px], py2, w);
        const f2 s1 = q1 - mu1 * mu1, s2 = q2 - mu2 * mu2, s12 = q12 - mu1 * mu2;
        const f2 A = 2.f * mu1 * mu2 + kSsimC1, B = 2.f * s12 + kSsimC2;
        const f2 Dd = mu1 * mu1 + mu2 * mu2 + kSsimC1, E = s1 + s2 + kSsimC2;
        const f2 iD = {1.0f / Dd.x, 1.0f / Dd.y}, iE = {1.0f / E.x, 1.0f / E.y};
        const f2 val = A * B * iD * iE;
        // partials w.r.t. the five convolved maps (s1, s12 depend on mu1 through -mu1^2, -mu1 mu2)
        const f2 d_s1 = -val * iE;                                          // d/d s1   (= d/d q1)
        const f2 d_s12 = 2.f * A * iD * iE;                                 // d/d s12  (= d/d q12)
        const f2 d_mu1 = 2.f * mu2 * B * iD * iE - 2.f * mu1 * val * iD;
        const f2 dm = d_mu1 - 2.f * mu1 * d_s1 - mu2 * d_s12;
        const int gx = tx * kSsimTW + px;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int gy = ty * kSsimTH + py2 + h;
            if (gy < H && gx < W) {
                local += h ? val.y : val.x;
                local_l1 += fabsf(s_x[py2 + h + kSsimHalo][px + kSsimHalo] - s_y[py2 + h + kSsimHalo][px + kSsimHalo]);   // fused L1
                if (Dm) {
                    const size_t o = (size_t)c * H * W + (size_t)gy * W + gx;
                    Dm[o] = h ? dm.y : dm.x;
                    Dq1[o] = h ? d_s1.y : d_s1.x;
                    Dq12[o] = h ? d_s12.y : d_s12.x;
                }
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
    __shared__ __attribute__((aligned(16))) float s_in[3][kSsimInH][kSsimXP];
    __shared__ __attribute__((aligned(16))) float s_h[3][kSsimTW][kSsimHP];
    const int tid = threadIdx.x;
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
        for (int idx = tid; idx < kSsimInH * kSsimXP; idx += 256) {
            const int rr = idx / kSsimXP, cc = idx - rr * kSsimXP;
            const int gy = ty * kSsimTH + rr - kSsimHalo, gx = tx * kSsimTW + cc - kSsimHalo;
            const bool in = cc < kSsimInW && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = plane + (size_t)gy * W + gx;
            s_in[0][rr][cc] = in ? Dm[o] : 0.f;
            s_in[1][rr][cc] = in ? Dq1[o] : 0.f;
            s_in[2][rr][cc] = in ? Dq12[o] : 0.f;
        }
        __syncthreads();
        if (tid < kSsimInH * (kSsimTW / 4)) {
            const int rr = tid >> 2, c0 = 4 * (tid & 3);
            float v[16], o[4];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                load16(&s_in[m][rr][c0], v);
                taps4(v, w, o);
#pragma unroll
                for (int q = 0; q < 4; ++q) s_h[m][c0 + q][rr] = o[q];
            }
        }
        __syncthreads();
        const int px = tid & 15, py2 = 2 * (tid >> 4);
        const f2 a = vtaps2(s_h[0][px], py2, w), b = vtaps2(s_h[1][px], py2, w), d = vtaps2(s_h[2][px], py2, w);
        const int gx = tx * kSsimTW + px;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int gy = ty * kSsimTH + py2 + h;
            if (gy < H && gx < W) {
                const size_t o = plane + (size_t)gy * W + gx;
                const float x = img1[o], y = img2[o], df = x - y;
                const float aa = h ? a.y : a.x, bb = h ? b.y : b.x, dd = h ? d.y : d.x;
                grad1[o] = scale * (aa + 2.f * x * bb + y * dd) + scale_l1 * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Round 4: the same SSIM as a MARCH down the image (ssim_march_*_kernel).
//
// The tiled kernels above re-load a (16+10) x (32+10) halo per 16 x 32 outputs (2.1x the pixels), idle a third of their
// threads in the horizontal pass and synchronise three times per tile: 123 + 63 us at 1080p against ~25 us of HBM traffic.
// Here a workgroup owns a vertical strip: thread = one input COLUMN (256 columns, 244 of them outputs + the 5-pixel halo on
// both sides + 2 idle), and it walks down a segment of output rows (ssim_march_rows):
//   A. vertical pass in REGISTERS: every thread keeps the last 11 rows of its column (x, y -- or the three derivative maps
//      of the backward) in a register ring and forms the vertically convolved maps of the finished row (the products x^2,
//      y^2, xy are formed on the fly: 7 instructions per tap for five maps) -- no LDS, every input pixel is loaded once per
//      strip (vertical halo: 10 rows per segment);
//   B. the convolved rows go to LDS, four rows at a time;
//   C. horizontal pass + the per-pixel formulas: thread = (row of the four, 4 adjacent output columns): four 16-byte LDS
//      reads per map feed 4 x 11 taps, the SSIM terms / the gradient of those four pixels are formed and stored (16-byte
//      stores when the image allows).
// HBM traffic per pixel and channel: forward 8 B x (1 + 10 / rows) + 12 B, backward 12 B x (1 + 10 / rows) + 8 B + 4 B.
// ---------------------------------------------------------------------------------------------------------
constexpr int kMarchW = 256;                 // threads = input columns of a strip
constexpr int kMarchUse = 244;               // output columns of a strip (61 groups of 4); kMarchUse + 10 <= kMarchW
constexpr int kMarchGroups = kMarchUse / 4;
constexpr int kMarchLd = 256;                // floats per LDS row (group 60 reads columns 240..255: nothing runs past the row)

template <int NMAP>
__device__ __forceinline__ void march_htaps(const float *row, int g, const float (&w)[11], float (&o)[4]) {
    float v[16];
    load16(row + 4 * g, v);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        a0 = fmaf(w[k], v[k], a0); a1 = fmaf(w[k], v[k + 1], a1); a2 = fmaf(w[k], v[k + 2], a2); a3 = fmaf(w[k], v[k + 3], a3);
    }
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    __builtin_amdgcn_sched_barrier(0);      // one map at a time: left alone the scheduler issues the LDS reads of all maps up front (196 VGPRs)
}

__global__ __launch_bounds__(kMarchW, 4) void ssim_march_fwd_kernel(int C, int H, int W, int strips, int segs, int seg_rows,
                                                                  const float *__restrict__ img1, const float *__restrict__ img2,
                                                                  float inv_n, float *__restrict__ out, float *__restrict__ Dm,
                                                                  float *__restrict__ Dq1, float *__restrict__ Dq12,
                                                                  float *__restrict__ out_l1, int vec_ok) {
    __shared__ __attribute__((aligned(16))) float s_v[8][5][kMarchLd];      // two buffers of four rows: ONE barrier per group (40 KB: four workgroups per CU)
    float *const s_part = &s_v[0][0][0];                   // (the loss partials reuse it at the very end)
    const int tid = threadIdx.x;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) w[k] = c_ssim_w[k];
    float local = 0.f, local_l1 = 0.f;
    const int nitems = C * strips * segs;
    if constexpr (kMarchLd > kMarchW) {                    // pad columns behind the kMarchW written ones (none since kMarchLd == kMarchW): zero once
        constexpr int pad = kMarchLd > kMarchW ? kMarchLd - kMarchW : 1;
        if (tid < 5 * 8 * pad) {
            const int r = tid / (5 * pad), q = tid % (5 * pad);
            s_v[r][q / pad][kMarchW + q % pad] = 0.f;
        }
    }
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int c = item / (strips * segs), r = item - c * strips * segs;
        const int sy = r / strips, sx = r - sy * strips;
        const float *p1 = img1 + (size_t)c * H * W, *p2 = img2 + (size_t)c * H * W;
        const size_t plane = (size_t)c * H * W;
        const int gx = sx * kMarchUse - kSsimHalo + tid;   // this thread's input column
        const bool col_in = gx >= 0 && gx < W;
        const bool col_out = tid >= kSsimHalo && tid < kSsimHalo + kMarchUse && gx < W;      // an output column of this strip
        const int y0 = sy * seg_rows;
        const int nout = min(seg_rows, H - y0);
        // a window of 14 input rows of this column in registers: FOUR output rows per group (rows i .. i + 10, i = 0..3), then the
        // window moves down by four (10 register moves per map and group instead of 10 per ROW for an 11-row ring) and the four
        // rows loaded meanwhile take the free places
        float rx[14], ry[14];
        auto load_row = [&](int ri, float &x, float &y) {  // input row ri of the segment: image row y0 - 5 + ri
            const int gy = y0 - kSsimHalo + ri;
            const bool in = col_in && gy >= 0 && gy < H;
            x = in ? p1[(size_t)gy * W + gx] : 0.f;
            y = in ? p2[(size_t)gy * W + gx] : 0.f;
        };
#pragma unroll
        for (int k = 0; k < 14; ++k) load_row(k, rx[k], ry[k]);
        float nx[4], ny[4];
        const int ngroups = (nout + 3) >> 2;
        for (int grp = 0; grp < ngroups; ++grp) {
            const int orow0 = 4 * grp, slot = min(3, nout - 1 - orow0), buf = (grp & 1) << 2;      // slot: last valid row of the group
            const int orow = orow0 + slot;
            if (grp + 1 < ngroups) {
#pragma unroll
                for (int j = 0; j < 4; ++j) load_row(orow0 + 14 + j, nx[j], ny[j]);    // the next group's new rows: in flight during this group
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sx1 = 0.f, sy1 = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const float wx = w[k] * rx[i + k], wy = w[k] * ry[i + k];
                    sx1 += wx; sy1 += wy;
                    sxx = fmaf(wx, rx[i + k], sxx); syy = fmaf(wy, ry[i + k], syy); sxy = fmaf(wx, ry[i + k], sxy);
                }
                s_v[buf + i][0][tid] = sx1; s_v[buf + i][1][tid] = sy1; s_v[buf + i][2][tid] = sxx; s_v[buf + i][3][tid] = syy; s_v[buf + i][4][tid] = sxy;
                if (col_out && i <= slot) local_l1 += fabsf(rx[i + 5] - ry[i + 5]);           // fused L1
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) { rx[k] = rx[k + 4]; ry[k] = ry[k + 4]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { rx[10 + j] = nx[j]; ry[10 + j] = ny[j]; }
            __syncthreads();
            // ---- horizontal pass + SSIM of up to four rows: thread = (row slot, 4 adjacent output columns) ----
            if (tid < 4 * kMarchGroups) {
                const int rs = tid / kMarchGroups, g = tid - rs * kMarchGroups;
                const int gy = y0 + (orow & ~3) + rs, ox = sx * kMarchUse + 4 * g;
                if (rs <= slot && ox < W) {
                    float mu1[4], mu2[4], q1[4], q2[4], q12[4];
                    march_htaps<5>(s_v[buf + rs][0], g, w, mu1);
                    march_htaps<5>(s_v[buf + rs][1], g, w, mu2);
                    march_htaps<5>(s_v[buf + rs][2], g, w, q1);
                    march_htaps<5>(s_v[buf + rs][3], g, w, q2);
                    march_htaps<5>(s_v[buf + rs][4], g, w, q12);
                    float dm[4], ds1[4], ds12[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float s1 = q1[j] - mu1[j] * mu1[j], s2 = q2[j] - mu2[j] * mu2[j], s12 = q12[j] - mu1[j] * mu2[j];
                        const float A = 2.f * mu1[j] * mu2[j] + kSsimC1, B = 2.f * s12 + kSsimC2;
                        const float Dd = mu1[j] * mu1[j] + mu2[j] * mu2[j] + kSsimC1, E = s1 + s2 + kSsimC2;
                        const float iD = __builtin_amdgcn_rcpf(Dd), iE = __builtin_amdgcn_rcpf(E);      // (1 ulp; a full-precision divide is ~10 instructions, eight of them per thread and group)
                        const float val = A * B * iD * iE;
                        ds1[j] = -val * iE;
                        ds12[j] = 2.f * A * iD * iE;
                        const float d_mu1 = 2.f * mu2[j] * B * iD * iE - 2.f * mu1[j] * val * iD;
                        dm[j] = d_mu1 - 2.f * mu1[j] * ds1[j] - mu2[j] * ds12[j];
                        if (ox + j < W) local += val;
                    }
                    if (Dm) {
                        const size_t o = plane + (size_t)gy * W + ox;
                        if (vec_ok && ox + 3 < W) {
                            *reinterpret_cast<float4 *>(Dm + o) = make_float4(dm[0], dm[1], dm[2], dm[3]);
                            *reinterpret_cast<float4 *>(Dq1 + o) = make_float4(ds1[0], ds1[1], ds1[2], ds1[3]);
                            *reinterpret_cast<float4 *>(Dq12 + o) = make_float4(ds12[0], ds12[1], ds12[2], ds12[3]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (ox + j < W) { Dm[o + j] = dm[j]; Dq1[o + j] = ds1[j]; Dq12[o + j] = ds12[j]; }
                        }
                    }
                }
            }
            // (no second barrier: the next group writes the OTHER buffer, and its own barrier orders this group's reads
            // before the writes of the group after it)
        }
        __syncthreads();                                  // (next item: both buffers are rewritten from row 0 on)
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

__global__ __launch_bounds__(kMarchW) void ssim_march_bwd_kernel(int C, int H, int W, int strips, int segs, int seg_rows,
                                                                  const float *__restrict__ img1, const float *__restrict__ img2,
                                                                  const float *__restrict__ Dm, const float *__restrict__ Dq1,
                                                                  const float *__restrict__ Dq12, const float *__restrict__ g,
                                                                  const float *__restrict__ g_l1, float inv_n,
                                                                  float *__restrict__ grad1, int vec_ok) {
    __shared__ __attribute__((aligned(16))) float s_v[8][3][kMarchLd];
    const int tid = threadIdx.x;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) w[k] = c_ssim_w[k];
    const float scale = g[0] * inv_n;
    const float scale_l1 = g_l1 ? g_l1[0] * inv_n : 0.f;
    const int nitems = C * strips * segs;
    if constexpr (kMarchLd > kMarchW) {
        constexpr int pad = kMarchLd > kMarchW ? kMarchLd - kMarchW : 1;
        if (tid < 3 * 8 * pad) {
            const int r = tid / (3 * pad), q = tid % (3 * pad);
            s_v[r][q / pad][kMarchW + q % pad] = 0.f;
        }
    }
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int c = item / (strips * segs), r = item - c * strips * segs;
        const int sy = r / strips, sx = r - sy * strips;
        const size_t plane = (size_t)c * H * W;
        const int gx = sx * kMarchUse - kSsimHalo + tid;
        const bool col_in = gx >= 0 && gx < W;
        const int y0 = sy * seg_rows;
        const int nout = min(seg_rows, H - y0);
        float ra[14], rb[14], rd[14];
        auto load_row = [&](int ri, float &a, float &b, float &d) {
            const int gy = y0 - kSsimHalo + ri;
            const bool in = col_in && gy >= 0 && gy < H;
            const size_t o = plane + (size_t)gy * W + gx;
            a = in ? Dm[o] : 0.f; b = in ? Dq1[o] : 0.f; d = in ? Dq12[o] : 0.f;
        };
#pragma unroll
        for (int k = 0; k < 14; ++k) load_row(k, ra[k], rb[k], rd[k]);
        float na[4], nb[4], nd[4];
        const int ngroups = (nout + 3) >> 2;
        for (int grp = 0; grp < ngroups; ++grp) {
            const int orow0 = 4 * grp, slot = min(3, nout - 1 - orow0), buf = (grp & 1) << 2;
            const int orow = orow0 + slot;
            if (grp + 1 < ngroups) {
#pragma unroll
                for (int j = 0; j < 4; ++j) load_row(orow0 + 14 + j, na[j], nb[j], nd[j]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sa = 0.f, sb = 0.f, sd = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) { sa = fmaf(w[k], ra[i + k], sa); sb = fmaf(w[k], rb[i + k], sb); sd = fmaf(w[k], rd[i + k], sd); }
                s_v[buf + i][0][tid] = sa; s_v[buf + i][1][tid] = sb; s_v[buf + i][2][tid] = sd;
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) { ra[k] = ra[k + 4]; rb[k] = rb[k + 4]; rd[k] = rd[k + 4]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { ra[10 + j] = na[j]; rb[10 + j] = nb[j]; rd[10 + j] = nd[j]; }
            __syncthreads();
            if (tid < 4 * kMarchGroups) {
                const int rs = tid / kMarchGroups, gq = tid - rs * kMarchGroups;
                const int gy = y0 + (orow & ~3) + rs, ox = sx * kMarchUse + 4 * gq;
                if (rs <= slot && ox < W) {
                    float a[4], b[4], d[4];
                    march_htaps<3>(s_v[buf + rs][0], gq, w, a);
                    march_htaps<3>(s_v[buf + rs][1], gq, w, b);
                    march_htaps<3>(s_v[buf + rs][2], gq, w, d);
                    const size_t o = plane + (size_t)gy * W + ox;
                    float x[4], y[4], outv[4];
                    if (vec_ok && ox + 3 < W) {
                        const float4 xv = *reinterpret_cast<const float4 *>(img1 + o), yv = *reinterpret_cast<const float4 *>(img2 + o);
                        x[0] = xv.x; x[1] = xv.y; x[2] = xv.z; x[3] = xv.w; y[0] = yv.x; y[1] = yv.y; y[2] = yv.z; y[3] = yv.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { x[j] = ox + j < W ? img1[o + j] : 0.f; y[j] = ox + j < W ? img2[o + j] : 0.f; }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float df = x[j] - y[j];
                        outv[j] = scale * (a[j] + 2.f * x[j] * b[j] + y[j] * d[j]) + scale_l1 * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
                    }
                    if (vec_ok && ox + 3 < W) *reinterpret_cast<float4 *>(grad1 + o) = make_float4(outv[0], outv[1], outv[2], outv[3]);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (ox + j < W) grad1[o + j] = outv[j];
                    }
                }
            }
            // (no second barrier: the next group writes the OTHER buffer, and its own barrier orders this group's reads
            // before the writes of the group after it)
        }
        __syncthreads();                                  // (next item: both buffers are rewritten from row 0 on)
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
    D3GA_HIP(zero_async(out, sizeof(float), s));
    const int64_t n4 = n / 4;
    // few, fat workgroups: every workgroup ends with ONE float atomic on the same word, and same-address device-scope
    // atomics serialise at ~12 ns each (2048 of them cost more than streaming the two images)
    hipLaunchKernelGGL(l1_mean_fwd_kernel, dim3(loss_grid(n4, 512)), dim3(kBlock), 0, s, n4, n, a, b, 1.0f / (float)n, out);
    return check_launch(s, 0);
}

extern "C" int d3ga_l1_mean_fwd_ws(int64_t n, const float *a, const float *b, float *out, float *partials,
                                   d3ga_stream_t stream) {
    if (n <= 0) return D3GA_E_SIZE;
    if (!a || !b || !out || !partials) return D3GA_E_NULL;
    if (((uintptr_t)a | (uintptr_t)b) & 15) return D3GA_E_CONFIG;       // 16-byte aligned inputs
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = n / 4;
    const int np = loss_grid(n4, D3GA_LOSS_PARTIALS);
    hipLaunchKernelGGL(l1_mean_partial_kernel, dim3(np), dim3(kBlock), 0, s, n4, n, a, b, (const float *const *)nullptr,
                       1.0f / (float)n, partials);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, s, np, (const float *)partials, out);
    return check_launch(s, 0);
}

extern "C" int d3ga_l1_mean_bwd(int64_t n, const float *a, const float *b, const float *g, float *grad_a,
                                d3ga_stream_t stream) {
    if (n <= 0) return D3GA_E_SIZE;
    if (!a || !b || !g || !grad_a) return D3GA_E_NULL;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad_a) & 15) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(l1_mean_bwd_kernel, dim3(loss_grid(n4, 2048)), dim3(kBlock), 0, s, n4, n, a, b,
                       (const float *const *)nullptr, g, 1.0f / (float)n, grad_a);
    return check_launch(s, 0);
}

extern "C" int d3ga_l1_mean_fwd_ws_cell(int64_t n, const float *a, const float *const *b_cell, float *out, float *partials,
                                        d3ga_stream_t stream) {
    if (n <= 0) return D3GA_E_SIZE;
    if (!a || !b_cell || !out || !partials) return D3GA_E_NULL;
    if ((uintptr_t)a & 15) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = n / 4;
    const int np = loss_grid(n4, D3GA_LOSS_PARTIALS);
    hipLaunchKernelGGL(l1_mean_partial_kernel, dim3(np), dim3(kBlock), 0, s, n4, n, a, (const float *)nullptr, b_cell,
                       1.0f / (float)n, partials);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, s, np, (const float *)partials, out);
    return check_launch(s, 0);
}

extern "C" int d3ga_l1_mean_bwd_cell(int64_t n, const float *a, const float *const *b_cell, const float *g, float *grad_a,
                                     d3ga_stream_t stream) {
    if (n <= 0) return D3GA_E_SIZE;
    if (!a || !b_cell || !g || !grad_a) return D3GA_E_NULL;
    if (((uintptr_t)a | (uintptr_t)grad_a) & 15) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(l1_mean_bwd_kernel, dim3(loss_grid(n4, 2048)), dim3(kBlock), 0, s, n4, n, a, (const float *)nullptr,
                       b_cell, g, 1.0f / (float)n, grad_a);
    return check_launch(s, 0);
}

static inline int ssim_grid(int ntiles, int cap) { return ntiles < 1 ? 1 : (ntiles > cap ? cap : ntiles); }
// Rows per strip segment of the marching kernels: a launch runs in ROUNDS of (CUs x resident workgroups) items, an item costs
// (rows + 10) input rows -- pick the multiple of 4 that minimises rounds x (rows + 10) (at 1080p x 3 channels and 768 slots:
// 36 rows = 720 items in one round; 32 rows = 816 items, i.e. a second round for 48 of them: measured 1.6x slower).
static int ssim_march_rows(const void *kernel, int C, int H, int strips) {
    int per_cu = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kMarchW, 0) != hipSuccess || per_cu < 1) per_cu = 3;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    const long slots = (long)per_cu * cus;
    int best = 32;
    double best_cost = 1e30;
    for (int rows = 16; rows <= 128; rows += 4) {
        const long items = (long)C * strips * ((H + rows - 1) / rows);
        const double cost = (double)((items + slots - 1) / slots) * (rows + 2 * kSsimHalo);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = rows; }
    }
    return best;
}
static inline int ssim_impl() { return debug_knob(D3GA_KNOB_SSIM_IMPL); }      // 1 (default) the marching kernels, 0 the LDS-tiled ones of round 3

extern "C" int d3ga_ssim_l1_fwd(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *out,
                                float *Dm, float *Dq1, float *Dq12, float *out_l1, d3ga_stream_t stream) {
    if (C <= 0 || H <= 0 || W <= 0) return D3GA_E_SIZE;
    if (!img1 || !img2 || !out) return D3GA_E_NULL;
    if ((Dm != nullptr) != (Dq1 != nullptr) || (Dm != nullptr) != (Dq12 != nullptr)) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    D3GA_HIP(zero_async(out, sizeof(float), s));
    if (out_l1) D3GA_HIP(zero_async(out_l1, sizeof(float), s));
    if (ssim_impl() == 1) {
        const int strips = (W + kMarchUse - 1) / kMarchUse;
        static int rows_cache[4] = {0, 0, 0, 0};            // (C, H, strips) -> rows: the occupancy query is not free
        if (rows_cache[0] != C || rows_cache[1] != H || rows_cache[2] != strips) {
            rows_cache[3] = ssim_march_rows((const void *)ssim_march_fwd_kernel, C, H, strips);
            rows_cache[0] = C; rows_cache[1] = H; rows_cache[2] = strips;
        }
        const int seg_rows = rows_cache[3], segs = (H + seg_rows - 1) / seg_rows;
        const int vec_ok = (W % 4 == 0) && !(((uintptr_t)Dm | (uintptr_t)Dq1 | (uintptr_t)Dq12) & 15);
        hipLaunchKernelGGL(ssim_march_fwd_kernel, dim3(ssim_grid(C * strips * segs, 4096)), dim3(kMarchW), 0, s, C, H, W, strips, segs, seg_rows,
                           img1, img2, 1.0f / ((float)C * (float)H * (float)W), out, Dm, Dq1, Dq12, out_l1, vec_ok);
        return check_launch(s, 0);
    }
    const int tx = (W + kSsimTW - 1) / kSsimTW, ty = (H + kSsimTH - 1) / kSsimTH;
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
    if (ssim_impl() == 1) {
        const int strips = (W + kMarchUse - 1) / kMarchUse;
        static int rows_cache[4] = {0, 0, 0, 0};
        if (rows_cache[0] != C || rows_cache[1] != H || rows_cache[2] != strips) {
            rows_cache[3] = ssim_march_rows((const void *)ssim_march_bwd_kernel, C, H, strips);
            rows_cache[0] = C; rows_cache[1] = H; rows_cache[2] = strips;
        }
        const int seg_rows = rows_cache[3], segs = (H + seg_rows - 1) / seg_rows;
        const int vec_ok = (W % 4 == 0) && !(((uintptr_t)img1 | (uintptr_t)img2 | (uintptr_t)grad_img1) & 15);
        hipLaunchKernelGGL(ssim_march_bwd_kernel, dim3(ssim_grid(C * strips * segs, 8192)), dim3(kMarchW), 0, s, C, H, W, strips, segs, seg_rows,
                           img1, img2, Dm, Dq1, Dq12, g, g_l1, 1.0f / ((float)C * (float)H * (float)W), grad_img1, vec_ok);
        return check_launch(s, 0);
    }
    const int tx = (W + kSsimTW - 1) / kSsimTW, ty = (H + kSsimTH - 1) / kSsimTH;
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3(ssim_grid(C * tx * ty, 8192)), dim3(256), 0, s, C, H, W, tx, ty, img1, img2,
                       Dm, Dq1, Dq12, g, g_l1, 1.0f / ((float)C * (float)H * (float)W), grad_img1);
    return check_launch(s, 0);
}

extern "C" int d3ga_ssim_bwd(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, const float *Dm,
                             const float *Dq1, const float *Dq12, const float *g, float *grad_img1,
                             d3ga_stream_t stream) {
    return d3ga_ssim_l1_bwd(C, H, W, img1, img2, Dm, Dq1, Dq12, g, nullptr, grad_img1, stream);
}

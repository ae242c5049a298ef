// encoding.hip -- the two element-wise ops in front of ColorField:
//   view_dirs_*     v = (means3D - campos) / |means3D - campos|        (models/cage_net.py:233-235)
//   sh4_encoding_*  the 16 real spherical-harmonics polynomials of degree < 4 on x = 2 v - 1: the degree-4
//                   "SphericalHarmonics" direction encoding models/mlp.py:166-179 takes from tiny-cuda-nn.
// ATen ran these as ~45 element-wise kernels forward and ~110 backward (1.16 ms at 500k Gaussians); here one kernel
// each way, bound by the 64-byte encoding row: (12 + 64) B and (12 + 64 + 12) B per Gaussian.
#include "d3ga_internal.h"

namespace d3ga {

namespace sh4 {
constexpr float c0 = 0.28209479177387814f, c1 = 0.48860251190291987f;
constexpr float c2a = 1.0925484305920792f, c2b = 0.94617469575755997f, c2c = 0.31539156525251999f,
                c2d = 0.54627421529603959f;
constexpr float c3a = 0.59004358992664352f, c3b = 2.8906114426405538f, c3c = 0.45704579946446572f,
                c3d = 0.3731763325901154f, c3e = 1.4453057213202769f;
}  // namespace sh4

__global__ __launch_bounds__(kBlock) void view_dirs_fwd_kernel(int P, const float *__restrict__ means,
                                                               const float *__restrict__ campos, float *__restrict__ dirs) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const float dx = means[3 * i] - campos[0], dy = means[3 * i + 1] - campos[1], dz = means[3 * i + 2] - campos[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    dirs[3 * i] = dx * inv, dirs[3 * i + 1] = dy * inv, dirs[3 * i + 2] = dz * inv;
}

__global__ __launch_bounds__(kBlock) void view_dirs_bwd_kernel(int P, const float *__restrict__ means,
                                                               const float *__restrict__ campos, const float *__restrict__ g,
                                                               float *__restrict__ d_means) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const float dx = means[3 * i] - campos[0], dy = means[3 * i + 1] - campos[1], dz = means[3 * i + 2] - campos[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float vx = dx * inv, vy = dy * inv, vz = dz * inv;
    const float gx = g[3 * i], gy = g[3 * i + 1], gz = g[3 * i + 2];
    const float vg = vx * gx + vy * gy + vz * gz;          // d v / d m = (I - v v^T) / |m - c|
    d_means[3 * i] = (gx - vx * vg) * inv, d_means[3 * i + 1] = (gy - vy * vg) * inv, d_means[3 * i + 2] = (gz - vz * vg) * inv;
}

__global__ __launch_bounds__(kBlock) void sh4_encoding_fwd_kernel(int P, const float *__restrict__ dirs,
                                                                  float *__restrict__ enc) {
    using namespace sh4;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const float x = 2.0f * dirs[3 * i] - 1.0f, y = 2.0f * dirs[3 * i + 1] - 1.0f, z = 2.0f * dirs[3 * i + 2] - 1.0f;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    float4 *o = reinterpret_cast<float4 *>(enc + 16 * (size_t)i);
    o[0] = make_float4(c0, -c1 * y, c1 * z, -c1 * x);
    o[1] = make_float4(c2a * xy, -c2a * yz, c2b * zz - c2c, -c2a * xz);
    o[2] = make_float4(c2d * xx - c2d * yy, c3a * y * (-3.0f * xx + yy), c3b * xy * z, c3c * y * (1.0f - 5.0f * zz));
    o[3] = make_float4(c3d * z * (5.0f * zz - 3.0f), c3c * x * (1.0f - 5.0f * zz), c3e * z * (xx - yy),
                       c3a * x * (-xx + 3.0f * yy));
}

__global__ __launch_bounds__(kBlock) void sh4_encoding_bwd_kernel(int P, const float *__restrict__ dirs,
                                                                  const float *__restrict__ d_enc, float *__restrict__ d_dirs) {
    using namespace sh4;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const float x = 2.0f * dirs[3 * i] - 1.0f, y = 2.0f * dirs[3 * i + 1] - 1.0f, z = 2.0f * dirs[3 * i + 2] - 1.0f;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    const float4 *gp = reinterpret_cast<const float4 *>(d_enc + 16 * (size_t)i);
    const float4 g0 = gp[0], g1 = gp[1], g2 = gp[2], g3 = gp[3];
    // g0 = (e0..e3), g1 = (e4..e7), g2 = (e8..e11), g3 = (e12..e15)
    const float q = 1.0f - 5.0f * zz, r = 3.0f * (yy - xx);
    float gx = -c1 * g0.w + c2a * (y * g1.x - z * g1.w) + 2.0f * c2d * x * g2.x - 6.0f * c3a * xy * g2.y + c3b * yz * g2.z +
               c3c * q * g3.y + 2.0f * c3e * xz * g3.z + c3a * r * g3.w;
    float gy = -c1 * g0.y + c2a * (x * g1.x - z * g1.y) - 2.0f * c2d * y * g2.x + c3a * r * g2.y + c3b * xz * g2.z +
               c3c * q * g2.w - 2.0f * c3e * yz * g3.z + 6.0f * c3a * xy * g3.w;
    float gz = c1 * g0.z - c2a * (y * g1.y + x * g1.w) + 2.0f * c2b * z * g1.z + c3b * xy * g2.z - 10.0f * c3c * yz * g2.w +
               c3d * (15.0f * zz - 3.0f) * g3.x - 10.0f * c3c * xz * g3.y + c3e * (xx - yy) * g3.z;
    d_dirs[3 * i] = 2.0f * gx, d_dirs[3 * i + 1] = 2.0f * gy, d_dirs[3 * i + 2] = 2.0f * gz;     // x = 2 d - 1
}

// ColorField's per-row input columns in one pass (round 5): x (P, 16 + F) = [ sh4 encoding of the view direction | features ]
// -- what models/mlp.py:208-226 forms with an encoding call and a torch.cat (ATen: the 64-byte encoding rows written, then
// 2 x 320 B per Gaussian moved again by the cat; backward: the (P, 16 + F) input gradient split by two strided copies) --
// and its backward (input gradient -> d(direction), d(features)).  One thread per 16-byte group of x: consecutive threads
// touch consecutive groups of the row-major output, the features move as whole float4.
__global__ __launch_bounds__(kBlock) void color_rows_fwd_kernel(int P, int G /* 4 + F / 4 groups per row */, const float *__restrict__ dirs,
                                                                const float4 *__restrict__ feats, float4 *__restrict__ x) {
    using namespace sh4;
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (size_t)P * G) return;
    const int row = (int)(idx / (unsigned)G), g = (int)(idx - (size_t)row * G);
    if (g >= 4) { x[idx] = feats[(size_t)row * (G - 4) + (g - 4)]; return; }
    const float vx = 2.0f * dirs[3 * (size_t)row] - 1.0f, vy = 2.0f * dirs[3 * (size_t)row + 1] - 1.0f, vz = 2.0f * dirs[3 * (size_t)row + 2] - 1.0f;
    const float xx = vx * vx, yy = vy * vy, zz = vz * vz, xy = vx * vy, yz = vy * vz, xz = vx * vz;
    float4 o;                                                // the same expressions as sh4_encoding_fwd_kernel, group by group
    if (g == 0) o = make_float4(c0, -c1 * vy, c1 * vz, -c1 * vx);
    else if (g == 1) o = make_float4(c2a * xy, -c2a * yz, c2b * zz - c2c, -c2a * xz);
    else if (g == 2) o = make_float4(c2d * xx - c2d * yy, c3a * vy * (-3.0f * xx + yy), c3b * xy * vz, c3c * vy * (1.0f - 5.0f * zz));
    else o = make_float4(c3d * vz * (5.0f * zz - 3.0f), c3c * vx * (1.0f - 5.0f * zz), c3e * vz * (xx - yy), c3a * vx * (-xx + 3.0f * yy));
    x[idx] = o;
}

__global__ __launch_bounds__(kBlock) void color_rows_bwd_kernel(int P, int G, const float *__restrict__ dirs, const float4 *__restrict__ gx4,
                                                                float *__restrict__ d_dirs, float4 *__restrict__ d_feats) {
    using namespace sh4;
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (size_t)P * G) return;
    const int row = (int)(idx / (unsigned)G), g = (int)(idx - (size_t)row * G);
    if (g >= 4) { if (d_feats) d_feats[(size_t)row * (G - 4) + (g - 4)] = gx4[idx]; return; }
    if (g != 0 || !d_dirs) return;                           // the row's first thread takes the four encoding groups (64 bytes)
    const float x = 2.0f * dirs[3 * (size_t)row] - 1.0f, y = 2.0f * dirs[3 * (size_t)row + 1] - 1.0f, z = 2.0f * dirs[3 * (size_t)row + 2] - 1.0f;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    const float4 g0 = gx4[idx], g1 = gx4[idx + 1], g2 = gx4[idx + 2], g3 = gx4[idx + 3];
    const float q = 1.0f - 5.0f * zz, r = 3.0f * (yy - xx);          // (as sh4_encoding_bwd_kernel)
    const float gxv = -c1 * g0.w + c2a * (y * g1.x - z * g1.w) + 2.0f * c2d * x * g2.x - 6.0f * c3a * xy * g2.y + c3b * yz * g2.z +
                      c3c * q * g3.y + 2.0f * c3e * xz * g3.z + c3a * r * g3.w;
    const float gyv = -c1 * g0.y + c2a * (x * g1.x - z * g1.y) - 2.0f * c2d * y * g2.x + c3a * r * g2.y + c3b * xz * g2.z +
                      c3c * q * g2.w - 2.0f * c3e * yz * g3.z + 6.0f * c3a * xy * g3.w;
    const float gzv = c1 * g0.z - c2a * (y * g1.y + x * g1.w) + 2.0f * c2b * z * g1.z + c3b * xy * g2.z - 10.0f * c3c * yz * g2.w +
                      c3d * (15.0f * zz - 3.0f) * g3.x - 10.0f * c3c * xz * g3.y + c3e * (xx - yy) * g3.z;
    d_dirs[3 * (size_t)row] = 2.0f * gxv, d_dirs[3 * (size_t)row + 1] = 2.0f * gyv, d_dirs[3 * (size_t)row + 2] = 2.0f * gzv;
}

// Output heads of a field network: pred (P,N) -> up to four column groups, each written as its own contiguous (P, w_h)
// block of one planar buffer (block h starts at P * start_h floats) through its activation:
//   0: y = x        1: y = a * tanh(x)        2: y = sigmoid(x + a)
// (CanonicalField: tanh(pred[:, :4]) * scale_bary | pred[:, 4:8] | pred[:, 8:], models/mlp.py:107-110; ColorField:
// sigmoid(pred[:, :3]) | sigmoid(0.1 + pred[:, 3:4]), models/mlp.py:232.)  ATen ran each head as slice + activation (+ a
// copy when the consumer needs contiguous rows) and, backward, a zero-filled (P,N) buffer + strided copy + add per head.
struct HeadSpec { int n_heads; int start[4]; int width[4]; int act[4]; float a[4]; };

__global__ __launch_bounds__(kBlock) void field_heads_fwd_kernel(int P, int N, HeadSpec hs, const float *__restrict__ pred,
                                                                 float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (size_t)P * N) return;
    const int r = (int)(i / N), n = (int)(i - (size_t)r * N);
    int h = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) h = (q < hs.n_heads && n >= hs.start[q]) ? q : h;
    const float x = pred[i], a = hs.a[h];
    float y = x;
    if (hs.act[h] == 1) y = a * tanhf(x);
    else if (hs.act[h] == 2) y = 1.0f / (1.0f + expf(-(x + a)));
    out[(size_t)P * hs.start[h] + (size_t)r * hs.width[h] + (n - hs.start[h])] = y;
}

__global__ __launch_bounds__(kBlock) void field_heads_bwd_kernel(int P, int N, HeadSpec hs, const float *__restrict__ out,
                                                                 const float *__restrict__ g0, const float *__restrict__ g1,
                                                                 const float *__restrict__ g2, const float *__restrict__ g3,
                                                                 float *__restrict__ d_pred) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (size_t)P * N) return;
    const int r = (int)(i / N), n = (int)(i - (size_t)r * N);
    int h = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) h = (q < hs.n_heads && n >= hs.start[q]) ? q : h;
    const float *g = h == 0 ? g0 : (h == 1 ? g1 : (h == 2 ? g2 : g3));
    const size_t o = (size_t)r * hs.width[h] + (n - hs.start[h]);
    float d = 0.f;
    if (g) {                                               // a head nobody used has no gradient
        d = g[o];
        const float y = out[(size_t)P * hs.start[h] + o], a = hs.a[h];
        if (hs.act[h] == 1) d *= a - y * y / a;            // d/dx a tanh(x) = a (1 - tanh^2)
        else if (hs.act[h] == 2) d *= y * (1.0f - y);
    }
    d_pred[i] = d;
}

static inline int ew_grid(int P) { return (P + kBlock - 1) / kBlock; }

}  // namespace d3ga

using namespace d3ga;

static int make_heads(int N, int n_heads, const int32_t *width, const int32_t *act, const float *param, HeadSpec &hs) {
    if (n_heads < 1 || n_heads > 4 || !width || !act || !param) return D3GA_E_CONFIG;
    hs.n_heads = n_heads;
    int c = 0;
    for (int h = 0; h < 4; ++h) {
        hs.start[h] = c; hs.width[h] = h < n_heads ? width[h] : 0; hs.act[h] = h < n_heads ? act[h] : 0;
        hs.a[h] = h < n_heads ? param[h] : 0.f;
        if (h < n_heads && (width[h] < 1 || act[h] < 0 || act[h] > 2 || (act[h] == 1 && param[h] == 0.f))) return D3GA_E_CONFIG;
        c += hs.width[h];
    }
    return c == N ? D3GA_OK : D3GA_E_SIZE;
}

extern "C" int d3ga_field_heads_fwd(int32_t P, int32_t N, int32_t n_heads, const int32_t *width, const int32_t *act,
                                    const float *param, const float *pred, float *out, d3ga_stream_t stream) {
    if (P < 0 || N < 1) return D3GA_E_SIZE;
    HeadSpec hs;
    D3GA_TRY(make_heads(N, n_heads, width, act, param, hs));
    if (P == 0) return D3GA_OK;
    if (!pred || !out) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)P * N;
    hipLaunchKernelGGL(field_heads_fwd_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, P, N, hs, pred, out);
    return check_launch(s, 0);
}

extern "C" int d3ga_field_heads_bwd(int32_t P, int32_t N, int32_t n_heads, const int32_t *width, const int32_t *act,
                                    const float *param, const float *out, const float *g0, const float *g1, const float *g2,
                                    const float *g3, float *d_pred, d3ga_stream_t stream) {
    if (P < 0 || N < 1) return D3GA_E_SIZE;
    HeadSpec hs;
    D3GA_TRY(make_heads(N, n_heads, width, act, param, hs));
    if (P == 0) return D3GA_OK;
    if (!out || !d_pred) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)P * N;
    hipLaunchKernelGGL(field_heads_bwd_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, P, N, hs, out, g0,
                       g1, g2, g3, d_pred);
    return check_launch(s, 0);
}

extern "C" int d3ga_view_dirs_fwd(int32_t P, const float *means3D, const float *campos, float *dirs, d3ga_stream_t stream) {
    if (P <= 0) return D3GA_E_SIZE;
    if (!means3D || !campos || !dirs) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(view_dirs_fwd_kernel, dim3(ew_grid(P)), dim3(kBlock), 0, s, P, means3D, campos, dirs);
    return check_launch(s, 0);
}

extern "C" int d3ga_view_dirs_bwd(int32_t P, const float *means3D, const float *campos, const float *d_dirs, float *d_means3D,
                                  d3ga_stream_t stream) {
    if (P <= 0) return D3GA_E_SIZE;
    if (!means3D || !campos || !d_dirs || !d_means3D) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(view_dirs_bwd_kernel, dim3(ew_grid(P)), dim3(kBlock), 0, s, P, means3D, campos, d_dirs, d_means3D);
    return check_launch(s, 0);
}

extern "C" int d3ga_sh4_encoding_fwd(int32_t P, const float *dirs, float *enc, d3ga_stream_t stream) {
    if (P <= 0) return D3GA_E_SIZE;
    if (!dirs || !enc) return D3GA_E_NULL;
    if ((uintptr_t)enc & 15) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sh4_encoding_fwd_kernel, dim3(ew_grid(P)), dim3(kBlock), 0, s, P, dirs, enc);
    return check_launch(s, 0);
}

extern "C" int d3ga_color_rows_fwd(int32_t P, int32_t F, const float *dirs, const float *feats, float *x, d3ga_stream_t stream) {
    if (P <= 0 || F < 0 || (F & 3)) return D3GA_E_SIZE;
    if (!dirs || !x || (F > 0 && !feats)) return D3GA_E_NULL;
    if (((uintptr_t)x | (uintptr_t)feats) & 15) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const int G = 4 + F / 4;
    const size_t n = (size_t)P * G;
    hipLaunchKernelGGL(color_rows_fwd_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, P, G, dirs,
                       reinterpret_cast<const float4 *>(feats), reinterpret_cast<float4 *>(x));
    return check_launch(s, 0);
}

extern "C" int d3ga_color_rows_bwd(int32_t P, int32_t F, const float *dirs, const float *d_x, float *d_dirs, float *d_feats,
                                   d3ga_stream_t stream) {
    if (P <= 0 || F < 0 || (F & 3)) return D3GA_E_SIZE;
    if (!dirs || !d_x) return D3GA_E_NULL;
    if (((uintptr_t)d_x | (uintptr_t)d_feats) & 15) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const int G = 4 + F / 4;
    const size_t n = (size_t)P * G;
    hipLaunchKernelGGL(color_rows_bwd_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, P, G, dirs,
                       reinterpret_cast<const float4 *>(d_x), d_dirs, reinterpret_cast<float4 *>(d_feats));
    return check_launch(s, 0);
}

extern "C" int d3ga_sh4_encoding_bwd(int32_t P, const float *dirs, const float *d_enc, float *d_dirs, d3ga_stream_t stream) {
    if (P <= 0) return D3GA_E_SIZE;
    if (!dirs || !d_enc || !d_dirs) return D3GA_E_NULL;
    if ((uintptr_t)d_enc & 15) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sh4_encoding_bwd_kernel, dim3(ew_grid(P)), dim3(kBlock), 0, s, P, dirs, d_enc, d_dirs);
    return check_launch(s, 0);
}

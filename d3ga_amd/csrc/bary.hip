// bary.hip -- init-time point location in a tetrahedral cage (replaces tetra_sampler.compute_bary,
// lib/cage.py:325-327).  Exhaustive: each workgroup owns 256 points and streams all tets through LDS in chunks,
// so every tet record is read from HBM once per workgroup and 256 times from LDS (broadcast).  Barycentric
// weights follow submodules/tetrahedralize/include/tet/tetrahedron.h:77-101 (order a,b,c,d = corners 0..3).
#include "d3ga_internal.h"

namespace d3ga {

constexpr int kChunk = 256;

__device__ __forceinline__ float stp(V3 a, V3 b, V3 c) {   // a . (b x c)
    return a.x * (b.y * c.z - b.z * c.y) + a.y * (b.z * c.x - b.x * c.z) + a.z * (b.x * c.y - b.y * c.x);
}

__global__ __launch_bounds__(kBlock) void compute_bary_kernel(int P, int T, const float *__restrict__ points,
                                                              const float *__restrict__ corners,
                                                              float *__restrict__ barys, int32_t *__restrict__ tetra_id,
                                                              uint8_t *__restrict__ active) {
    __shared__ float s_c[kChunk * 12];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    const bool live = i < P;
    V3 p = v3(0.f, 0.f, 0.f);
    if (live) p = v3(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]);
    float best_min = -INFINITY;
    int best_t = 0;
    float best_w[4] = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < T; base += kChunk) {
        const int cnt = min(kChunk, T - base);
        __syncthreads();
        for (int k = tid; k < cnt * 12; k += kBlock) s_c[k] = corners[(size_t)base * 12 + k];
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < cnt; ++j) {
            const float *c = s_c + 12 * j;
            const V3 a = v3(c[0], c[1], c[2]), b = v3(c[3], c[4], c[5]), cc = v3(c[6], c[7], c[8]), d = v3(c[9], c[10], c[11]);
            const V3 vap = p - a, vbp = p - b, vab = b - a, vac = cc - a, vad = d - a, vbc = cc - b, vbd = d - b;
            const float v6 = 1.0f / stp(vab, vac, vad);
            const float wa = stp(vbp, vbd, vbc) * v6, wb = stp(vap, vac, vad) * v6, wc = stp(vap, vad, vab) * v6,
                        wd = stp(vap, vab, vac) * v6;
            const float mn = fminf(fminf(wa, wb), fminf(wc, wd));
            if (mn > best_min) {            // strict: ties keep the lowest tet index
                best_min = mn; best_t = base + j;
                best_w[0] = wa; best_w[1] = wb; best_w[2] = wc; best_w[3] = wd;
            }
        }
    }
    if (live) {
        reinterpret_cast<float4 *>(barys)[i] = make_float4(best_w[0], best_w[1], best_w[2], best_w[3]);
        tetra_id[i] = best_t;
        active[i] = best_min >= 0.f ? 1 : 0;
    }
}

}  // namespace d3ga

using namespace d3ga;

extern "C" int d3ga_compute_bary(int P, int T, const float *points, const float *tetra_corners, float *barys,
                                 int32_t *tetra_id, uint8_t *active, d3ga_stream_t stream) {
    if (P < 0 || T <= 0) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!points || !tetra_corners || !barys || !tetra_id || !active) return D3GA_E_NULL;
    hipLaunchKernelGGL(compute_bary_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, P, T,
                       points, tetra_corners, barys, tetra_id, active);
    return check_launch((hipStream_t)stream, 1);
}

// ---------------------------------------------------------------------------------------------------------
// Init-time scale seed: mean squared distance of every point to its 3 nearest neighbours.
// Replaces simple_knn._C.distCUDA2 (models/mesh_net.py:22,66) and the pytorch3d form used for cages,
// knn_points(p, p, K=4)[0][0, :, 1:].mean(-1) (models/cage_net.py:66).  Exhaustive, points streamed through LDS.
// ---------------------------------------------------------------------------------------------------------
namespace d3ga {

__global__ __launch_bounds__(kBlock) void knn3_mean_dist2_kernel(int P, const float *__restrict__ points,
                                                                 float *__restrict__ out) {
    __shared__ float s_p[kBlock * 3];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    const bool live = i < P;
    V3 p = v3(0.f, 0.f, 0.f);
    if (live) p = v3(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]);
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;       // three smallest squared distances, ascending
    for (int base = 0; base < P; base += kBlock) {
        const int cnt = min(kBlock, P - base);
        __syncthreads();
        for (int k = tid; k < cnt * 3; k += kBlock) s_p[k] = points[(size_t)base * 3 + k];
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < cnt; ++j) {
            if (base + j == i) continue;                      // the point itself
            const float dx = s_p[3 * j] - p.x, dy = s_p[3 * j + 1] - p.y, dz = s_p[3 * j + 2] - p.z;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < d2) {
                if (d < d1) {
                    d2 = d1;
                    if (d < d0) { d1 = d0; d0 = d; } else { d1 = d; }
                } else {
                    d2 = d;
                }
            }
        }
    }
    if (live) {
        // fewer than 3 other points: average what exists (0 for a single point)
        float sum = 0.f; int n = 0;
        if (d0 < INFINITY) { sum += d0; ++n; }
        if (d1 < INFINITY) { sum += d1; ++n; }
        if (d2 < INFINITY) { sum += d2; ++n; }
        out[i] = n ? sum / (float)n : 0.f;
    }
}

}  // namespace d3ga

extern "C" int d3ga_knn3_mean_dist2(int P, const float *points, float *out, d3ga_stream_t stream) {
    if (P < 0) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!points || !out) return D3GA_E_NULL;
    hipLaunchKernelGGL(d3ga::knn3_mean_dist2_kernel, dim3((P + d3ga::kBlock - 1) / d3ga::kBlock), dim3(d3ga::kBlock), 0,
                       (hipStream_t)stream, P, points, out);
    return d3ga::check_launch((hipStream_t)stream, 0);
}

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

// weights of p in the tet (a,b,c,d) and their minimum -- ONE definition for the exhaustive and the grid kernel, so that both
// produce bit-identical numbers for the same (point, tet)
struct TetW { float w[4]; float mn; };
__device__ __forceinline__ TetW tet_weights(V3 p, const float *c) {
    const V3 a = v3(c[0], c[1], c[2]), b = v3(c[3], c[4], c[5]), cc = v3(c[6], c[7], c[8]), d = v3(c[9], c[10], c[11]);
    const V3 vap = p - a, vbp = p - b, vab = b - a, vac = cc - a, vad = d - a, vbc = cc - b, vbd = d - b;
    const float v6 = 1.0f / stp(vab, vac, vad);
    TetW r;
    r.w[0] = stp(vbp, vbd, vbc) * v6; r.w[1] = stp(vap, vac, vad) * v6; r.w[2] = stp(vap, vad, vab) * v6;
    r.w[3] = stp(vap, vab, vac) * v6;
    r.mn = fminf(fminf(r.w[0], r.w[1]), fminf(r.w[2], r.w[3]));
    return r;
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
            const TetW tw = tet_weights(p, s_c + 12 * j);
            if (tw.mn > best_min) {         // strict: ties keep the lowest tet index
                best_min = tw.mn; best_t = base + j;
                best_w[0] = tw.w[0]; best_w[1] = tw.w[1]; best_w[2] = tw.w[2]; best_w[3] = tw.w[3];
            }
        }
    }
    if (live) {
        reinterpret_cast<float4 *>(barys)[i] = make_float4(best_w[0], best_w[1], best_w[2], best_w[3]);
        tetra_id[i] = best_t;
        active[i] = best_min >= 0.f ? 1 : 0;
    }
}

// Uniform-grid candidate pruning.  A point INSIDE a tet lies inside that tet's bounding box, and only a
// containing tet can have a non-negative minimum weight, so for points inside the cage the arg-max of the exhaustive search
// is found among the tets whose (slightly inflated) boxes overlap the point's grid cell: cell_start / cell_tets is that CSR
// list (built by d3ga_amd/tetra.py with device-side sorts).  Same tet_weights(), same tie rule (largest minimum, then the
// lowest tet index) -> the same barys / tetra_id, bit for bit.  Points for which no candidate contains them get
// min_weight < 0: the caller re-runs exactly those through the exhaustive kernel.
__global__ __launch_bounds__(kBlock) void compute_bary_grid_kernel(int P, const float *__restrict__ points,
                                                                   const float *__restrict__ corners,
                                                                   const int32_t *__restrict__ cell_start,
                                                                   const int32_t *__restrict__ cell_tets, float ox, float oy,
                                                                   float oz, float inv_h, int nx, int ny, int nz,
                                                                   float *__restrict__ barys, int32_t *__restrict__ tetra_id,
                                                                   float *__restrict__ min_weight) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const V3 p = v3(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]);
    float best_min = -INFINITY;
    int best_t = 0x7fffffff;
    float best_w[4] = {0.f, 0.f, 0.f, 0.f};
    const int cx = (int)floorf((p.x - ox) * inv_h), cy = (int)floorf((p.y - oy) * inv_h), cz = (int)floorf((p.z - oz) * inv_h);
    if (cx >= 0 && cy >= 0 && cz >= 0 && cx < nx && cy < ny && cz < nz) {
        const int cell = (cz * ny + cy) * nx + cx;
        for (int k = cell_start[cell]; k < cell_start[cell + 1]; ++k) {
            const int t = cell_tets[k];
            const TetW tw = tet_weights(p, corners + 12 * (size_t)t);
            if (tw.mn > best_min || (tw.mn == best_min && t < best_t)) {
                best_min = tw.mn; best_t = t;
                best_w[0] = tw.w[0]; best_w[1] = tw.w[1]; best_w[2] = tw.w[2]; best_w[3] = tw.w[3];
            }
        }
    }
    reinterpret_cast<float4 *>(barys)[i] = make_float4(best_w[0], best_w[1], best_w[2], best_w[3]);
    tetra_id[i] = best_t == 0x7fffffff ? 0 : best_t;
    min_weight[i] = best_min;
}

}  // namespace d3ga

using namespace d3ga;

extern "C" int d3ga_compute_bary_grid(int P, const float *points, const float *tetra_corners, const int32_t *cell_start,
                                      const int32_t *cell_tets, const float *origin_h, const int32_t *dims, float *barys,
                                      int32_t *tetra_id, float *min_weight, d3ga_stream_t stream) {
    if (P < 0) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!points || !tetra_corners || !cell_start || !cell_tets || !origin_h || !dims || !barys || !tetra_id || !min_weight)
        return D3GA_E_NULL;
    if (dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0 || !(origin_h[3] > 0.f)) return D3GA_E_SIZE;
    hipLaunchKernelGGL(compute_bary_grid_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, P, points,
                       tetra_corners, cell_start, cell_tets, origin_h[0], origin_h[1], origin_h[2], 1.0f / origin_h[3], dims[0],
                       dims[1], dims[2], barys, tetra_id, min_weight);
    return check_launch((hipStream_t)stream, 1);
}

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

// keeps the three smallest squared distances (ascending) -- shared by the exhaustive and the grid kernel
__device__ __forceinline__ void knn3_insert(V3 p, const float *q, float &d0, float &d1, float &d2) {
    const float dx = q[0] - p.x, dy = q[1] - p.y, dz = q[2] - p.z;
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
// knn_points(p, p, K=4)[0][0, :, 1:].mean(-1) (models/cage_net.py:66): a mean over THREE slots; with fewer than three other
// points pytorch3d pads the missing distances with 0, so the sum of what exists is divided by 3 as well
__device__ __forceinline__ float knn3_mean(float d0, float d1, float d2) {
    float sum = 0.f;
    if (d0 < INFINITY) sum += d0;
    if (d1 < INFINITY) sum += d1;
    if (d2 < INFINITY) sum += d2;
    return sum / 3.0f;
}

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
            knn3_insert(p, s_p + 3 * j, d0, d1, d2);
        }
    }
    if (live) out[i] = knn3_mean(d0, d1, d2);
}

// Uniform-grid version: the points are bucketed into cells of edge h (cell_start / cell_points: CSR built
// by d3ga_amd/tetra.py), a point scans the cells at Chebyshev ring r = 0, 1, 2, ... around its own and stops as soon as its
// third-smallest distance is within r h -- every unvisited point is at least that far away.  Same distance expression and the
// same three smallest values as the exhaustive kernel, hence the same mean.
__global__ __launch_bounds__(kBlock) void knn3_grid_kernel(int P, const float *__restrict__ points,
                                                           const int32_t *__restrict__ cell_start,
                                                           const int32_t *__restrict__ cell_points, float ox, float oy, float oz,
                                                           float h, int nx, int ny, int nz, float *__restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const V3 p = v3(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]);
    const float inv_h = 1.0f / h;
    const int cx = min(max((int)floorf((p.x - ox) * inv_h), 0), nx - 1), cy = min(max((int)floorf((p.y - oy) * inv_h), 0), ny - 1);
    const int cz = min(max((int)floorf((p.z - oz) * inv_h), 0), nz - 1);
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    const int rmax = max(nx, max(ny, nz));
    for (int r = 0; r <= rmax; ++r) {
        if (r > 0) {
            const float reach = (float)(r - 1) * h;                    // everything within ring r-1 has been seen
            if (d2 <= reach * reach) break;
        }
        auto visit = [&](int x, int y, int z) {
            const int cell = (z * ny + y) * nx + x;
            for (int k = cell_start[cell]; k < cell_start[cell + 1]; ++k) {
                const int j = cell_points[k];
                if (j != i) knn3_insert(p, points + 3 * (size_t)j, d0, d1, d2);
            }
        };
        for (int z = max(cz - r, 0); z <= min(cz + r, nz - 1); ++z)
            for (int y = max(cy - r, 0); y <= min(cy + r, ny - 1); ++y) {
                const bool face = (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
                if (face || r == 0) {                                  // a whole row of the shell
                    for (int x = max(cx - r, 0); x <= min(cx + r, nx - 1); ++x) visit(x, y, z);
                } else {                                               // interior row: only its two end cells are on the shell
                    if (cx - r >= 0) visit(cx - r, y, z);
                    if (cx + r < nx) visit(cx + r, y, z);
                }
            }
    }
    out[i] = knn3_mean(d0, d1, d2);
}

}  // namespace d3ga

extern "C" int d3ga_knn3_mean_dist2_grid(int P, const float *points, const int32_t *cell_start, const int32_t *cell_points,
                                         const float *origin_h, const int32_t *dims, float *out, d3ga_stream_t stream) {
    if (P < 0) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!points || !cell_start || !cell_points || !origin_h || !dims || !out) return D3GA_E_NULL;
    if (dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0 || !(origin_h[3] > 0.f)) return D3GA_E_SIZE;
    hipLaunchKernelGGL(d3ga::knn3_grid_kernel, dim3((P + d3ga::kBlock - 1) / d3ga::kBlock), dim3(d3ga::kBlock), 0, (hipStream_t)stream,
                       P, points, cell_start, cell_points, origin_h[0], origin_h[1], origin_h[2], origin_h[3], dims[0], dims[1],
                       dims[2], out);
    return d3ga::check_launch((hipStream_t)stream, 0);
}

extern "C" int d3ga_knn3_mean_dist2(int P, const float *points, float *out, d3ga_stream_t stream) {
    if (P < 0) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!points || !out) return D3GA_E_NULL;
    hipLaunchKernelGGL(d3ga::knn3_mean_dist2_kernel, dim3((P + d3ga::kBlock - 1) / d3ga::kBlock), dim3(d3ga::kBlock), 0,
                       (hipStream_t)stream, P, points, out);
    return d3ga::check_launch((hipStream_t)stream, 0);
}

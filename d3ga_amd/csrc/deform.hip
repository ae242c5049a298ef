// deform.hip -- cage-side kernels for gfx950: LBS of cage vertices (D0), fused tetrahedral-cage deformation of
// Gaussians forward/backward (D1-D5, A1) and the FEM regulariser (D6).   SURVEY.md sec. 8a.
//
// All three are HBM-streaming, one thread per element, no LDS, no MFMA (no dense contraction on this path).
// Cage vertices (V*12 B, a few hundred KB) and tets (T*16 B) are gathered through L2; the per-Gaussian streams
// (tet id, barycentrics, canonical gradient, scale, rotation -> mean, covariance) are read/written once with
// lane-contiguous addresses.  Gaussians are expected sorted by tet id (static during training), so the four
// corner gathers of neighbouring lanes hit the same cache lines and the backward's vertex-gradient atomics
// of a wavefront collapse onto few addresses.
#include "d3ga_internal.h"

namespace d3ga {

__device__ __forceinline__ V3 load3(const float *p, int i) { return v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }

// ---------------------------------------------------------------------------------------------------------
// D0: v' = Rh (sum_k w_k A[idx_k]) [v + delta; 1] + Th
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void lbs_fwd_kernel(int V, int K, const float *__restrict__ tmpl,
                                                         const float *__restrict__ delta,
                                                         const float *__restrict__ A,
                                                         const int32_t *__restrict__ idx, const float *__restrict__ w,
                                                         const float *__restrict__ Rh, const float *__restrict__ Th,
                                                         float *__restrict__ out) {
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= V) return;
    V3 p = load3(tmpl, v);
    if (delta) p = p + load3(delta, v);
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.f;
    for (int k = 0; k < K; ++k) {
        const float wk = w[(size_t)v * K + k];
        const float *a = A + 16 * (size_t)idx[(size_t)v * K + k];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] += wk * a[i];
    }
    V3 o = v3(T[0] * p.x + T[1] * p.y + T[2] * p.z + T[3], T[4] * p.x + T[5] * p.y + T[6] * p.z + T[7],
              T[8] * p.x + T[9] * p.y + T[10] * p.z + T[11]);
    if (Rh) o = v3(Rh[0] * o.x + Rh[1] * o.y + Rh[2] * o.z, Rh[3] * o.x + Rh[4] * o.y + Rh[5] * o.z,
                   Rh[6] * o.x + Rh[7] * o.y + Rh[8] * o.z);
    if (Th) o = o + v3(Th[0], Th[1], Th[2]);
    out[3 * v] = o.x; out[3 * v + 1] = o.y; out[3 * v + 2] = o.z;
}

__global__ __launch_bounds__(kBlock) void lbs_bwd_kernel(int V, int K, const float *__restrict__ A,
                                                         const int32_t *__restrict__ idx, const float *__restrict__ w,
                                                         const float *__restrict__ Rh, const float *__restrict__ g,
                                                         float *__restrict__ gdelta) {
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= V) return;
    V3 go = load3(g, v);
    if (Rh) go = v3(Rh[0] * go.x + Rh[3] * go.y + Rh[6] * go.z, Rh[1] * go.x + Rh[4] * go.y + Rh[7] * go.z,
                    Rh[2] * go.x + Rh[5] * go.y + Rh[8] * go.z);
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.f;
    for (int k = 0; k < K; ++k) {
        const float wk = w[(size_t)v * K + k];
        const float *a = A + 16 * (size_t)idx[(size_t)v * K + k];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] += wk * a[i];
    }
    gdelta[3 * v] = T[0] * go.x + T[4] * go.y + T[8] * go.z;
    gdelta[3 * v + 1] = T[1] * go.x + T[5] * go.y + T[9] * go.z;
    gdelta[3 * v + 2] = T[2] * go.x + T[6] * go.y + T[10] * go.z;
}

// ---------------------------------------------------------------------------------------------------------
// D1-D5
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_deform_in(int i, const float *__restrict__ tetpoints,
                                               const int32_t *__restrict__ tetras, const int32_t *__restrict__ tetra_id,
                                               const float *__restrict__ barys, const float *__restrict__ canon_grad,
                                               const float *__restrict__ scales, const float *__restrict__ rots,
                                               const float *__restrict__ delta_barys, int flags, DeformIn &in,
                                               int4 &vid) {
    const int t = tetra_id[i];
    vid = reinterpret_cast<const int4 *>(tetras)[t];
    in.x0 = load3(tetpoints, vid.x); in.x1 = load3(tetpoints, vid.y);
    in.x2 = load3(tetpoints, vid.z); in.x3 = load3(tetpoints, vid.w);
    const float4 b = reinterpret_cast<const float4 *>(barys)[i];
    in.bary[0] = b.x; in.bary[1] = b.y; in.bary[2] = b.z; in.bary[3] = b.w;
    if (delta_barys) {                       // canon_barys = barys + delta_bary (models/cage_net.py:213), fused
        const float4 d = reinterpret_cast<const float4 *>(delta_barys)[i];
        in.bary[0] += d.x; in.bary[1] += d.y; in.bary[2] += d.z; in.bary[3] += d.w;
    }
    // the canonical gradient is a property of the TETRAHEDRON: per Gaussian as the reference stores it (lib/cage.py:329,
    // 36 B x P streamed per pass), or -- D3GA_DEFORM_GRAD_PER_TET -- one (T,3,3) table read through tetra_id (L2-resident)
    const size_t gi = (flags & D3GA_DEFORM_GRAD_PER_TET) ? (size_t)t : (size_t)i;
#pragma unroll
    for (int k = 0; k < 9; ++k) in.G.m[k] = canon_grad[9 * gi + k];
    in.s[0] = scales[3 * (size_t)i]; in.s[1] = scales[3 * (size_t)i + 1]; in.s[2] = scales[3 * (size_t)i + 2];
    if (flags & D3GA_DEFORM_LOG_SCALES) {    // scales = exp(scaling) (models/cage_net.py:214), fused
        in.s[0] = expf(in.s[0]); in.s[1] = expf(in.s[1]); in.s[2] = expf(in.s[2]);
    }
    const float4 q = reinterpret_cast<const float4 *>(rots)[i];
    in.q[0] = q.x; in.q[1] = q.y; in.q[2] = q.z; in.q[3] = q.w;
}

__global__ __launch_bounds__(kBlock) void cage_deform_fwd_kernel(int P, const float *__restrict__ tetpoints,
                                                                 const int32_t *__restrict__ tetras,
                                                                 const int32_t *__restrict__ tetra_id,
                                                                 const float *__restrict__ barys,
                                                                 const float *__restrict__ canon_grad,
                                                                 const float *__restrict__ scales,
                                                                 const float *__restrict__ rots,
                                                                 const float *__restrict__ delta_barys, int flags,
                                                                 float *__restrict__ means, float *__restrict__ cov6) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    DeformIn in;
    int4 vid;
    load_deform_in(i, tetpoints, tetras, tetra_id, barys, canon_grad, scales, rots, delta_barys, flags, in, vid);
    float m[3], c[6];
    deform_fwd(in, m, c);
    means[3 * (size_t)i] = m[0]; means[3 * (size_t)i + 1] = m[1]; means[3 * (size_t)i + 2] = m[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) cov6[6 * (size_t)i + k] = c[k];
}

__device__ __forceinline__ void atomic_add3(float *base, int v, V3 g) {
    atomicAdd(base + 3 * (size_t)v, g.x);
    atomicAdd(base + 3 * (size_t)v + 1, g.y);
    atomicAdd(base + 3 * (size_t)v + 2, g.z);
}

// Block-level merge of the corner gradients (round 4; d3ga_cage_deform_bwd_merged).  The binding is static, so for every
// workgroup of 256 consecutive Gaussians the positions of its 1024 (Gaussian, corner) items in vertex-sorted order
// (item_pos), the segments of equal vertex (seg_ptr / seg_begin) and where each segment's sum goes (partial g = global
// segment index) are built ONCE (cage_deform.py: merge_plan).  The kernel drops its corner gradients into LDS at those
// positions, sums every segment in a fixed order (no atomics: bit-reproducible) and writes one partial per (workgroup,
// vertex) -- with spatially coherent numbering a few hundred per workgroup instead of 1024 corner records; the vertex
// gather then runs over the partials.
struct DeformMerge {
    const uint16_t *item_pos;      // (P,4)  position of item 4 i + c among its workgroup's items sorted by vertex
    const int32_t *seg_ptr;        // (workgroups + 1)  first global segment of each workgroup
    const uint16_t *seg_begin;     // (segments)  first position of the segment inside its workgroup
    float *partials;               // (segments,3)
};

__global__ __launch_bounds__(kBlock) void cage_deform_bwd_kernel(
    int P, const float *__restrict__ tetpoints, const int32_t *__restrict__ tetras, const int32_t *__restrict__ tetra_id,
    const float *__restrict__ barys, const float *__restrict__ canon_grad, const float *__restrict__ scales,
    const float *__restrict__ rots, const float *__restrict__ delta_barys, int flags,
    const float *__restrict__ g_means, const float *__restrict__ g_cov6,
    float *__restrict__ g_tetpoints, float *__restrict__ g_barys, float *__restrict__ g_scales,
    float *__restrict__ g_rots, float *__restrict__ corner_grads, DeformMerge mg) {
    __shared__ float s_val[3][4 * kBlock];                 // merged path: the workgroup's corner gradients in vertex order
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P && !mg.item_pos) return;
    DeformGrad o;
    int4 vid = make_int4(0, 0, 0, 0);
    if (i < P) {
        DeformIn in;
        load_deform_in(i, tetpoints, tetras, tetra_id, barys, canon_grad, scales, rots, delta_barys, flags, in, vid);
        float gm[3] = {g_means[3 * (size_t)i], g_means[3 * (size_t)i + 1], g_means[3 * (size_t)i + 2]};
        float gc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) gc[k] = g_cov6[6 * (size_t)i + k];
        deform_bwd(in, gm, gc, o);
        if (g_barys) reinterpret_cast<float4 *>(g_barys)[i] = make_float4(o.gbary[0], o.gbary[1], o.gbary[2], o.gbary[3]);
        if (g_scales) {
            if (flags & D3GA_DEFORM_LOG_SCALES) {    // d/d(log s) = s * d/ds
                o.gs[0] *= in.s[0]; o.gs[1] *= in.s[1]; o.gs[2] *= in.s[2];
            }
            g_scales[3 * (size_t)i] = o.gs[0]; g_scales[3 * (size_t)i + 1] = o.gs[1]; g_scales[3 * (size_t)i + 2] = o.gs[2];
        }
        if (g_rots) reinterpret_cast<float4 *>(g_rots)[i] = make_float4(o.gq[0], o.gq[1], o.gq[2], o.gq[3]);
    }
    if (mg.item_pos) {
        if (i < P) {
            const ushort4 ps = reinterpret_cast<const ushort4 *>(mg.item_pos)[i];
            s_val[0][ps.x] = o.gx0.x; s_val[1][ps.x] = o.gx0.y; s_val[2][ps.x] = o.gx0.z;
            s_val[0][ps.y] = o.gx1.x; s_val[1][ps.y] = o.gx1.y; s_val[2][ps.y] = o.gx1.z;
            s_val[0][ps.z] = o.gx2.x; s_val[1][ps.z] = o.gx2.y; s_val[2][ps.z] = o.gx2.z;
            s_val[0][ps.w] = o.gx3.x; s_val[1][ps.w] = o.gx3.y; s_val[2][ps.w] = o.gx3.z;
        }
        __syncthreads();
        const int g0 = mg.seg_ptr[blockIdx.x], g1 = mg.seg_ptr[blockIdx.x + 1];
        const int nitems = 4 * min(kBlock, P - (int)blockIdx.x * kBlock);
        for (int g = g0 + (int)threadIdx.x; g < g1; g += kBlock) {
            const int b = mg.seg_begin[g], e = g + 1 < g1 ? (int)mg.seg_begin[g + 1] : nitems;
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (int k = b; k < e; ++k) { sx += s_val[0][k]; sy += s_val[1][k]; sz += s_val[2][k]; }
            mg.partials[3 * (size_t)g] = sx; mg.partials[3 * (size_t)g + 1] = sy; mg.partials[3 * (size_t)g + 2] = sz;
        }
    } else if (corner_grads) {            // deterministic path: per-corner gradients, summed per vertex by vertex_gather_kernel
        float4 *c = reinterpret_cast<float4 *>(corner_grads + 12 * (size_t)i);
        c[0] = make_float4(o.gx0.x, o.gx0.y, o.gx0.z, o.gx1.x);
        c[1] = make_float4(o.gx1.y, o.gx1.z, o.gx2.x, o.gx2.y);
        c[2] = make_float4(o.gx2.z, o.gx3.x, o.gx3.y, o.gx3.z);
    } else if (g_tetpoints) {
        atomic_add3(g_tetpoints, vid.x, o.gx0); atomic_add3(g_tetpoints, vid.y, o.gx1);
        atomic_add3(g_tetpoints, vid.z, o.gx2); atomic_add3(g_tetpoints, vid.w, o.gx3);
    }
}

// 64-lane sum through DPP (see raster_composite.hip); result broadcast from lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_(float v) {
    v = dpp_add_<0xB1, 0xf>(v); v = dpp_add_<0x4E, 0xf>(v); v = dpp_add_<0x141, 0xf>(v); v = dpp_add_<0x140, 0xf>(v);
    v = dpp_add_<0x142, 0xa>(v); v = dpp_add_<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// One wavefront per cage vertex: sums the corner gradients of every (Gaussian, corner) incident to the vertex.
// vert_start (V+1) / vert_items (4P, item = 4*gaussian + corner) is the static CSR adjacency built once per cage.
// No atomics, bit-reproducible.
__global__ __launch_bounds__(kBlock) void vertex_gather_kernel(int V, const int32_t *__restrict__ vert_start,
                                                               const int32_t *__restrict__ vert_items,
                                                               const float *__restrict__ corner_grads,
                                                               float *__restrict__ g_tetpoints) {
    const int v = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (v >= V) return;                                       // wave-uniform
    const int lane = threadIdx.x & 63;
    const int b = vert_start[v], e = vert_start[v + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = b + lane; k < e; k += 64) {
        const float *c = corner_grads + 3 * (size_t)vert_items[k];
        sx += c[0]; sy += c[1]; sz += c[2];
    }
    sx = wave_sum_(sx); sy = wave_sum_(sy); sz = wave_sum_(sz);
    if (lane == 0) { g_tetpoints[3 * (size_t)v] = sx; g_tetpoints[3 * (size_t)v + 1] = sy; g_tetpoints[3 * (size_t)v + 2] = sz; }
}

// The same with ONE DPP ROW (16 lanes) per vertex, four vertices per wavefront: for short item lists -- the partials of the
// block-merged backward, two to four per vertex with coherent numbering -- a whole wavefront per vertex leaves 60 lanes idle
// and the launch is 4x the wavefronts (7.3 -> .. us at C3).
__device__ __forceinline__ float row_sum_(float v) {          // every lane of the row ends with the row's total
    v = dpp_add_<0xB1, 0xf>(v); v = dpp_add_<0x4E, 0xf>(v); v = dpp_add_<0x141, 0xf>(v); v = dpp_add_<0x140, 0xf>(v);
    return v;
}
__global__ __launch_bounds__(kBlock) void vertex_gather_row_kernel(int V, const int32_t *__restrict__ vert_start,
                                                                   const int32_t *__restrict__ vert_items,
                                                                   const float *__restrict__ values,
                                                                   float *__restrict__ g_tetpoints) {
    const int v = blockIdx.x * (kBlock / 16) + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    const bool live = v < V;
    const int b = live ? vert_start[v] : 0, e = live ? vert_start[v + 1] : 0;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = b + l16; k < e; k += 16) {
        const float *c = values + 3 * (size_t)vert_items[k];
        sx += c[0]; sy += c[1]; sz += c[2];
    }
    sx = row_sum_(sx); sy = row_sum_(sy); sz = row_sum_(sz);
    if (live && l16 == 0) { g_tetpoints[3 * (size_t)v] = sx; g_tetpoints[3 * (size_t)v + 1] = sy; g_tetpoints[3 * (size_t)v + 2] = sz; }
}
// The same gather with the LBS backward of D0 behind it (round 5: d3ga_cage_deform_bwd_merged_lbs -- the posed cage vertices came
// from d3ga_lbs_cage_fwd, so dL/d(delta) = (sum_k w_k A_k[:3,:3])^T Rh^T dL/d(tetpoint) can be formed while the vertex gradient sits
// in the row's registers: one launch instead of two for ~24 k vertices).  Lane k of the row takes joint k of the vertex.
// g_extra: a gradient that reaches the vertices by another route (the FEM regulariser), added before the skinning; g_tetpoints
// (optional): the vertex gradient itself.
__global__ __launch_bounds__(kBlock) void vertex_gather_lbs_row_kernel(int V, int K, const int32_t *__restrict__ vert_start,
                                                                       const int32_t *__restrict__ vert_items,
                                                                       const float *__restrict__ values,
                                                                       const float *__restrict__ g_extra,
                                                                       const float *__restrict__ A, const int32_t *__restrict__ idx,
                                                                       const float *__restrict__ w, const float *__restrict__ Rh,
                                                                       float *__restrict__ g_tetpoints, float *__restrict__ gdelta) {
    const int v = blockIdx.x * (kBlock / 16) + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    const bool live = v < V;
    const int b = live ? vert_start[v] : 0, e = live ? vert_start[v + 1] : 0;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = b + l16; k < e; k += 16) {
        const float *c = values + 3 * (size_t)vert_items[k];
        sx += c[0]; sy += c[1]; sz += c[2];
    }
    if (g_extra && live && l16 == 0) { sx += g_extra[3 * (size_t)v]; sy += g_extra[3 * (size_t)v + 1]; sz += g_extra[3 * (size_t)v + 2]; }
    sx = row_sum_(sx); sy = row_sum_(sy); sz = row_sum_(sz);
    if (g_tetpoints && live && l16 == 0) { g_tetpoints[3 * (size_t)v] = sx; g_tetpoints[3 * (size_t)v + 1] = sy; g_tetpoints[3 * (size_t)v + 2] = sz; }
    V3 go = v3(sx, sy, sz);
    if (Rh) go = v3(Rh[0] * go.x + Rh[3] * go.y + Rh[6] * go.z, Rh[1] * go.x + Rh[4] * go.y + Rh[7] * go.z,
                    Rh[2] * go.x + Rh[5] * go.y + Rh[8] * go.z);
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (live) {
        for (int k = l16; k < K; k += 16) {
            const float wk = w[(size_t)v * K + k];
            const float *a = A + 16 * (size_t)idx[(size_t)v * K + k];
            dx += wk * (a[0] * go.x + a[4] * go.y + a[8] * go.z);
            dy += wk * (a[1] * go.x + a[5] * go.y + a[9] * go.z);
            dz += wk * (a[2] * go.x + a[6] * go.y + a[10] * go.z);
        }
    }
    dx = row_sum_(dx); dy = row_sum_(dy); dz = row_sum_(dz);
    if (live && l16 == 0) { gdelta[3 * (size_t)v] = dx; gdelta[3 * (size_t)v + 1] = dy; gdelta[3 * (size_t)v + 2] = dz; }
}

// ---------------------------------------------------------------------------------------------------------
// D6
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void fem_fwd_kernel(int T, const float *__restrict__ tetpoints,
                                                         const int32_t *__restrict__ tetras,
                                                         const float *__restrict__ Dn_inv, float *__restrict__ energy) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= T) return;
    const int4 vid = reinterpret_cast<const int4 *>(tetras)[t];
    M3 D;
#pragma unroll
    for (int k = 0; k < 9; ++k) D.m[k] = Dn_inv[9 * (size_t)t + k];
    energy[t] = fem_energy_fwd(load3(tetpoints, vid.x), load3(tetpoints, vid.y), load3(tetpoints, vid.z),
                               load3(tetpoints, vid.w), D);
}
__global__ __launch_bounds__(kBlock) void fem_bwd_kernel(int T, const float *__restrict__ tetpoints,
                                                         const int32_t *__restrict__ tetras,
                                                         const float *__restrict__ Dn_inv, const float *__restrict__ g,
                                                         float *__restrict__ g_tetpoints) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= T) return;
    const int4 vid = reinterpret_cast<const int4 *>(tetras)[t];
    M3 D;
#pragma unroll
    for (int k = 0; k < 9; ++k) D.m[k] = Dn_inv[9 * (size_t)t + k];
    V3 gx[4];
    fem_energy_bwd(load3(tetpoints, vid.x), load3(tetpoints, vid.y), load3(tetpoints, vid.z), load3(tetpoints, vid.w),
                   D, g[t], gx);
    atomic_add3(g_tetpoints, vid.x, gx[0]); atomic_add3(g_tetpoints, vid.y, gx[1]);
    atomic_add3(g_tetpoints, vid.z, gx[2]); atomic_add3(g_tetpoints, vid.w, gx[3]);
}

}  // namespace d3ga

using namespace d3ga;

static inline int nblocks(int n) { return (n + kBlock - 1) / kBlock; }

extern "C" int d3ga_lbs_cage_fwd(int V, int K, const float *tmpl, const float *delta, const float *joint_mats,
                                 const int32_t *skin_idx, const float *skin_w, const float *Rh, const float *Th,
                                 float *out, d3ga_stream_t stream) {
    if (V < 0 || K <= 0) return D3GA_E_SIZE;
    if (V == 0) return D3GA_OK;
    if (!tmpl || !joint_mats || !skin_idx || !skin_w || !out) return D3GA_E_NULL;
    hipLaunchKernelGGL(lbs_fwd_kernel, dim3(nblocks(V)), dim3(kBlock), 0, (hipStream_t)stream, V, K, tmpl, delta,
                       joint_mats, skin_idx, skin_w, Rh, Th, out);
    return check_launch((hipStream_t)stream, 0);
}

extern "C" int d3ga_lbs_cage_bwd(int V, int K, const float *joint_mats, const int32_t *skin_idx, const float *skin_w,
                                 const float *Rh, const float *grad_out, float *grad_delta, d3ga_stream_t stream) {
    if (V < 0 || K <= 0) return D3GA_E_SIZE;
    if (V == 0) return D3GA_OK;
    if (!joint_mats || !skin_idx || !skin_w || !grad_out || !grad_delta) return D3GA_E_NULL;
    hipLaunchKernelGGL(lbs_bwd_kernel, dim3(nblocks(V)), dim3(kBlock), 0, (hipStream_t)stream, V, K, joint_mats,
                       skin_idx, skin_w, Rh, grad_out, grad_delta);
    return check_launch((hipStream_t)stream, 0);
}

extern "C" int d3ga_cage_deform_fwd_ex(int P, const float *tetpoints, const int32_t *tetras, const int32_t *tetra_id,
                                       const float *barys, const float *canon_grad, const float *scales,
                                       const float *rots, const float *delta_barys, int32_t flags, float *means3D,
                                       float *cov6, d3ga_stream_t stream) {
    if (flags & ~(D3GA_DEFORM_LOG_SCALES | D3GA_DEFORM_GRAD_PER_TET)) return D3GA_E_CONFIG;
    if (P < 0) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!tetpoints || !tetras || !tetra_id || !barys || !canon_grad || !scales || !rots || !means3D || !cov6)
        return D3GA_E_NULL;
    hipLaunchKernelGGL(cage_deform_fwd_kernel, dim3(nblocks(P)), dim3(kBlock), 0, (hipStream_t)stream, P, tetpoints,
                       tetras, tetra_id, barys, canon_grad, scales, rots, delta_barys, (int)flags, means3D, cov6);
    return check_launch((hipStream_t)stream, 0);
}

extern "C" int d3ga_cage_deform_fwd(int P, const float *tetpoints, const int32_t *tetras, const int32_t *tetra_id,
                                    const float *barys, const float *canon_grad, const float *scales,
                                    const float *rots, float *means3D, float *cov6, d3ga_stream_t stream) {
    return d3ga_cage_deform_fwd_ex(P, tetpoints, tetras, tetra_id, barys, canon_grad, scales, rots, nullptr, 0, means3D,
                                   cov6, stream);
}

extern "C" int d3ga_cage_deform_bwd_ex(int P, int V, const float *tetpoints, const int32_t *tetras,
                                       const int32_t *tetra_id, const float *barys, const float *canon_grad,
                                       const float *scales, const float *rots, const float *delta_barys, int32_t flags,
                                       const float *g_means, const float *g_cov6, float *g_tetpoints, float *g_barys,
                                       float *g_scales, float *g_rots, const int32_t *vert_start,
                                       const int32_t *vert_items, float *corner_grads, d3ga_stream_t stream) {
    if (P < 0 || V < 0) return D3GA_E_SIZE;
    if (flags & ~(D3GA_DEFORM_LOG_SCALES | D3GA_DEFORM_GRAD_PER_TET)) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const bool csr = g_tetpoints && vert_start && vert_items && corner_grads;
    if (g_tetpoints && V > 0 && (!csr || P == 0)) D3GA_HIP(zero_async(g_tetpoints, sizeof(float) * 3 * (size_t)V, s));
    if (P == 0) return D3GA_OK;
    if (!tetpoints || !tetras || !tetra_id || !barys || !canon_grad || !scales || !rots || !g_means || !g_cov6)
        return D3GA_E_NULL;
    hipLaunchKernelGGL(cage_deform_bwd_kernel, dim3(nblocks(P)), dim3(kBlock), 0, s, P, tetpoints, tetras, tetra_id,
                       barys, canon_grad, scales, rots, delta_barys, (int)flags, g_means, g_cov6, g_tetpoints, g_barys,
                       g_scales, g_rots, csr ? corner_grads : nullptr, DeformMerge{nullptr, nullptr, nullptr, nullptr});
    D3GA_TRY(check_launch(s, 0));
    if (csr && V > 0) {
        hipLaunchKernelGGL(vertex_gather_kernel, dim3((V + 3) / 4), dim3(kBlock), 0, s, V, vert_start, vert_items,
                           corner_grads, g_tetpoints);
        return check_launch(s, 0);
    }
    return D3GA_OK;
}

extern "C" int d3ga_cage_deform_bwd_merged(int P, int V, const float *tetpoints, const int32_t *tetras,
                                           const int32_t *tetra_id, const float *barys, const float *canon_grad,
                                           const float *scales, const float *rots, const float *delta_barys, int32_t flags,
                                           const float *g_means, const float *g_cov6, float *g_tetpoints, float *g_barys,
                                           float *g_scales, float *g_rots, const uint16_t *item_pos, const int32_t *seg_ptr,
                                           const uint16_t *seg_begin, int32_t n_segments, const int32_t *vert_start,
                                           const int32_t *vert_parts, float *partials, d3ga_stream_t stream) {
    if (P < 0 || V < 0 || n_segments < 0) return D3GA_E_SIZE;
    if (flags & ~(D3GA_DEFORM_LOG_SCALES | D3GA_DEFORM_GRAD_PER_TET)) return D3GA_E_CONFIG;
    if (!g_tetpoints) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) {
        if (V > 0) D3GA_HIP(zero_async(g_tetpoints, sizeof(float) * 3 * (size_t)V, s));
        return D3GA_OK;
    }
    if (!tetpoints || !tetras || !tetra_id || !barys || !canon_grad || !scales || !rots || !g_means || !g_cov6 || !item_pos ||
        !seg_ptr || !seg_begin || !vert_start || !vert_parts || !partials)
        return D3GA_E_NULL;
    if ((uintptr_t)item_pos & 7) return D3GA_E_CONFIG;                       // read 8 bytes per Gaussian
    hipLaunchKernelGGL(cage_deform_bwd_kernel, dim3(nblocks(P)), dim3(kBlock), 0, s, P, tetpoints, tetras, tetra_id,
                       barys, canon_grad, scales, rots, delta_barys, (int)flags, g_means, g_cov6, g_tetpoints, g_barys,
                       g_scales, g_rots, (float *)nullptr, DeformMerge{item_pos, seg_ptr, seg_begin, partials});
    D3GA_TRY(check_launch(s, 0));
    if (V > 0) {
        if ((int64_t)n_segments <= 24 * (int64_t)V)          // short lists (the usual case): a DPP row per vertex
            hipLaunchKernelGGL(vertex_gather_row_kernel, dim3((V + kBlock / 16 - 1) / (kBlock / 16)), dim3(kBlock), 0, s, V,
                               vert_start, vert_parts, (const float *)partials, g_tetpoints);
        else
            hipLaunchKernelGGL(vertex_gather_kernel, dim3((V + 3) / 4), dim3(kBlock), 0, s, V, vert_start, vert_parts,
                               (const float *)partials, g_tetpoints);
        return check_launch(s, 0);
    }
    return D3GA_OK;
}

extern "C" int d3ga_cage_deform_bwd_merged_lbs(int P, int V, const float *tetpoints, const int32_t *tetras,
                                               const int32_t *tetra_id, const float *barys, const float *canon_grad,
                                               const float *scales, const float *rots, const float *delta_barys, int32_t flags,
                                               const float *g_means, const float *g_cov6, float *g_tetpoints, float *g_barys,
                                               float *g_scales, float *g_rots, const uint16_t *item_pos, const int32_t *seg_ptr,
                                               const uint16_t *seg_begin, int32_t n_segments, const int32_t *vert_start,
                                               const int32_t *vert_parts, float *partials, int K, const float *joint_mats,
                                               const int32_t *skin_idx, const float *skin_w, const float *Rh,
                                               const float *g_tetpoints_extra, float *g_delta, d3ga_stream_t stream) {
    if (P < 0 || V < 0 || n_segments < 0 || K <= 0) return D3GA_E_SIZE;
    if (flags & ~(D3GA_DEFORM_LOG_SCALES | D3GA_DEFORM_GRAD_PER_TET)) return D3GA_E_CONFIG;
    if (!g_delta || !joint_mats || !skin_idx || !skin_w) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    if (V == 0) return D3GA_OK;
    if (P > 0) {
        if (!tetpoints || !tetras || !tetra_id || !barys || !canon_grad || !scales || !rots || !g_means || !g_cov6 || !item_pos ||
            !seg_ptr || !seg_begin || !vert_start || !vert_parts || !partials)
            return D3GA_E_NULL;
        if ((uintptr_t)item_pos & 7) return D3GA_E_CONFIG;                   // read 8 bytes per Gaussian
        // (g_tetpoints of the corner kernel is unused on the merged path: the partials carry the corner gradients)
        hipLaunchKernelGGL(cage_deform_bwd_kernel, dim3(nblocks(P)), dim3(kBlock), 0, s, P, tetpoints, tetras, tetra_id,
                           barys, canon_grad, scales, rots, delta_barys, (int)flags, g_means, g_cov6, g_tetpoints, g_barys,
                           g_scales, g_rots, (float *)nullptr, DeformMerge{item_pos, seg_ptr, seg_begin, partials});
        D3GA_TRY(check_launch(s, 0));
    } else if (!vert_start || !vert_parts) {
        return D3GA_E_NULL;                                                  // (P == 0: an all-empty CSR is still read)
    }
    hipLaunchKernelGGL(vertex_gather_lbs_row_kernel, dim3((V + kBlock / 16 - 1) / (kBlock / 16)), dim3(kBlock), 0, s, V, K,
                       vert_start, vert_parts, (const float *)partials, g_tetpoints_extra, joint_mats, skin_idx, skin_w, Rh,
                       g_tetpoints, g_delta);
    return check_launch(s, 0);
}

extern "C" int d3ga_cage_deform_bwd(int P, int V, const float *tetpoints, const int32_t *tetras,
                                    const int32_t *tetra_id, const float *barys, const float *canon_grad,
                                    const float *scales, const float *rots, const float *g_means, const float *g_cov6,
                                    float *g_tetpoints, float *g_barys, float *g_scales, float *g_rots,
                                    const int32_t *vert_start, const int32_t *vert_items, float *corner_grads,
                                    d3ga_stream_t stream) {
    return d3ga_cage_deform_bwd_ex(P, V, tetpoints, tetras, tetra_id, barys, canon_grad, scales, rots, nullptr, 0, g_means,
                                   g_cov6, g_tetpoints, g_barys, g_scales, g_rots, vert_start, vert_items, corner_grads,
                                   stream);
}

extern "C" int d3ga_fem_energy_fwd(int T, const float *tetpoints, const int32_t *tetras, const float *Dn_inv,
                                   float *energy, d3ga_stream_t stream) {
    if (T < 0) return D3GA_E_SIZE;
    if (T == 0) return D3GA_OK;
    if (!tetpoints || !tetras || !Dn_inv || !energy) return D3GA_E_NULL;
    hipLaunchKernelGGL(fem_fwd_kernel, dim3(nblocks(T)), dim3(kBlock), 0, (hipStream_t)stream, T, tetpoints, tetras,
                       Dn_inv, energy);
    return check_launch((hipStream_t)stream, 0);
}

extern "C" int d3ga_fem_energy_bwd(int T, int V, const float *tetpoints, const int32_t *tetras, const float *Dn_inv,
                                   const float *g_energy, float *g_tetpoints, d3ga_stream_t stream) {
    if (T < 0 || V < 0) return D3GA_E_SIZE;
    if (!g_tetpoints) return D3GA_E_NULL;
    if (V > 0) D3GA_HIP(zero_async(g_tetpoints, sizeof(float) * 3 * (size_t)V, (hipStream_t)stream));
    if (T == 0) return D3GA_OK;
    if (!tetpoints || !tetras || !Dn_inv || !g_energy) return D3GA_E_NULL;
    hipLaunchKernelGGL(fem_bwd_kernel, dim3(nblocks(T)), dim3(kBlock), 0, (hipStream_t)stream, T, tetpoints, tetras,
                       Dn_inv, g_energy, g_tetpoints);
    return check_launch((hipStream_t)stream, 0);
}

// raster_composite.hip -- alpha compositing forward / backward for gfx950.
//
// One 256-thread workgroup per 16x16 tile = four 64-lane wavefronts, each owning an 8x8 pixel quadrant (a compact
// footprint keeps the per-wavefront early-outs effective: a quadrant saturates, or falls outside a splat, sooner
// than a 16x4 strip does).  The tile's depth-ordered list is staged through LDS 256 entries at a time: every
// duplicate's 36 B record (xy, conic+opacity, rgb, 1/depth) is fetched from HBM/L2 once per tile with one
// gathered load per lane and then broadcast-read from LDS by all lanes (same address => conflict-free).
//
// forward : front-to-back; a wavefront stops as soon as all 64 pixels are saturated (T < 1e-4), the workgroup
//           stops loading when all four have.
// backward: back-to-front starting at the deepest contributor of the tile; the nine per-pixel partial derivatives
//           are reduced across the wavefront with interleaved DPP row operations (VALU only), the four wavefronts'
//           totals meet in a per-entry LDS accumulator (ds_add_f32), and after each batch ONE thread per staged
//           Gaussian issues the nine global float atomics -- 256x fewer than one per (pixel, Gaussian).  Wavefronts
//           in which no pixel is touched by a Gaussian skip it after a single ballot.
#include "d3ga_internal.h"

namespace d3ga {

// ---- wavefront (64 lanes) sum through DPP; result valid in lane 63, returned broadcast ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);   // row_mirror        -> every lane holds its 16-lane row sum
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3 -> row 3 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ void pixel_of_thread(int tid, int &lx, int &ly) {
    const int wave = tid >> 6, lane = tid & 63;
    lx = ((wave & 1) << 3) | (lane & 7);
    ly = ((wave >> 1) << 3) | (lane >> 3);
}

__global__ __launch_bounds__(kBlock) void composite_fwd_kernel(
    int W, int H, const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list, uint64_t dcap,
    const float2 *__restrict__ xy, const float4 *__restrict__ conic_o, const float4 *__restrict__ rgb_invd,
    const float *__restrict__ bg, float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
    float *__restrict__ out_color, float *__restrict__ out_invdepth) {
    __shared__ float2 s_xy[kBlock];
    __shared__ float4 s_co[kBlock];
    __shared__ float4 s_rgb[kBlock];
    const int tid = threadIdx.x;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    int lx, ly;
    pixel_of_thread(tid, lx, ly);
    const int px = blockIdx.x * kTile + lx, py = blockIdx.y * kTile + ly;
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const uint32_t begin = (uint32_t)min((uint64_t)tile_start[tile], dcap);
    const uint32_t end = (uint32_t)min((uint64_t)tile_start[tile + 1], dcap);

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t contributor = 0, last = 0;
    bool done = !inside;

    for (uint32_t base = begin; base < end; base += kBlock) {
        if (__syncthreads_and(done)) break;
        if (base + tid < end) {
            const uint32_t g = point_list[base + tid];
            s_xy[tid] = xy[g];
            s_co[tid] = conic_o[g];
            s_rgb[tid] = rgb_invd[g];
        }
        __syncthreads();
        const int cnt = (int)min((uint32_t)kBlock, end - base);
        for (int j = 0; j < cnt; ++j) {
            if (__all(done)) break;                       // wave-uniform
            if (!done) {
                ++contributor;
                const float2 c = s_xy[j];
                const float4 co = s_co[j];
                float alpha, G;
                if (splat_alpha(c.x - fx, c.y - fy, co.x, co.y, co.z, co.w, alpha, G)) {
                    const float test_T = T * (1.0f - alpha);
                    if (test_T < kTmin) {
                        done = true;
                    } else {
                        const float4 col = s_rgb[j];
                        const float w = alpha * T;
                        C0 += col.x * w; C1 += col.y * w; C2 += col.z * w; Dp += col.w * w;
                        T = test_T;
                        last = contributor;
                    }
                }
            }
        }
        // lanes that left the inner loop early (wave done) need no fix-up: they never look at contributor again
    }
    if (inside) {
        const size_t pid = (size_t)py * W + px;
        const size_t hw = (size_t)H * W;
        final_T[pid] = T;
        n_contrib[pid] = last;
        out_color[pid] = C0 + T * bg[0];
        out_color[hw + pid] = C1 + T * bg[1];
        out_color[2 * hw + pid] = C2 + T * bg[2];
        if (out_invdepth) out_invdepth[pid] = Dp;
    }
}

// Reduce NV values across the 64 lanes at once: the NV dependency chains are independent, so the scheduler
// interleaves them and the DPP wait states of one chain are filled by the others.  Totals end up in every lane of
// row 3 (lanes 48..63).
template <int NV>
__device__ __forceinline__ void wave_sum_multi(float (&v)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = dpp_add<0xB1, 0xf>(v[k]);
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = dpp_add<0x4E, 0xf>(v[k]);
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = dpp_add<0x141, 0xf>(v[k]);
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = dpp_add<0x140, 0xf>(v[k]);
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = dpp_add<0x142, 0xa>(v[k]);
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = dpp_add<0x143, 0xc>(v[k]);
}

constexpr int kNG = 9;   // partial derivatives per (pixel, Gaussian): mean2D x,y | conic a,b/2,c | opacity | r,g,b

__global__ __launch_bounds__(kBlock) void composite_bwd_kernel(
    int W, int H, const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list, uint64_t dcap,
    const float2 *__restrict__ xy, const float4 *__restrict__ conic_o, const float4 *__restrict__ rgb_invd,
    const float *__restrict__ bg, const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpix, float *__restrict__ acc) {
    __shared__ float2 s_xy[kBlock];
    __shared__ float4 s_co[kBlock];
    __shared__ float4 s_rgb[kBlock];
    __shared__ uint32_t s_id[kBlock];
    __shared__ float s_acc[kBlock * kNG];     // per staged entry: the tile's 9 partial sums (stride 9: conflict-free)
    __shared__ uint32_t s_maxlast;
    const int tid = threadIdx.x;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    int lx, ly;
    pixel_of_thread(tid, lx, ly);
    const int px = blockIdx.x * kTile + lx, py = blockIdx.y * kTile + ly;
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const uint32_t begin = (uint32_t)min((uint64_t)tile_start[tile], dcap);
    const uint32_t end = (uint32_t)min((uint64_t)tile_start[tile + 1], dcap);
    if (begin >= end) return;                              // uniform: empty tile

    const size_t pid = (size_t)py * W + px;
    const size_t hw = (size_t)H * W;
    const float T_final = inside ? final_T[pid] : 0.f;
    const uint32_t last = inside ? n_contrib[pid] : 0u;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (inside) { g0 = dL_dpix[pid]; g1 = dL_dpix[hw + pid]; g2 = dL_dpix[2 * hw + pid]; }
    const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;

    if (tid == 0) s_maxlast = 0;
    __syncthreads();
    atomicMax(&s_maxlast, last);
    __syncthreads();
    const uint32_t maxlast = s_maxlast;                    // deepest 1-based list position any pixel used

    float T = T_final;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;                    // colour accumulated behind the current splat
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    const int lane = tid & 63;
    const int slot = lane - 48;                            // lanes 48..56 publish value `slot` of the wave's totals

    // positions hi, hi-1, ... (1-based) in batches of 256, thread t stages position hi - t
    for (uint32_t hi = maxlast; hi > 0; hi = hi > kBlock ? hi - kBlock : 0) {
        __syncthreads();
        if ((uint32_t)tid < hi) {
            const uint32_t g = point_list[begin + (hi - 1 - tid)];
            s_id[tid] = g;
            s_xy[tid] = xy[g];
            s_co[tid] = conic_o[g];
            s_rgb[tid] = rgb_invd[g];
        }
#pragma unroll
        for (int k = 0; k < kNG; ++k) s_acc[tid * kNG + k] = 0.f;
        __syncthreads();
        const int cnt = (int)min((uint32_t)kBlock, hi);
        for (int j = 0; j < cnt; ++j) {
            const uint32_t pos = hi - j;                   // 1-based position of this entry
            const float2 c = s_xy[j];
            const float4 co = s_co[j];
            const float dx = c.x - fx, dy = c.y - fy;
            float alpha = 0.f, G = 0.f;
            const bool hit = inside && pos <= last && splat_alpha(dx, dy, co.x, co.y, co.z, co.w, alpha, G);
            if (!__any(hit)) continue;                     // wave-uniform skip
            float v[kNG];
#pragma unroll
            for (int k = 0; k < kNG; ++k) v[k] = 0.f;
            if (hit) {
                const float4 col = s_rgb[j];
                T = T / (1.0f - alpha);
                const float dch = alpha * T;
                a0 = last_alpha * lc0 + (1.f - last_alpha) * a0;
                a1 = last_alpha * lc1 + (1.f - last_alpha) * a1;
                a2 = last_alpha * lc2 + (1.f - last_alpha) * a2;
                lc0 = col.x; lc1 = col.y; lc2 = col.z;
                float dL_dalpha = ((col.x - a0) * g0 + (col.y - a1) * g1 + (col.z - a2) * g2) * T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                const float dL_dG = co.w * dL_dalpha;      // the 0.99 clamp passes the gradient through
                const float gdx = G * dx, gdy = G * dy;
                v[0] = dL_dG * (-gdx * co.x - gdy * co.y) * ddelx_dx;
                v[1] = dL_dG * (-gdy * co.z - gdx * co.y) * ddely_dy;
                v[2] = -0.5f * gdx * dx * dL_dG;
                v[3] = -0.5f * gdx * dy * dL_dG;            // half of dL/dB, doubled in the per-Gaussian backward
                v[4] = -0.5f * gdy * dy * dL_dG;
                v[5] = G * dL_dalpha;
                v[6] = dch * g0; v[7] = dch * g1; v[8] = dch * g2;
            }
            wave_sum_multi<kNG>(v);
            // lanes 48..56 each add one of the nine totals into the entry's LDS accumulator (ds_add_f32)
            float mine = v[0];
#pragma unroll
            for (int k = 1; k < kNG; ++k) mine = (slot == k) ? v[k] : mine;
            if (slot >= 0 && slot < kNG) atomicAdd(&s_acc[j * kNG + slot], mine);
        }
        __syncthreads();
        // flush: one thread per staged entry, nine global atomics per (tile, Gaussian) that any pixel touched
        if (tid < cnt) {
            float r[kNG];
            bool any = false;
#pragma unroll
            for (int k = 0; k < kNG; ++k) { r[k] = s_acc[tid * kNG + k]; any |= (r[k] != 0.f); }
            if (any) {
                float *o = acc + 12 * (size_t)s_id[tid];
                atomicAdd(o + 0, r[0]); atomicAdd(o + 1, r[1]);
                atomicAdd(o + 3, r[2]); atomicAdd(o + 4, r[3]); atomicAdd(o + 5, r[4]);
                atomicAdd(o + 6, r[5]);
                atomicAdd(o + 7, r[6]); atomicAdd(o + 8, r[7]); atomicAdd(o + 9, r[8]);
            }
        }
    }
}

// self-test kernel for wave_sum (tests/): out[w] = sum over the wave's 64 inputs
__global__ void wave_sum_selftest_kernel(const float *__restrict__ in, float *__restrict__ out) {
    const float s = wave_sum(in[blockIdx.x * blockDim.x + threadIdx.x]);
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
}

}  // namespace d3ga

using namespace d3ga;

extern "C" int d3ga_raster_composite_fwd(const d3ga_raster_params *prm, const float *bg, const void *geom,
                                         const void *binning, int64_t d_capacity, void *img, float *out_color,
                                         float *out_invdepth, d3ga_stream_t stream) {
    if (!prm || !bg || !geom || !binning || !img || !out_color) return D3GA_E_NULL;
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0 || d_capacity < 0) return D3GA_E_SIZE;
    hipStream_t s = (hipStream_t)stream;
    const int gx = tiles_x(prm->W), gy = tiles_y(prm->H);
    const BinBuf bin = carve_bin(const_cast<void *>(binning), (int64_t)gx * gy, d_capacity);
    const GeomBuf g = carve_geom(const_cast<void *>(geom), prm->P);
    const ImgBuf im = carve_img(img, prm->W, prm->H);
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(gx, gy), dim3(kBlock), 0, s, prm->W, prm->H, bin.tile_start,
                       bin.point_list, (uint64_t)d_capacity, g.xy, g.conic_o, g.rgb_invd, bg, im.final_T, im.n_contrib,
                       out_color, out_invdepth);
    return check_launch(s, prm->debug);
}

extern "C" int d3ga_raster_composite_bwd(const d3ga_raster_params *prm, const float *bg, const void *geom,
                                         const void *binning, int64_t d_capacity, const void *img,
                                         const float *dL_dpix, float *acc, d3ga_stream_t stream) {
    if (!prm) return D3GA_E_NULL;
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0 || d_capacity < 0) return D3GA_E_SIZE;
    if (prm->P == 0) return D3GA_OK;
    if (!bg || !geom || !binning || !img || !dL_dpix || !acc) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    const int gx = tiles_x(prm->W), gy = tiles_y(prm->H);
    const BinBuf bin = carve_bin(const_cast<void *>(binning), (int64_t)gx * gy, d_capacity);
    const GeomBuf g = carve_geom(const_cast<void *>(geom), prm->P);
    const ImgBuf im = carve_img(const_cast<void *>(img), prm->W, prm->H);
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(gx, gy), dim3(kBlock), 0, s, prm->W, prm->H, bin.tile_start,
                       bin.point_list, (uint64_t)d_capacity, g.xy, g.conic_o, g.rgb_invd, bg, im.final_T, im.n_contrib,
                       dL_dpix, acc);
    return check_launch(s, prm->debug);
}

// test hook (not part of the drop-in surface): n multiple of 64, in (n) -> out (n/64)
extern "C" int d3ga_selftest_wave_sum(int n, const float *in, float *out, d3ga_stream_t stream) {
    if (n <= 0 || (n % 256) != 0) return D3GA_E_SIZE;
    if (!in || !out) return D3GA_E_NULL;
    hipLaunchKernelGGL(wave_sum_selftest_kernel, dim3(n / 256), dim3(256), 0, (hipStream_t)stream, in, out);
    return check_launch((hipStream_t)stream, 1);
}

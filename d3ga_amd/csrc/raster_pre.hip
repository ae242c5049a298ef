// raster_pre.hip -- per-Gaussian stages of the tile rasterizer for gfx950.
//
//   preprocess_kernel      project, EWA 2D covariance, conic, radius, tile rectangle, SH colour, tile histogram
//   preprocess_bwd_kernel  conic/mean2D/colour/opacity gradients -> mean3D, cov3D | (scale, rotation), SH
//
// Both are HBM-streaming: one thread per Gaussian, every per-Gaussian array read/written once with
// lane-contiguous addresses (algorithmic bytes per Gaussian: fwd 88 + 12 M).  The camera
// matrices are wave-uniform and come through the scalar cache.  No LDS, no MFMA.
#include "d3ga_internal.h"
#include "raster_pre_body.h"

namespace d3ga {

__global__ __launch_bounds__(kBlock) void preprocess_kernel(
    d3ga_raster_params prm, const float *__restrict__ means3D, const float *__restrict__ shs,
    const float *__restrict__ colors_precomp, const float *__restrict__ opacities, const float *__restrict__ scales,
    const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp, const float *__restrict__ viewmatrix,
    const float *__restrict__ projmatrix, const float *__restrict__ campos, GeomBuf geom,
    uint32_t *__restrict__ tile_count, uint32_t *__restrict__ counters, int32_t *__restrict__ radii) {
    __shared__ int s_box[4];
    __shared__ uint32_t s_cnt[kWinTiles];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    bool visible = false;
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    if (i < prm.P) {
        const PreOut o = preprocess_one(prm, i, means3D, shs, colors_precomp, opacities, scales, rotations,
                                        cov3D_precomp, viewmatrix, projmatrix, campos);
        const Splat &sp = o.sp;
#pragma unroll
        for (int k = 0; k < 6; ++k) geom.cov3D[6 * (size_t)i + k] = o.c6[k];
        radii[i] = sp.radius;
        geom.depth[i] = sp.depth;
        geom.xy[i] = make_float2(sp.px, sp.py);
        // culled Gaussians keep an EMPTY rectangle: the scatter pass and the backward test visibility through it
        geom.rect[i] = sp.visible ? make_uint2((uint32_t)sp.rect[0] | ((uint32_t)sp.rect[1] << 16),
                                               (uint32_t)sp.rect[2] | ((uint32_t)sp.rect[3] << 16))
                                  : make_uint2(0u, 0u);
        geom.conic_o[i] = make_float4(sp.conic[0], sp.conic[1], sp.conic[2], o.opacity);
        geom.rgb_invd[i] = make_float4(o.rgb[0], o.rgb[1], o.rgb[2], sp.visible ? 1.0f / sp.depth : 0.f);
        geom.clamped[i] = o.clampmask;
        visible = sp.visible;
        r0 = sp.rect[0]; r1 = sp.rect[1]; r2 = sp.rect[2]; r3 = sp.rect[3];
    }
    // ---- tile histogram (counting-sort pass 1) through the block's LDS window ----
    const int gx = (prm.W + kTile - 1) / kTile;
    const TileWindow win = block_tile_window(s_box, visible, r0, r1, r2, r3);
    const int nvis = __syncthreads_count(visible);
    if (tid == 0 && nvis) atomicAdd(&counters[D3GA_CNT_VISIBLE], (uint32_t)nvis);
    const int area = win.area();
    if (area == 0) return;                                   // uniform
    if (win.fits()) {
        for (int k = tid; k < area; k += kBlock) s_cnt[k] = 0;
        __syncthreads();
        if (visible)
            for (int ty = r1; ty < r3; ++ty)
                for (int tx = r0; tx < r2; ++tx) atomicAdd(&s_cnt[(ty - win.y0) * win.w + (tx - win.x0)], 1u);
        __syncthreads();
        for (int k = tid; k < area; k += kBlock) {
            const uint32_t c = s_cnt[k];
            if (c) atomicAdd(&tile_count[(win.y0 + k / win.w) * gx + win.x0 + k % win.w], c);
        }
    } else if (visible) {                                    // huge footprints: straight to global memory
        for (int ty = r1; ty < r3; ++ty)
            for (int tx = r0; tx < r2; ++tx) atomicAdd(&tile_count[ty * gx + tx], 1u);
    }
}

__global__ __launch_bounds__(kBlock) void preprocess_bwd_kernel(
    d3ga_raster_params prm, const float *__restrict__ means3D, const float *__restrict__ shs,
    const float *__restrict__ scales, const float *__restrict__ rotations, const float *__restrict__ viewmatrix,
    const float *__restrict__ projmatrix, const float *__restrict__ campos, GeomBuf geom,
    const float *__restrict__ acc, float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D,
    float *__restrict__ dL_dopacity, float *__restrict__ dL_dsh, float *__restrict__ dL_dcolors,
    float *__restrict__ dL_dcov3D, float *__restrict__ dL_dscales, float *__restrict__ dL_drots) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= prm.P) return;
    const uint2 rc = geom.rect[i];
    const bool visible = ((rc.y & 0xffffu) > (rc.x & 0xffffu)) && ((rc.y >> 16) > (rc.x >> 16));
    float a[12], c6[6];
    if (visible) {
        const float4 *ap = reinterpret_cast<const float4 *>(acc + 12 * (size_t)i);
        const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
        a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
        a[8] = a2.x; a[9] = a2.y; a[10] = a2.z; a[11] = a2.w;
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = geom.cov3D[6 * (size_t)i + k];
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) a[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = 0.f;
    }
    preprocess_bwd_one(prm, i, visible, means3D, shs, scales, rotations, viewmatrix, projmatrix, campos, c6,
                       geom.clamped[i], a, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dsh, dL_dcolors, dL_dcov3D,
                       dL_dscales, dL_drots);
}

__global__ __launch_bounds__(kBlock) void mark_visible_kernel(int P, const float *__restrict__ means3D,
                                                              const float *__restrict__ viewmatrix,
                                                              uint8_t *__restrict__ visible) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    visible[i] = xform_point(viewmatrix, ld3(means3D, i)).z > kNear ? 1 : 0;
}

}  // namespace d3ga

using namespace d3ga;

static int validate(const d3ga_raster_params *prm) {
    if (!prm) return D3GA_E_NULL;
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0 || prm->M < 0 || prm->M > 16) return D3GA_E_SIZE;
    if (prm->sh_degree < 0 || prm->sh_degree > 3) return D3GA_E_CONFIG;
    if (prm->antialiasing) return D3GA_E_CONFIG;
    if (((prm->W + kTile - 1) / kTile) > 65535 || ((prm->H + kTile - 1) / kTile) > 65535) return D3GA_E_SIZE;
    return D3GA_OK;
}

extern "C" int d3ga_raster_preprocess(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                                      const float *colors_precomp, const float *opacities, const float *scales,
                                      const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                                      const float *projmatrix, const float *campos, void *geom, void *binning,
                                      int64_t d_capacity, int32_t *radii, d3ga_stream_t stream) {
    D3GA_TRY(validate(prm));
    if (!geom || !binning || !viewmatrix || !projmatrix || !campos) return D3GA_E_NULL;
    if (d_capacity < 0) return D3GA_E_SIZE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = (int64_t)tiles_x(prm->W) * tiles_y(prm->H);
    BinBuf bin = carve_bin(binning, tiles, d_capacity);
    // counters + tile_count are adjacent: one memset
    D3GA_HIP(hipMemsetAsync(bin.counters, 0, 256 + align256(4 * tiles), s));
    if (prm->P == 0) return D3GA_OK;          // empty scene: every per-Gaussian tensor is empty (NULL)
    if ((shs != nullptr) == (colors_precomp != nullptr)) return D3GA_E_CONFIG;
    const bool sr = scales != nullptr && rotations != nullptr;
    if (sr == (cov3D_precomp != nullptr)) return D3GA_E_CONFIG;
    if (shs && (prm->sh_degree + 1) * (prm->sh_degree + 1) > prm->M) return D3GA_E_CONFIG;
    if (!means3D || !opacities || !radii) return D3GA_E_NULL;
    GeomBuf g = carve_geom(geom, prm->P);
    hipLaunchKernelGGL(preprocess_kernel, dim3((prm->P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, *prm, means3D, shs,
                       colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, g,
                       bin.tile_count, bin.counters, radii);
    return check_launch(s, prm->debug);
}

extern "C" int d3ga_raster_preprocess_bwd(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                                          const float *scales, const float *rotations, const float *cov3D_precomp,
                                          const float *viewmatrix, const float *projmatrix, const float *campos,
                                          const void *geom, const float *acc, float *dL_dmeans3D, float *dL_dmeans2D,
                                          float *dL_dopacity, float *dL_dsh, float *dL_dcolors, float *dL_dcov3D,
                                          float *dL_dscales, float *dL_drots, d3ga_stream_t stream) {
    D3GA_TRY(validate(prm));
    (void)cov3D_precomp;
    if (prm->P == 0) return D3GA_OK;
    if (!means3D || !viewmatrix || !projmatrix || !campos || !geom || !acc || !dL_dmeans3D) return D3GA_E_NULL;
    if (dL_dsh && !shs) return D3GA_E_NULL;
    if ((dL_dscales != nullptr) != (dL_drots != nullptr)) return D3GA_E_CONFIG;
    if (dL_dscales && (!scales || !rotations)) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    GeomBuf g = carve_geom(const_cast<void *>(geom), prm->P);
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((prm->P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, *prm, means3D,
                       shs, scales, rotations, viewmatrix, projmatrix, campos, g, acc, dL_dmeans3D, dL_dmeans2D,
                       dL_dopacity, dL_dsh, dL_dcolors, dL_dcov3D, dL_dscales, dL_drots);
    return check_launch(s, prm->debug);
}

extern "C" int d3ga_raster_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, uint8_t *visible,
                                        d3ga_stream_t stream) {
    if (P < 0) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!means3D || !viewmatrix || !visible) return D3GA_E_NULL;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, P,
                       means3D, viewmatrix, visible);
    return check_launch((hipStream_t)stream, 0);
}

// raster_api.hip -- library identity, scratch sizing and the chained forward / backward entry points.
#include "d3ga_internal.h"

using namespace d3ga;

extern "C" int d3ga_version(void) { return D3GA_VERSION; }

extern "C" const char *d3ga_status_string(int status) {
    switch (status) {
        case D3GA_OK: return "ok";
        case D3GA_E_NULL: return "required pointer is NULL";
        case D3GA_E_SIZE: return "negative or inconsistent size";
        case D3GA_E_CONFIG: return "unsupported argument combination";
        case D3GA_E_CAPACITY: return "scratch buffer too small";
        default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}

extern "C" int d3ga_raster_scratch_bytes(int32_t P, int32_t W, int32_t H, int64_t d_capacity, int64_t sizes[3]) {
    if (!sizes) return D3GA_E_NULL;
    if (P < 0 || W <= 0 || H <= 0 || d_capacity < 0) return D3GA_E_SIZE;
    sizes[0] = geom_bytes(P > 0 ? P : 1);
    sizes[1] = bin_bytes((int64_t)tiles_x(W) * tiles_y(H), d_capacity > 0 ? d_capacity : 1);
    sizes[2] = img_bytes(W, H, (int64_t)tiles_x(W) * tiles_y(H), d_capacity > 0 ? d_capacity : 1);
    return D3GA_OK;
}

extern "C" int64_t d3ga_raster_img_bytes(int32_t W, int32_t H, int64_t d_capacity, int32_t forward_only) {
    if (W <= 0 || H <= 0 || d_capacity < 0) return D3GA_E_SIZE;
    const int64_t tiles = (int64_t)tiles_x(W) * tiles_y(H);
    return forward_only ? 2 * align256(4 * (int64_t)W * H) : img_bytes(W, H, tiles, d_capacity > 0 ? d_capacity : 1);
}

extern "C" int d3ga_raster_binning_layout(int32_t W, int32_t H, int64_t d_capacity, int64_t offsets[6]) {
    if (!offsets) return D3GA_E_NULL;
    if (W <= 0 || H <= 0 || d_capacity < 0) return D3GA_E_SIZE;
    char *base = (char *)nullptr + 256;   // any aligned non-null base; only differences are reported
    const BinBuf b = carve_bin(base, (int64_t)tiles_x(W) * tiles_y(H), d_capacity > 0 ? d_capacity : 1);
    offsets[0] = (char *)b.counters - base;
    offsets[1] = (char *)b.tile_count - base;
    offsets[2] = (char *)b.tile_start - base;
    offsets[3] = (char *)b.tile_cursor - base;
    offsets[4] = (char *)b.keys - base;
    offsets[5] = (char *)b.point_list - base;
    return D3GA_OK;
}

extern "C" int d3ga_raster_img_layout(int32_t W, int32_t H, int64_t offsets[2]) {
    if (!offsets) return D3GA_E_NULL;
    if (W <= 0 || H <= 0) return D3GA_E_SIZE;
    offsets[0] = 0;
    offsets[1] = align256(4 * (int64_t)W * H);
    return D3GA_OK;
}

extern "C" int d3ga_raster_img_layout_blocks(int32_t W, int32_t H, int64_t offsets[3]) {
    if (!offsets) return D3GA_E_NULL;
    if (W <= 0 || H <= 0) return D3GA_E_SIZE;
    const int64_t tiles = (int64_t)tiles_x(W) * tiles_y(H);
    char *base = (char *)nullptr + 256;
    const ImgBuf im = carve_img(base, W, H, tiles);
    offsets[0] = (char *)im.blk_count - base;
    offsets[1] = (char *)im.blk_total - base;
    offsets[2] = (char *)im.blk_list - base;
    return D3GA_OK;
}

extern "C" int d3ga_raster_forward(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                                   const float *colors_precomp, const float *opacities, const float *scales,
                                   const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                                   const float *projmatrix, const float *campos, const float *bg, void *geom,
                                   void *binning, void *img, int64_t d_capacity, float *out_color, int32_t *radii,
                                   float *out_invdepth, d3ga_stream_t stream) {
    D3GA_TRY(d3ga_raster_preprocess(prm, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                    viewmatrix, projmatrix, campos, geom, binning, d_capacity, radii, stream));
    int32_t lists = 0;
    D3GA_TRY(d3ga_raster_bin_sort_lists(prm, geom, binning, img, d_capacity, &lists, stream));
    d3ga_raster_params p2 = *prm;
    p2.block_lists = lists;
    return d3ga_raster_composite_fwd(&p2, bg, geom, binning, d_capacity, img, out_color, out_invdepth, stream);
}

extern "C" int d3ga_raster_backward(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                                    const float *scales, const float *rotations, const float *cov3D_precomp,
                                    const float *viewmatrix, const float *projmatrix, const float *campos,
                                    const float *bg, const void *geom, const void *binning, int64_t d_capacity,
                                    const void *img, const float *dL_dpix, float *acc, float *dL_dmeans3D,
                                    float *dL_dmeans2D, float *dL_dopacity, float *dL_dsh, float *dL_dcolors,
                                    float *dL_dcov3D, float *dL_dscales, float *dL_drots, d3ga_stream_t stream) {
    if (prm && prm->P > 0 && acc && !prm->acc_self_clearing)
        D3GA_HIP(zero_async(acc, sizeof(float) * D3GA_ACC_STRIDE * (size_t)prm->P, (hipStream_t)stream));
    D3GA_TRY(d3ga_raster_composite_bwd(prm, bg, geom, binning, d_capacity, img, dL_dpix, acc, stream));
    return d3ga_raster_preprocess_bwd(prm, means3D, shs, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                      campos, geom, acc, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dsh, dL_dcolors,
                                      dL_dcov3D, dL_dscales, dL_drots, stream);
}

extern "C" int d3ga_raster_backward_l1(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                                       const float *scales, const float *rotations, const float *cov3D_precomp,
                                       const float *viewmatrix, const float *projmatrix, const float *campos,
                                       const float *bg, const void *geom, const void *binning, int64_t d_capacity,
                                       const void *img, const float *image, const float *target, const void *target_cell,
                                       const float *g_loss, const float *dL_dpix, float *acc, float *dL_dmeans3D,
                                       float *dL_dmeans2D, float *dL_dopacity, float *dL_dsh, float *dL_dcolors,
                                       float *dL_dcov3D, float *dL_dscales, float *dL_drots, d3ga_stream_t stream) {
    if (prm && prm->P > 0 && acc && !prm->acc_self_clearing)
        D3GA_HIP(zero_async(acc, sizeof(float) * D3GA_ACC_STRIDE * (size_t)prm->P, (hipStream_t)stream));
    D3GA_TRY(d3ga_raster_composite_bwd_l1(prm, bg, geom, binning, d_capacity, img, image, target, target_cell, g_loss, dL_dpix,
                                          acc, stream));
    return d3ga_raster_preprocess_bwd(prm, means3D, shs, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                      campos, geom, acc, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dsh, dL_dcolors,
                                      dL_dcov3D, dL_dscales, dL_drots, stream);
}

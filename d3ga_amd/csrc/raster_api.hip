// raster_api.hip -- library identity, scratch sizing and the chained forward / backward entry points.
#include "d3ga_internal.h"

#include <atomic>

namespace d3ga {
// The debug knobs (include/d3ga.h): compiled defaults and the values in effect.  The table is the library's only mutable
// process state; nothing reads the environment.
static const int kKnobDefault[D3GA_KNOB_COUNT] = {
    /* COMPOSITE_VARIANT */ 32 | 128, /* MERGE_SLOTS */ 512, /* TILE_ASSIGN */ 2, /* BWD_SPLIT */ -1, /* SORT_MERGE */ -1,
    /* SSIM_IMPL */ 1, /* WGRAD_WS */ 1, /* CHAIN_ABL */ 0, /* CHAIN_GRID */ 0};
static std::atomic<int> g_knob[D3GA_KNOB_COUNT] = {{32 | 128}, {512}, {2}, {-1}, {-1}, {1}, {1}, {0}, {0}};
int debug_knob(int key) { return (key >= 0 && key < D3GA_KNOB_COUNT) ? g_knob[key].load(std::memory_order_relaxed) : 0; }
}  // namespace d3ga

using namespace d3ga;

extern "C" int d3ga_debug_set(int32_t key, int32_t value) {
    if (key < 0 || key >= D3GA_KNOB_COUNT) return D3GA_E_SIZE;
    g_knob[key].store(value == D3GA_KNOB_DEFAULT ? kKnobDefault[key] : value, std::memory_order_relaxed);
    return D3GA_OK;
}

// What this build of the library runs by default and right now: see include/d3ga.h.
extern "C" int d3ga_debug_defaults(int32_t *out, int32_t n) {
    if (!out) return D3GA_E_NULL;
    if (n < 2 + 2 * D3GA_KNOB_COUNT) return D3GA_E_SIZE;
#ifdef D3GA_SCAN_ABL
    out[0] = D3GA_SCAN_ABL;                     // != 0: a timing ablation -- results are WRONG by design (_lib.py refuses it)
#else
    out[0] = 0;
#endif
#ifdef D3GA_DIAG
    out[1] = 1;
#else
    out[1] = 0;
#endif
    for (int k = 0; k < D3GA_KNOB_COUNT; ++k) { out[2 + 2 * k] = kKnobDefault[k]; out[3 + 2 * k] = debug_knob(k); }
    return D3GA_OK;
}

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

extern "C" int d3ga_raster_scratch_bytes_views(int32_t P, int32_t W, int32_t H, int32_t n_views, int64_t d_capacity,
                                               int32_t forward_only, int64_t sizes[3]) {
    if (!sizes) return D3GA_E_NULL;
    if (P < 0 || W <= 0 || H <= 0 || d_capacity < 0 || n_views < 0) return D3GA_E_SIZE;
    const int64_t k = n_views > 1 ? n_views : 1;
    if ((int64_t)tiles_y(H) * k > 65535 || (int64_t)P * k >= (1ll << 31)) return D3GA_E_SIZE;     // tile rows are 16-bit fields of the rectangle records
    const int64_t tiles = (int64_t)tiles_x(W) * tiles_y(H) * k, cap = d_capacity > 0 ? d_capacity : 1;
    sizes[0] = geom_bytes(P > 0 ? P * k : 1);
    sizes[1] = bin_bytes(tiles, cap);
    sizes[2] = forward_only ? 2 * align256(4 * (int64_t)W * H * k) : img_bytes(W, H, tiles, cap, k);
    return D3GA_OK;
}

extern "C" int d3ga_raster_scratch_bytes(int32_t P, int32_t W, int32_t H, int64_t d_capacity, int64_t sizes[3]) {
    return d3ga_raster_scratch_bytes_views(P, W, H, 1, d_capacity, 0, sizes);
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

extern "C" int d3ga_raster_img_layout_blocks(int32_t W, int32_t H, int64_t offsets[2]) {
    if (!offsets) return D3GA_E_NULL;
    if (W <= 0 || H <= 0) return D3GA_E_SIZE;
    const int64_t tiles = (int64_t)tiles_x(W) * tiles_y(H);
    char *base = (char *)nullptr + 256;
    const ImgBuf im = carve_img(base, W, H, tiles);
    offsets[0] = (char *)im.blk_count - base;
    offsets[1] = (char *)im.blk_list - base;
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
    D3GA_TRY(d3ga_raster_bin_sort(prm, geom, binning, d_capacity, stream));
    return d3ga_raster_composite_fwd(prm, bg, geom, binning, d_capacity, img, out_color, out_invdepth, stream);
}

extern "C" int d3ga_raster_backward(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                                    const float *scales, const float *rotations, const float *cov3D_precomp,
                                    const float *viewmatrix, const float *projmatrix, const float *campos,
                                    const float *bg, const void *geom, const void *binning, int64_t d_capacity,
                                    const void *img, const float *dL_dpix, float *acc, float *dL_dmeans3D,
                                    float *dL_dmeans2D, float *dL_dopacity, float *dL_dsh, float *dL_dcolors,
                                    float *dL_dcov3D, float *dL_dscales, float *dL_drots, d3ga_stream_t stream) {
    if (prm && prm->P > 0 && acc && !prm->acc_self_clearing)
        D3GA_HIP(zero_async(acc, sizeof(float) * D3GA_ACC_STRIDE * (size_t)prm->P * (size_t)n_views_of(prm), (hipStream_t)stream));
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
        D3GA_HIP(zero_async(acc, sizeof(float) * D3GA_ACC_STRIDE * (size_t)prm->P * (size_t)n_views_of(prm), (hipStream_t)stream));
    D3GA_TRY(d3ga_raster_composite_bwd_l1(prm, bg, geom, binning, d_capacity, img, image, target, target_cell, g_loss, dL_dpix,
                                          acc, stream));
    return d3ga_raster_preprocess_bwd(prm, means3D, shs, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                      campos, geom, acc, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dsh, dL_dcolors,
                                      dL_dcov3D, dL_dscales, dL_drots, stream);
}

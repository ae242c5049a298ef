/*
 * d3ga.h -- C ABI of libd3ga_hip.so: the MI355X (gfx950) deform-and-rasterize hot path of D3GA.
 *
 * Every entry point is `extern "C"`, takes raw DEVICE pointers (unless marked host), element counts,
 * scalar settings and a HIP stream, and returns an int status:
 *      0  ok        <0  invalid argument (D3GA_E_*)        >0  a hipError_t from the runtime.
 * No entry point throws, allocates persistent device memory, or synchronises the stream (except
 * d3ga_compute_bary, an init-time call, and any call made with params.debug != 0, which synchronises and
 * checks after every kernel).  All scratch is caller-owned; the library keeps no mutable global state,
 * so it is re-entrant across devices and streams.  Tensors are dense, row-major, float32 unless stated;
 * index tensors are int32.
 *
 * Each entry point cites the interface of the reference (facebookresearch/D3GA, paths relative to its root)
 * that it replaces.  Rows refer to SURVEY.md sec. 8a.
 */
#ifndef D3GA_H
#define D3GA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the functions declared in this header are exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define D3GA_VERSION 110 /* round 6 (frozen for the round): view-batched rendering (d3ga_raster_params::n_views, d3ga_raster_scratch_bytes_views), d3ga_debug_set / D3GA_KNOB_* replace every environment knob, the opt-in list forwards of round 5 are gone (d3ga_raster_bin_sort_lists, d3ga_raster_params::block_lists), only the functions declared here are exported */

#define D3GA_OK 0
#define D3GA_E_NULL (-1)     /* required pointer is NULL */
#define D3GA_E_SIZE (-2)     /* negative / inconsistent size */
#define D3GA_E_CONFIG (-3)   /* unsupported combination (e.g. both shs and colors_precomp) */
#define D3GA_E_CAPACITY (-4) /* scratch buffer too small */

typedef void *d3ga_stream_t; /* hipStream_t */

int d3ga_version(void);
const char *d3ga_status_string(int status);
/* Debug knobs (host only, no device work).  The library reads NO environment variable; the one piece of mutable process state it
 * keeps is this table of integers, for tests and A/B timing runs (d3ga_amd/_lib.py applies D3GA_KNOBS="name=value,..." at load).
 * Product code never calls d3ga_debug_set; with every knob at its default the entry points are re-entrant across devices and streams.
 *   d3ga_debug_set(key, value): set knob `key` (D3GA_KNOB_*); value == D3GA_KNOB_DEFAULT restores the compiled default.
 *   d3ga_debug_defaults(out, n): out[0] = D3GA_SCAN_ABL the library was compiled with (0 = product; anything else is a TIMING
 *   ABLATION whose results are wrong by design -- the Python layer refuses such a library unless D3GA_ALLOW_ABLATION=1), out[1] = 1
 *   for a diagnostic (counter) build, then for knob k < D3GA_KNOB_COUNT: out[2 + 2k] = compiled default, out[3 + 2k] = value in
 *   effect; n = capacity of out in int32 (>= 2 + 2 * D3GA_KNOB_COUNT, else D3GA_E_SIZE). */
#define D3GA_KNOB_COMPOSITE_VARIANT 0 /* bit 5 (32) work-ordered dispatch of the compositing kernels, bit 7 (128) exact block culling; default 160 */
#define D3GA_KNOB_MERGE_SLOTS 1       /* slots of the compositing backward's per-tile merge cache: 256 | 512 (default) */
#define D3GA_KNOB_TILE_ASSIGN 2       /* block -> wavefront assignment of the compositing backward: 0 quadrants, 1 interleaved, 2 by list length (default); +8: no early exit */
#define D3GA_KNOB_BWD_SPLIT 3         /* compositing backward: -1 (default) the D3GA_CNT_HEAVY heaviest tiles get two workgroups, 0 none, n > 0 the n heaviest */
#define D3GA_KNOB_SORT_MERGE 4        /* -1 (default) by size: the 2049..4096 list class rides in the 8192-key sort launch when few such lists are expected; 0 never; 1 always */
#define D3GA_KNOB_SSIM_IMPL 5         /* 1 (default) the marching register-window SSIM kernels, 0 the LDS-tiled ones of round 3 */
#define D3GA_KNOB_WGRAD_WS 6          /* 1 (default) the wavefront-specialised weight-gradient kernel for two wide operands and for a narrow dPre, 0 the barrier-phased one, 2 wavefront-specialised for every shape */
#define D3GA_KNOB_CHAIN_ABL 7         /* 0 (default); != 0: TIMING ablations of the fused field-network kernel (wrong results) */
#define D3GA_KNOB_CHAIN_GRID 8        /* 0 (default) = 2048 / wavefronts per workgroup; > 0: workgroups of the fused field-network kernel */
#define D3GA_KNOB_COUNT 9
#define D3GA_KNOB_DEFAULT (-2147483647 - 1)
int d3ga_debug_set(int32_t key, int32_t value);
int d3ga_debug_defaults(int32_t *out, int32_t n);

/* ---------------------------------------------------------------------------------------------------------
 * D0  Linear blend skinning of cage vertices (K-sparse weights).
 * Replaces: lib/smplman.py:155-171 Smplman.deform  (T = W.A ; v' = T[v+delta;1] ; v'.Rh^T + Th) and
 *           lbsmodel/body_model.py:208-234 LinearBlendSkinning.skinning (8-sparse form).
 *   tmpl (V,3), delta (V,3)|NULL, joint_mats (J,4,4), skin_idx (V,K) int32, skin_w (V,K),
 *   Rh (3,3)|NULL, Th (3)|NULL  ->  out (V,3).
 * bwd: grad_out (V,3) -> grad_delta (V,3)  (= gradient w.r.t. the template offset / deformation_field output).
 * ------------------------------------------------------------------------------------------------------- */
int d3ga_lbs_cage_fwd(int V, int K, const float *tmpl, const float *delta, const float *joint_mats,
                      const int32_t *skin_idx, const float *skin_w, const float *Rh, const float *Th, float *out,
                      d3ga_stream_t stream);
int d3ga_lbs_cage_bwd(int V, int K, const float *joint_mats, const int32_t *skin_idx, const float *skin_w,
                      const float *Rh, const float *grad_out, float *grad_delta, d3ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * D1-D5  Fused tetrahedral-cage deformation.
 * Replaces: models/cage_net.py:218-230 (tetpoints[tetra_faces], compute_def_grad, J S J^T, strip_symmetric,
 *           einsum bary means) with lib/cage.py:339-342 and utils/general_utils.py:24-35,58-90.
 *   tetpoints (V,3) posed cage vertices; tetras (T,4) int32; tetra_id (P) int32; barys (P,4) (= barys+delta_bary);
 *   canon_grad (P,3,3) = inv(Dm) (lib/cage.py:329); scales (P,3) activated; rots (P,4) wxyz (normalised inside)
 *   -> means3D (P,3), cov6 (P,6) in order xx,xy,xz,yy,yz,zz.
 * bwd: g_means (P,3), g_cov6 (P,6) -> g_tetpoints (V,3), g_barys (P,4), g_scales (P,3), g_rots (P,4).  Any of the
 *      four outputs may be NULL (skipped).  The vertex gradient is a scatter-add over 4 corners x P Gaussians:
 *      - with the static adjacency of the cage given -- vert_start (V+1) int32, vert_items (4P) int32 listing, per
 *        vertex, the items 4*gaussian+corner incident to it, plus corner_grads (P,4,3) float scratch -- it is
 *        computed WITHOUT atomics (one wavefront per vertex gathers and sums; deterministic);
 *      - with those three NULL it falls back to float atomics into g_tetpoints (zeroed by the call).
 * ------------------------------------------------------------------------------------------------------- */
#define D3GA_DEFORM_LOG_SCALES 1 /* `scales` holds log-scales: exp() applied inside, g_scales is d/d(log-scale) */
#define D3GA_DEFORM_GRAD_PER_TET 2 /* `canon_grad` is (T,3,3), one matrix per tetrahedron, read through tetra_id (the reference
                                    * stores the same matrices gathered per Gaussian, lib/cage.py:329: 36 B x P per pass) */
int d3ga_cage_deform_fwd(int P, const float *tetpoints, const int32_t *tetras, const int32_t *tetra_id,
                         const float *barys, const float *canon_grad, const float *scales, const float *rots,
                         float *means3D, float *cov6, d3ga_stream_t stream);
int d3ga_cage_deform_bwd(int P, int V, const float *tetpoints, const int32_t *tetras, const int32_t *tetra_id,
                         const float *barys, const float *canon_grad, const float *scales, const float *rots,
                         const float *g_means, const float *g_cov6, float *g_tetpoints, float *g_barys,
                         float *g_scales, float *g_rots, const int32_t *vert_start, const int32_t *vert_items,
                         float *corner_grads, d3ga_stream_t stream);
/* Same ops with the two activations of models/cage_net.py:213-214 fused: delta_barys (P,4) or NULL is added to barys
 * (g_barys is then the gradient of both), and flags & D3GA_DEFORM_LOG_SCALES applies scales = exp(.) on load. */
int d3ga_cage_deform_fwd_ex(int P, const float *tetpoints, const int32_t *tetras, const int32_t *tetra_id,
                            const float *barys, const float *canon_grad, const float *scales, const float *rots,
                            const float *delta_barys, int32_t flags, float *means3D, float *cov6, d3ga_stream_t stream);
int d3ga_cage_deform_bwd_ex(int P, int V, const float *tetpoints, const int32_t *tetras, const int32_t *tetra_id,
                            const float *barys, const float *canon_grad, const float *scales, const float *rots,
                            const float *delta_barys, int32_t flags, const float *g_means, const float *g_cov6,
                            float *g_tetpoints, float *g_barys, float *g_scales, float *g_rots,
                            const int32_t *vert_start, const int32_t *vert_items, float *corner_grads,
                            d3ga_stream_t stream);

/* The same backward with the corner gradients merged per WORKGROUP before they leave the CU (round 4).  The binding
 * (tetras, tetra_id) is static, so a plan is built once (d3ga_amd/cage_deform.py: merge_plan): for every block of 256
 * consecutive Gaussians, item_pos (P,4 u16; 8-byte aligned) = position of item 4 i + corner among the block's items sorted by
 * cage vertex, seg_ptr (blocks + 1) / seg_begin (segments, u16) = the runs of equal vertex, and a second-level CSR
 * vert_start (V + 1) / vert_parts (segments) from vertices to segments.  The kernel sums every run in a fixed order in LDS
 * and writes one partial per run (partials: (segments,3) scratch); the vertex gather then adds a vertex's partials.
 * No atomics, bit-reproducible; with spatially coherent numbering (tetra.spatial_order) a block has a few hundred runs
 * instead of 1024 items.  g_tetpoints is required. */
int d3ga_cage_deform_bwd_merged(int P, int V, const float *tetpoints, const int32_t *tetras, const int32_t *tetra_id,
                                const float *barys, const float *canon_grad, const float *scales, const float *rots,
                                const float *delta_barys, int32_t flags, const float *g_means, const float *g_cov6,
                                float *g_tetpoints, float *g_barys, float *g_scales, float *g_rots,
                                const uint16_t *item_pos, const int32_t *seg_ptr, const uint16_t *seg_begin,
                                int32_t n_segments, const int32_t *vert_start, const int32_t *vert_parts, float *partials,
                                d3ga_stream_t stream);

/* ... with the LBS backward of D0 in the vertex-gather launch (round 5): when the posed cage vertices came from d3ga_lbs_cage_fwd
 * (lib/smplman.py:155-171 feeding models/cage_net.py:218), dL/d(delta) = (sum_k w_k A_k[:3,:3])^T Rh^T dL/d(tetpoint) is formed while
 * the gathered vertex gradient sits in registers: one launch instead of d3ga_cage_deform_bwd_merged's gather + d3ga_lbs_cage_bwd.
 * K, joint_mats, skin_idx, skin_w, Rh: as d3ga_lbs_cage_bwd.  g_tetpoints_extra (V,3) | NULL: a gradient that reaches the posed
 * vertices by another route (the FEM regulariser), added before the skinning.  g_tetpoints (V,3) | NULL: also write the vertex
 * gradient itself.  g_delta (V,3): required. */
int d3ga_cage_deform_bwd_merged_lbs(int P, int V, const float *tetpoints, const int32_t *tetras, const int32_t *tetra_id,
                                    const float *barys, const float *canon_grad, const float *scales, const float *rots,
                                    const float *delta_barys, int32_t flags, const float *g_means, const float *g_cov6,
                                    float *g_tetpoints, float *g_barys, float *g_scales, float *g_rots,
                                    const uint16_t *item_pos, const int32_t *seg_ptr, const uint16_t *seg_begin,
                                    int32_t n_segments, const int32_t *vert_start, const int32_t *vert_parts, float *partials,
                                    int K, const float *joint_mats, const int32_t *skin_idx, const float *skin_w, const float *Rh,
                                    const float *g_tetpoints_extra, float *g_delta, d3ga_stream_t stream);

/* D6  FEM regulariser (lib/cage.py:349-361): per-tet energy 0.5(det F-1)^2 + 0.5(|F|_F^2-3), F = Ds Dn^-1.
 *   fwd: energy (T).  bwd: g_energy (T) -> g_tetpoints (V,3) [zeroed by the call]. */
int d3ga_fem_energy_fwd(int T, const float *tetpoints, const int32_t *tetras, const float *Dn_inv, float *energy,
                        d3ga_stream_t stream);
int d3ga_fem_energy_bwd(int T, int V, const float *tetpoints, const int32_t *tetras, const float *Dn_inv,
                        const float *g_energy, float *g_tetpoints, d3ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * R1-R6  Tile rasterizer.  Replaces the un-vendored package `diff_gaussian_rasterization`
 * (graphdeco-inria, branch dr_aa; /root/reference/.gitmodules:9-12) as called from renderer.py:79-141:
 * _C.rasterize_gaussians / _C.rasterize_gaussians_backward / _C.mark_visible.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct d3ga_raster_params {
    int32_t P;           /* number of Gaussians */
    int32_t M;           /* SH coefficients per Gaussian in `shs` (stride), 0 if colors_precomp */
    int32_t sh_degree;   /* active degree 0..3 */
    int32_t W, H;        /* raster size (renderer.py:80-81) */
    float tanfovx, tanfovy; /* tan(FoV/2) (renderer.py:76-77).  tanfovx <= 0: CAMERA SLOT -- the kernels read both from device
                             * memory, campos[3] and campos[4] (campos is then 5 floats), so that a captured hipGraph can be
                             * replayed with another camera by rewriting one device buffer (d3ga_amd/cameras.py:CameraSlot) */
    float scale_modifier;
    int32_t antialiasing; /* branch dr_aa [UPSTREAM-RECALL]: opacity x sqrt(max(2.5e-5, det(cov2D) / det(cov2D + 0.3 I))); D3GA passes 0 (renderer.py:92) */
    int32_t prefiltered;  /* accepted, ignored (renderer.py:90) */
    int32_t debug;        /* !=0: synchronise + check after every kernel (renderer.py:91 passes 0) */
    /* D8 (models/cage_net.py:139-159, 247-249: opacity = sigmoid(opacities)): 0 = `opacities` holds activated values
     * (upstream's contract); D3GA_OPACITY_SIGMOID = `opacities` holds LOGITS, the sigmoid is applied on load in
     * d3ga_raster_preprocess and dL_dopacity of d3ga_raster_preprocess_bwd is the gradient w.r.t. the logit. */
    int32_t opacity_activation;
    /* != 0: no backward will follow this forward (inference / no input requires a gradient): d3ga_raster_composite_fwd does
     * not write the per-4x4-block lists its backward walks, and the img buffer only needs d3ga_raster_img_bytes(..., 1) bytes
     * (2 x 4 B per pixel instead of + 128 B per duplicate of capacity).  Calling a backward entry point afterwards is an error
     * (D3GA_E_CONFIG). */
    int32_t forward_only;
    /* != 0: `acc` of the backward entry points is a buffer the CALLER keeps from call to call and guarantees to be all zero
     * on entry: d3ga_raster_backward / _l1 then skip their clear (a 64 B x P fill kernel per backward) and
     * d3ga_raster_preprocess_bwd zero-fills every record it consumed, so that the buffer is all zero again when it returns.
     * All backwards sharing one such buffer must be ordered on one stream. */
    int32_t acc_self_clearing;
    /* View-batched rendering (round 6; no counterpart upstream: the reference renders one camera per call, renderer.py:69).
     * 0 or 1: one camera (everything below reads as before).  k > 1: the SAME Gaussians seen from k cameras are rasterised in ONE
     * grid per stage -- the launches of a single avatar view fill a third of the chip, k views fill it.  Then
     *   viewmatrix / projmatrix are (k,16), campos (k,3) -- (k,5) for camera slots -- radii (k,P);
     *   geom / binning / img are sized by d3ga_raster_scratch_bytes_views and hold k x P records / k x tiles lists: view v's Gaussian
     *   i is record v P + i, its tile (tx, ty) is tile (v gy + ty) gx + tx; W, H, tanfov*, bg are shared by the views;
     *   out_color (k,3,H,W), out_invdepth (k,H,W), dL_dpix (k,3,H,W), the L1 target (k,3,H,W) and its loss = mean over all k images;
     *   d3ga_raster_composite_fwd2 / _bwd2: colors2 stays (P,3) (shared by the views), out_color2 / dL_dpix2 are (k,3,H,W);
     *   acc (k P, D3GA_ACC_STRIDE);  d3ga_raster_preprocess_bwd SUMS dL/dmeans3D, dL/dopacity, dL/dcov3D | (dL/dscales, dL/drots) and a
     *   precomputed colour's gradient over the views, writes dL_dmeans2D per view (k,P,3), and for SH colours needs dL_dcolors
     *   (k,P,3) = the per-view factors of the rank-1 SH gradient, from which dL_dsh (P,M,3), when given, is rebuilt in one pass
     *   (d3ga_sh_grad_from_views) -- one 12 M-byte row per Gaussian and BATCH instead of per view.
     * Every view's image and the summed gradients equal k single-view calls (same kernels, same arithmetic per view).
     * Not available batched (D3GA_E_CONFIG): d3ga_raster_recolor. */
    int32_t n_views;
    /* n_views > 1 only.  != 0: a batch of FRAMES, not only of cameras -- every view has its own geometry (the reference's batch
     * holds frames of different poses, train.py:218-221: the avatar is deformed per frame, its appearance parameters are shared):
     * means3D is (k,P,3) and cov3D_precomp (k,P,6) | scales (k,P,3) + rotations (k,P,4); their gradients are written PER VIEW,
     * (k,P,.), not summed; opacities, shs | colors_precomp stay (P,.) with gradients summed over the views. */
    int32_t per_view_geometry;
    /* n_views > 1, SH colours: rows between consecutive views' factors in the dL_dcolors handed to d3ga_raster_preprocess_bwd (0 = P).
     * The camera-sharded exchange keeps one extra row per view (the view's camera position) so that factors and positions travel in
     * ONE all-gather (d3ga_amd/dist.py): P + 1. */
    int32_t factor_rows;
} d3ga_raster_params;
#define D3GA_OPACITY_SIGMOID 1

/* Byte sizes of the three caller-owned scratch buffers (the analogue of upstream's geomBuffer /
 * binningBuffer / imgBuffer).  d_capacity = capacity in (tile,Gaussian) duplicates of the binning lists.
 * sizes[0]=geom, sizes[1]=binning, sizes[2]=img.  Buffers must be 256-byte aligned. */
int d3ga_raster_scratch_bytes(int32_t P, int32_t W, int32_t H, int64_t d_capacity, int64_t sizes[3]);
/* Bytes of the img buffer alone; forward_only != 0: without the per-block lists (d3ga_raster_params.forward_only). */
int64_t d3ga_raster_img_bytes(int32_t W, int32_t H, int64_t d_capacity, int32_t forward_only);
/* The same for a batch of n_views cameras (d3ga_raster_params::n_views): d_capacity counts the duplicates of ALL views;
 * sizes[2] is the img buffer with (forward_only == 0) or without the per-block lists. */
int d3ga_raster_scratch_bytes_views(int32_t P, int32_t W, int32_t H, int32_t n_views, int64_t d_capacity, int32_t forward_only,
                                    int64_t sizes[3]);

/* Byte offsets of the sections of the binning buffer, for inspection/tests:
 * offsets[0] counters (8 x u32), [1] tile_count (tiles x u32), [2] tile_start (tiles+1 x u32, exclusive prefix),
 * [3] tile_cursor (tiles x u32), [4] keys (d_capacity x u64: depth bits << 32 | index, grouped by tile),
 * [5] point_list (d_capacity x u32: Gaussian indices, each tile's segment ascending in (depth, index)). */
int d3ga_raster_binning_layout(int32_t W, int32_t H, int64_t d_capacity, int64_t offsets[6]);
/* Same for the image buffer: offsets[0] final_T (H*W f32), [1] n_contrib (H*W u32). */
int d3ga_raster_img_layout(int32_t W, int32_t H, int64_t offsets[2]);
/* ... and its per-block lists (absent from a forward_only buffer), for inspection/tests: offsets[0] blk_count (16 x tiles u32: the
 * length of the prefix of a block's list the backward walks = up to the last entry some pixel of the block blended),
 * [1] blk_list (16 x d_capacity {u32 1-based position in the tile's list, u32 Gaussian index}; block b of a tile whose list is
 * [begin, end) starts at element 16 begin + b (end - begin); b = 4 x quadrant + block within the quadrant). */
int d3ga_raster_img_layout_blocks(int32_t W, int32_t H, int64_t offsets[2]);

/* The binning buffer starts with 8 uint32 counters the host may read back after the forward:
 *   [0] D = duplicates required (sum of tiles touched)      [1] 1 if D > d_capacity (lists truncated: re-run)
 *   [2] longest tile list                                    [3] number of visible Gaussians
 *   [4] tiles with 4097..8192 entries   [5] tiles with > 8192 entries   [6] tiles with 2049..4096 entries   [7] reserved */
#define D3GA_CNT_D 0
#define D3GA_CNT_OVERFLOW 1
#define D3GA_CNT_MAXTILE 2
#define D3GA_CNT_VISIBLE 3
#define D3GA_CNT_BIG 4  /* tiles with 4097..8192 entries (72 KB-LDS sort kernel) */
#define D3GA_CNT_HUGE 5 /* tiles with more than 8192 entries (sorted in global memory) */
#define D3GA_CNT_MID 6  /* tiles with 2049..4096 entries (36 KB-LDS sort kernel) */
#define D3GA_CNT_HEAVY 7 /* (non-empty tiles + 9) / 10, or 0 beyond 4096 non-empty tiles: the head of the work order whose tiles get two workgroups each in the compositing backward */

/* R1 per-Gaussian stage + tile histogram.  Exactly one of (shs | colors_precomp) and of
 * ((scales,rotations) | cov3D_precomp) is non-NULL.  viewmatrix/projmatrix are the reference's transposed
 * 4x4 matrices (lib/cameras.py:68-74), campos (3): all DEVICE pointers.  radii (P) int32 is an output. */
int d3ga_raster_preprocess(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                           const float *colors_precomp, const float *opacities, const float *scales,
                           const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                           const float *projmatrix, const float *campos, void *geom, void *binning,
                           int64_t d_capacity, int32_t *radii, d3ga_stream_t stream);
/* R2+R3 tile offsets (scan), scatter of (depth,index) keys, per-tile sort in LDS. */
int d3ga_raster_bin_sort(const d3ga_raster_params *prm, void *geom, void *binning, int64_t d_capacity,
                         d3ga_stream_t stream);
/* R4 front-to-back compositing.  bg (3) device.  out_color (3,H,W); out_invdepth (H,W)|NULL. */
int d3ga_raster_composite_fwd(const d3ga_raster_params *prm, const float *bg, const void *geom, const void *binning,
                              int64_t d_capacity, void *img, float *out_color, float *out_invdepth,
                              d3ga_stream_t stream);
/* R5 back-to-front compositing backward.  dL_dpix (3,H,W).  Accumulates (atomically) into acc (P, D3GA_ACC_STRIDE) float,
 * which the CALLER must have zeroed (d3ga_raster_backward does it itself): [0..2] dL/dmean2D (x,y in NDC-scaled units, z
 * unused), [3..5] dL/dconic (a, b/2, c), [6] dL/dopacity, [7..9] dL/dcolor, [10] dL/d(1/depth) (d3ga_raster_composite_bwd_depth), [11..15] pad.  One record = one 64-byte line:
 * the nine float atomics of a (tile, Gaussian) contribution then meet the memory side as ONE request (with the former
 * 48-byte stride half of the records straddled two lines: compositing backward 246 -> 185 us at C3). */
#define D3GA_ACC_STRIDE 16
int d3ga_raster_composite_bwd(const d3ga_raster_params *prm, const float *bg, const void *geom, const void *binning,
                              int64_t d_capacity, const void *img, const float *dL_dpix, float *acc,
                              d3ga_stream_t stream);
/* The same with the gradient of the inverse-depth image of branch dr_aa [UPSTREAM-RECALL]: dL_dinvdepth (H,W) | NULL, dL_dpix
 * (3,H,W) | NULL (at least one).  The inverse depth takes part as a fourth channel whose per-Gaussian "colour" is 1 / depth:
 * acc[10] receives dL/d(1/depth) = sum alpha T dL/dinvdepth, which d3ga_raster_preprocess_bwd chains into dL/dmeans3D
 * (always: the slot is zero otherwise). */
int d3ga_raster_composite_bwd_depth(const d3ga_raster_params *prm, const float *bg, const void *geom, const void *binning,
                                    int64_t d_capacity, const void *img, const float *dL_dpix, const float *dL_dinvdepth,
                                    float *acc, d3ga_stream_t stream);

/* Two images from one pass (an extension over upstream's rasterizer; the reference's training step renders every package
 * twice with the same geometry and opacities -- RGB, then constant silhouette colours on black, models/trainer.py:102-110):
 * colors2 (P,3) is blended with the same alphas into out_color2 over bg2; alpha, T, the tile lists and the early exit are
 * shared.  The backward adds the second image's dL/dpixel to dL/dalpha; NO gradient is produced for colors2 (constants).
 * geom / binning / img exactly as for the single-image calls. */
int d3ga_raster_composite_fwd2(const d3ga_raster_params *prm, const float *bg, const float *bg2, const void *geom,
                               const float *colors2, const void *binning, int64_t d_capacity, void *img, float *out_color,
                               float *out_color2, float *out_invdepth, d3ga_stream_t stream);
int d3ga_raster_composite_bwd2(const d3ga_raster_params *prm, const float *bg, const float *bg2, const void *geom,
                               const float *colors2, const void *binning, int64_t d_capacity, const void *img,
                               const float *dL_dpix, const float *dL_dpix2, float *acc, d3ga_stream_t stream);

/* L1 image loss fused into the backward (extension; the loss of SURVEY sec. 8d's frame, utils/loss_utils.py:29 l1_loss =
 * mean |image - target|): the compositing backward forms dL/dpixel = g_loss[0] / (3 W H) * sign(image - target) (+ dL_dpix
 * when given, else NULL) per pixel, instead of reading a (3,H,W) gradient image that a separate kernel had to write.
 * image = the forward's out_color; target (3,H,W), or target_cell = device cell holding its address (graph.TensorSlot);
 * g_loss = dL/dloss (device scalar).  d3ga_raster_backward_l1 = clear + this + d3ga_raster_preprocess_bwd. */
/* ... and its VALUE fused into the compositing forward (round 4): d3ga_raster_composite_fwd plus loss[0] = mean |out_color -
 * target|.  Every quadrant wavefront adds |colour - target| of its 64 pixels while the colours are still in registers and
 * leaves one partial (partials: at least 4 * ceil(W/16) * ceil(H/16) floats, scratch); a second one-workgroup kernel adds the
 * partials in index order (bit-reproducible).  Replaces the pass over the finished image (d3ga_l1_mean_fwd_ws). */
int d3ga_raster_composite_fwd_l1(const d3ga_raster_params *prm, const float *bg, const void *geom, const void *binning,
                                 int64_t d_capacity, void *img, float *out_color, float *out_invdepth, const float *target,
                                 const void *target_cell, float *loss, float *partials, d3ga_stream_t stream);
int d3ga_raster_composite_bwd_l1(const d3ga_raster_params *prm, const float *bg, const void *geom, const void *binning,
                                 int64_t d_capacity, const void *img, const float *image, const float *target,
                                 const void *target_cell, const float *g_loss, const float *dL_dpix, float *acc,
                                 d3ga_stream_t stream);
int d3ga_raster_backward_l1(const d3ga_raster_params *prm, const float *means3D, const float *shs, const float *scales,
                            const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                            const float *projmatrix, const float *campos, const float *bg, const void *geom,
                            const void *binning, int64_t d_capacity, const void *img, const float *image,
                            const float *target, const void *target_cell, const float *g_loss, const float *dL_dpix,
                            float *acc, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dopacity, float *dL_dsh,
                            float *dL_dcolors, float *dL_dcov3D, float *dL_dscales, float *dL_drots, d3ga_stream_t stream);
/* Re-render of the SAME geometry (same means3D / covariance / opacities / camera / image size) with other colours: the
 * reference's training step renders an RGB and a silhouette pass from one package (models/trainer.py:102-110).
 * Copies the geometry records of geom_src (a d3ga_raster_preprocess result) to geom_dst and evaluates only the colour
 * (SH from this camera, or colors_precomp).  geom_dst is then used with the binning buffer of the first pass in
 * d3ga_raster_composite_fwd / _bwd / _preprocess_bwd; the binning buffer is only read by those. */
int d3ga_raster_recolor(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                        const float *colors_precomp, const float *campos, const void *geom_src, void *geom_dst,
                        d3ga_stream_t stream);

/* R6 per-Gaussian backward.  Writes every element of the outputs (zeros for culled Gaussians):
 * dL_dmeans3D (P,3), dL_dmeans2D (P,3), dL_dopacity (P,1), and dL_dsh (P,M,3) | dL_dcolors (P,3),
 * dL_dcov3D (P,6) | (dL_dscales (P,3), dL_drots (P,4)).
 * SH path with dL_dsh == NULL and dL_dcolors != NULL: FACTORED SH gradient -- dL_dcolors receives the clamp-masked
 * dL/dcolour (the (P,3) factor of the rank-1 SH gradient, see d3ga_sh_grad_from_views); dL/dmeans3D is complete.
 * cov3D_precomp: the SAME tensor, unchanged, that the forward was given (or NULL with scales / rotations): since round 4 the
 * forward keeps no copy of a precomputed covariance in `geom` (24 B x P less written per frame), the backward reads it here;
 * cov3D_precomp == NULL without (scales, rotations) returns D3GA_E_NULL (ABI 101).
 * `geom` must be what d3ga_raster_preprocess / d3ga_raster_recolor of THIS library left for THIS prm (forward_only == 0): since
 * round 5 it carries, for SH colours with M <= 16 and 3 M a multiple of 4, d(colour)/d(view direction) of every Gaussian (36 B,
 * nine planes of P floats) from which the direction term of dL/dmeans3D is formed -- `shs` is then not read here at all
 * (it must still be non-NULL: it selects the SH path). */
int d3ga_raster_preprocess_bwd(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                               const float *scales, const float *rotations, const float *cov3D_precomp,
                               const float *viewmatrix, const float *projmatrix, const float *campos,
                               const void *geom, const float *acc, float *dL_dmeans3D, float *dL_dmeans2D,
                               float *dL_dopacity, float *dL_dsh, float *dL_dcolors, float *dL_dcov3D,
                               float *dL_dscales, float *dL_drots, d3ga_stream_t stream);

/* Convenience: the whole forward / backward as one call (same stream, no synchronisation). */
int d3ga_raster_forward(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                        const float *colors_precomp, const float *opacities, const float *scales,
                        const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                        const float *projmatrix, const float *campos, const float *bg, void *geom, void *binning,
                        void *img, int64_t d_capacity, float *out_color, int32_t *radii, float *out_invdepth,
                        d3ga_stream_t stream);
int d3ga_raster_backward(const d3ga_raster_params *prm, const float *means3D, const float *shs, const float *scales,
                         const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                         const float *projmatrix, const float *campos, const float *bg, const void *geom,
                         const void *binning, int64_t d_capacity, const void *img, const float *dL_dpix, float *acc,
                         float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dopacity, float *dL_dsh,
                         float *dL_dcolors, float *dL_dcov3D, float *dL_dscales, float *dL_drots,
                         d3ga_stream_t stream);

/* View-sharded training (no reference counterpart: the reference trains on one GPU).  Per view v
 * the SH gradient is rank-1 per Gaussian: dL/dsh[i][k][c] = Y_k(normalize(means3D[i] - campos_v)) * g_v[i][c] with g_v the
 * factored output of d3ga_raster_preprocess_bwd.  Rebuilds  dL_dsh (P,M,3) = scale * sum_v Y(dir_v) (x) g_v  from the
 * gathered factors: g_views + v*g_stride -> (P,3) floats of view v, campos_views + v*campos_stride -> 3 floats
 * (strides in floats).  Coefficients k >= (sh_degree+1)^2 get 0. */
int d3ga_sh_grad_from_views(int32_t P, int32_t M, int32_t sh_degree, int32_t n_views, const float *means3D,
                            const float *g_views, int64_t g_stride, const float *campos_views, int64_t campos_stride,
                            float scale, float *dL_dsh, d3ga_stream_t stream);

/* _C.mark_visible: visible[i] = 1 if view-space z > 0.2 (uint8 output). */
int d3ga_raster_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, uint8_t *visible,
                             d3ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Init-time point location.  Replaces tetra_sampler.compute_bary (lib/cage.py:325-327): for each point the
 * containing tetrahedron (point-in-tet per submodules/tetrahedralize/include/tet/tetrahedron.h:46-71) and its
 * barycentric weights (same header :77-101, order a,b,c,d); points outside every tet take the tet with the
 * largest minimum barycentric weight (weights may be negative) and active[i] = 0.
 *   points (P,3), tetra_corners (T,4,3) -> barys (P,4), tetra_id (P) int32, active (P) uint8.
 * ------------------------------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------------------------
 * Loss tail.  Replaces utils/loss_utils.py:29  l1_loss = |network_output - gt|.mean().
 *   fwd: out[0] = mean |a - b| over n floats (out is zeroed by the call; float atomics across workgroups).
 *   bwd: grad_a = g[0] * sign(a - b) / n   (g: device scalar).   a, b, grad_a 16-byte aligned.
 * ------------------------------------------------------------------------------------------------------- */
int d3ga_l1_mean_fwd(int64_t n, const float *a, const float *b, float *out, d3ga_stream_t stream);
/* The same value in two stages: one partial sum per workgroup into `partials` (>= D3GA_LOSS_PARTIALS floats, contents
 * irrelevant), then one workgroup adds them in index order.  No zero fill, no atomics: the result is reproducible bit for
 * bit, and the call is ~2x faster at image sizes (the 512 same-address atomics of the form above serialise). */
#define D3GA_LOSS_PARTIALS 2048
int d3ga_l1_mean_fwd_ws(int64_t n, const float *a, const float *b, float *out, float *partials, d3ga_stream_t stream);
int d3ga_l1_mean_bwd(int64_t n, const float *a, const float *b, const float *g, float *grad_a, d3ga_stream_t stream);
/* The same two with `b` behind a device CELL: the kernels read its address from *b_cell (device memory) when they start.
 * A captured hipGraph is pointed at another resident target image by rewriting that 8-byte cell instead of copying the
 * image into a static buffer (d3ga_amd/graph.py: TensorSlot).  The tensor the cell names must be 16-byte aligned. */
int d3ga_l1_mean_fwd_ws_cell(int64_t n, const float *a, const float *const *b_cell, float *out, float *partials,
                             d3ga_stream_t stream);
int d3ga_l1_mean_bwd_cell(int64_t n, const float *a, const float *const *b_cell, const float *g, float *grad_a,
                          d3ga_stream_t stream);

/* 11x11 Gaussian-window SSIM, mean over all channels and pixels.  Replaces utils/loss_utils.py:46-86 (ssim / _ssim with
 * window_size = 11, sigma = 1.5, zero padding 5, size_average = True; called at train.py:192).
 *   fwd: img1, img2 (C,H,W) -> out[0] = mean ssim_map (zeroed by the call).  Dm, Dq1, Dq12: optional (C,H,W) outputs
 *        (all three or none) holding d ssim_map / d(w*img1), d(w*img1^2), d(w*img1*img2) for the backward.
 *   bwd: grad_img1 (C,H,W) = g[0]/(C*H*W) * ( w*Dm + 2 img1 (w*Dq1) + img2 (w*Dq12) ),  g: device scalar dL/d(mean).
 * The gradient w.r.t. img2 is the same call with the two images swapped (SSIM is symmetric). */
int d3ga_ssim_fwd(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *out, float *Dm,
                  float *Dq1, float *Dq12, d3ga_stream_t stream);
int d3ga_ssim_bwd(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, const float *Dm,
                  const float *Dq1, const float *Dq12, const float *g, float *grad_img1, d3ga_stream_t stream);
/* The same two kernels with the L1 term of train.py:190-193 riding along (both losses read the same two images):
 * out_l1[0] = mean |img1 - img2| (NULL: skipped); g_l1: device scalar dL/d(l1) (NULL: no L1 term in the gradient). */
int d3ga_ssim_l1_fwd(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *out, float *Dm,
                     float *Dq1, float *Dq12, float *out_l1, d3ga_stream_t stream);
int d3ga_ssim_l1_bwd(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, const float *Dm,
                     const float *Dq1, const float *Dq12, const float *g, const float *g_l1, float *grad_img1,
                     d3ga_stream_t stream);

/* Test hook, not part of the drop-in surface: the 16-lane DPP row scans of the compositing backward.  n multiple of
 * 256; in (n) -> out (8n): for element i (lane l of its row), out[8i+k] = sum over lanes <= l of (k+1) in, k < 4, and
 * out[8i+4+k] = product over lanes <= l of (1 + (k+1)/8 in). */
int d3ga_selftest_row_scan(int n, const float *in, float *out, d3ga_stream_t stream);
/* Test hook, not part of the drop-in surface: the compositing forward's own alpha evaluation (and its "touches the pixel"
 * decision: power <= 0 and alpha >= 1/255) for n listed pairs (Gaussian gid[i], pixel (px[i], py[i])) over the geometry
 * records `geom` of a forward with P Gaussians.  ok (n) u8; alpha (n) f32 or NULL.  Lets a parity test share the product's
 * threshold decisions with the oracle (tests/test_gpu_parity.py: shared decisions). */
int d3ga_selftest_alpha(int32_t P, const void *geom, int n, const int32_t *gid, const int32_t *px, const int32_t *py,
                        uint8_t *ok, float *alpha, d3ga_stream_t stream);

int d3ga_compute_bary(int P, int T, const float *points, const float *tetra_corners, float *barys,
                      int32_t *tetra_id, uint8_t *active, d3ga_stream_t stream);

/* The same search with uniform-grid candidate pruning (bit-identical results for points inside the
 * cage).  cell_start (nx*ny*nz + 1) / cell_tets: per grid cell the tets whose bounding box overlaps it (CSR, int32);
 * origin_h = {ox, oy, oz, cell edge} (HOST array), dims = {nx, ny, nz} (HOST array).  min_weight (P) receives the
 * winner's smallest weight: < 0 means no candidate contains the point -- re-run those through d3ga_compute_bary. */
int d3ga_compute_bary_grid(int P, const float *points, const float *tetra_corners, const int32_t *cell_start,
                           const int32_t *cell_tets, const float *origin_h, const int32_t *dims, float *barys,
                           int32_t *tetra_id, float *min_weight, d3ga_stream_t stream);

/* Init-time scale seed.  Replaces simple_knn._C.distCUDA2 (models/mesh_net.py:22,66) and
 * pytorch3d knn_points(p, p, K=4)[0][0,:,1:].mean(-1) (models/cage_net.py:66): out[i] = mean squared distance of
 * point i to its 3 nearest other points (sum / 3; missing neighbours count as 0, like pytorch3d's padding).  points (P,3) -> out (P).
 * Exhaustive O(P^2). */
int d3ga_knn3_mean_dist2(int P, const float *points, float *out, d3ga_stream_t stream);
/* The same with a uniform grid over the points (cell_start / cell_points: CSR of point indices per cell; origin_h, dims as
 * above, HOST arrays): ring search outwards from the point's cell, exact. */
int d3ga_knn3_mean_dist2_grid(int P, const float *points, const int32_t *cell_start, const int32_t *cell_points,
                              const float *origin_h, const int32_t *dims, float *out, d3ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Field networks: the dense layer of models/mlp.py:39-232 -- every field is
 * z -> [Linear(128) + leaky_relu(0.1)] x (1 + n_layers) -> Linear -- on the matrix cores with f32-equivalent accuracy:
 * every f32 operand is split exactly into three bf16 pieces and the six leading cross products are accumulated in f32
 * (v_mfma_f32_32x32x16_bf16); the dropped products are below 2^-24 |x||w|, the rounding of an f32 fmaf chain.
 *   Y (P, n_out) = act_out( X (P,K) . W + bias ),  act_out(y) = y > 0 ? y : out_slope * y   (out_slope = 1: identity)
 *   sign_out (P, ceil(n_out/32)) uint32, optional: bit (n & 31) of word [r][n >> 5] = (Y[r][n] > 0) -- all the backward
 *   needs of a leaky_relu output.
 *   mask_bits (same layout), optional:  Y (.)= (bit ? 1 : mask_slope)  -- the backward chain: X = dPre of a layer, W its
 *   transposed weights, mask_bits = the sign bits of the layer below, so that Y is THAT layer's dPre (the operand of
 *   its weight gradient dW = dPre^T . input and of the next call); no masked copy is ever written.
 *   panel: the split weights in the kernel's operand order, d3ga_mlp_panel_bytes(K, n_out) bytes, written by
 *   d3ga_mlp_pack_weights from plain f32 weights: weight of input k for output n = W[k*ld_k + n*ld_n]
 *   (an nn.Linear weight (n_out,K): ld_k = 1, ld_n = K; the input-gradient GEMM of the same layer contracts over the
 *   layer's outputs: K := n_out, n_out := K, ld_k = K_layer, ld_n = 1).  Re-pack whenever the weights change.
 *   K <= 128, n_out <= 128; X, panel 16-byte aligned.  bias may be NULL.
 * ------------------------------------------------------------------------------------------------------- */
int64_t d3ga_mlp_panel_bytes(int32_t K, int32_t n_out);          /* < 0: D3GA_E_SIZE */
int d3ga_mlp_pack_weights(int32_t K, int32_t n_out, const float *W, int64_t ld_k, int64_t ld_n, void *panel,
                          d3ga_stream_t stream);
int d3ga_mlp_linear(int32_t P, int32_t K, int32_t n_out, const float *X, const void *panel, const float *bias,
                    float out_slope, uint32_t *sign_out, const uint32_t *mask_bits, float mask_slope, float *Y,
                    d3ga_stream_t stream);

/* One launch for a whole trunk (forward): h_{l+1} = act_l(h_l W_l^T + b_l), l = 0..L-1, with the activations kept in the
 * registers of the wavefront that owns the rows -- no activation is read back between the layers.  Same
 * arithmetic as d3ga_mlp_linear (exact 3-way bf16 split, six products, f32 accumulate).  Layer l: Ks[l] inputs (= Ns[l-1];
 * Ks[0] = K0 <= 128), Ns[l] <= 128 outputs, panels[l] = its weights packed by d3ga_mlp_pack_chain (d3ga_mlp_chain_panel_bytes;
 * the call writes biases[l] -- or zeros: biases / biases[l] may be NULL -- into the 512-byte tail of panels[l], which is why
 * the panels are not const), leaky_relu slope slopes[l] (1 = none) applied to its output; outs[l] (P, Ns[l]) receives that output and
 * signs[l] (or NULL) one bit per output element (P, ceil(Ns[l]/32)) words, bit c of word b = out[row][32 b + c] > 0 before the
 * slope -- exactly what d3ga_mlp_linear writes, so the per-layer backward applies unchanged.  L <= 8.
 * The same launch runs the BACKWARD's input-gradient chain: X = the gradient at the trunk's output, panels[l] = the transposed
 * weights (d3ga_mlp_pack_chain with ld_k / ld_n swapped, bias NULL), slopes[l] = 1, masks[l] (or NULL; masks itself may be
 * NULL) = the sign words the forward wrote for the layer BELOW output l: outs[l] (.)= bit ? 1 : mask_slopes[l] -- outs[l] is
 * then that layer's pre-activation gradient (d3ga_mlp_linear's mask_bits); a call with masks must have no bias, no activation
 * (slopes 1) and no sign output on any layer.  Supported shapes: L >= 2, every layer but the
 * last 128 wide, K0 <= 128, the last one <= 128 wide; D3GA_E_CONFIG otherwise (use d3ga_mlp_linear). */
int64_t d3ga_mlp_chain_panel_bytes(int32_t K, int32_t n_out);
int d3ga_mlp_pack_chain(int32_t K, int32_t n_out, const float *W, int64_t ld_k, int64_t ld_n, void *panel, d3ga_stream_t stream);
int d3ga_mlp_chain_fwd(int32_t P, int32_t K0, const float *X, int32_t L, const int32_t *Ks, const int32_t *Ns,
                       void *const *panels, const float *const *biases, const float *slopes, float *const *outs,
                       uint32_t *const *signs, const uint32_t *const *masks, const float *mask_slopes, d3ga_stream_t stream);
/* Weight and bias gradient of that layer: dW (N,K) = dPre^T . X, db (N) = column sums of dPre (db may be NULL);
 * dPre (P,N) = the gradient at the layer's pre-activation (see above), X (P,K) the layer's input.  Both outputs are zeroed by the call; partial sums meet through float atomics. */
int d3ga_mlp_wgrad(int32_t P, int32_t N, int32_t K, const float *dpre, const float *X, float *dW, float *db,
                   d3ga_stream_t stream);
/* Same, accumulating: dW += dPre^T . X, db += column sums; nothing is zeroed (a chain of layers zeroes ONE flat buffer
 * for all its weight gradients instead of two memsets per layer). */
int d3ga_mlp_wgrad_acc(int32_t P, int32_t N, int32_t K, const float *dpre, const float *X, float *dW, float *db,
                       d3ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * The element-wise ops in front of ColorField.
 *   view_dirs:     dirs (P,3) = (means3D - campos) / |means3D - campos|          models/cage_net.py:233-235
 *   sh4_encoding:  enc (P,16) = real spherical harmonics of degree < 4 on x = 2 dirs - 1: the degree-4
 *                  "SphericalHarmonics" direction encoding of models/mlp.py:166-179 (tiny-cuda-nn, un-vendored:
 *                  restated from its published definition, constants of utils/sh_utils.py:7-24).
 * The backward calls overwrite d_means3D / d_dirs (no accumulation).  enc, d_enc 16-byte aligned.
 * ------------------------------------------------------------------------------------------------------- */
/* Output heads of a field network (models/mlp.py:107-110, 232): pred (P,N) -> n_heads <= 4 consecutive column groups of
 * widths width[h] (sum = N), head h written as a contiguous (P, width[h]) block at out + P * (width[0] + .. + width[h-1])
 * through act[h]: 0 identity, 1 param[h] * tanh(x), 2 sigmoid(x + param[h]).  width, act, param are HOST arrays.
 * bwd: d_pred (P,N) from the heads' gradients g0..g3 (each (P, width[h]) contiguous, NULL = unused head -> zero) and the
 * forward's `out`. */
int d3ga_field_heads_fwd(int32_t P, int32_t N, int32_t n_heads, const int32_t *width, const int32_t *act, const float *param,
                         const float *pred, float *out, d3ga_stream_t stream);
int d3ga_field_heads_bwd(int32_t P, int32_t N, int32_t n_heads, const int32_t *width, const int32_t *act, const float *param,
                         const float *out, const float *g0, const float *g1, const float *g2, const float *g3, float *d_pred,
                         d3ga_stream_t stream);
int d3ga_view_dirs_fwd(int32_t P, const float *means3D, const float *campos, float *dirs, d3ga_stream_t stream);
int d3ga_view_dirs_bwd(int32_t P, const float *means3D, const float *campos, const float *d_dirs, float *d_means3D,
                       d3ga_stream_t stream);
int d3ga_sh4_encoding_fwd(int32_t P, const float *dirs, float *enc, d3ga_stream_t stream);
int d3ga_sh4_encoding_bwd(int32_t P, const float *dirs, const float *d_enc, float *d_dirs, d3ga_stream_t stream);
/* ColorField's per-row input columns in one pass (models/mlp.py:208-226: z's per-row groups when no shadow column is present):
 * x (P, 16 + F) = [ sh4_encoding(dirs) | feats (P,F) ], F a multiple of 4, x / feats 16-byte aligned -- instead of the encoding
 * call and a torch.cat; backward: d_x (P, 16 + F) -> d_dirs (P,3) and d_feats (P,F), either may be NULL (ABI 104). */
int d3ga_color_rows_fwd(int32_t P, int32_t F, const float *dirs, const float *feats, float *x, d3ga_stream_t stream);
int d3ga_color_rows_bwd(int32_t P, int32_t F, const float *dirs, const float *d_x, float *d_dirs, float *d_feats,
                        d3ga_stream_t stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* D3GA_H */

// composite_common.h -- device helpers shared by the compositing kernels (raster_composite.hip: forward;
// raster_composite_scan.hip: backward): the work item -> (tile, quadrant) maps, the alpha evaluation and the 4x4-block
// culling tests (bounding box, then exact ellipse / rectangle).
#pragma once
#include "d3ga_internal.h"

namespace d3ga {

// Debug knobs of the compositing kernels (d3ga_debug_set; include/d3ga.h D3GA_KNOB_*; defaults in raster_api.hip):
//   composite variant: bit 5 (32) work-ordered dispatch -- quadrants / tiles are handed out heaviest tile first (tile_order of
//     the bin stage); bit 7 (128) exact ellipse / block-rectangle test behind the bounding-box test of the forward's culling;
//   merge slots: slots of the backward's tile-level merge cache (256 | 512 | 1024);
//   tile assign: block -> wavefront assignment of the backward's tile kernel: 0 quadrants, 1 interleaved (blocks 8 px apart),
//     2 by list length (the default: DESIGN.md sec. 4);
//   bwd split: the heaviest tiles (in work order) get two workgroups each (raster_composite_scan.hip): -1 (default) = as many as
//     the order kernel counts (a tenth of the non-empty tiles: D3GA_CNT_HEAVY), 0 = none, n > 0 = the n heaviest.
constexpr int kVariantOrdered = 32, kVariantExactCull = 128;
static inline int composite_variant() { return debug_knob(D3GA_KNOB_COMPOSITE_VARIANT); }
static inline int composite_merge_slots() { return debug_knob(D3GA_KNOB_MERGE_SLOTS); }
static inline int composite_tile_assign() { return debug_knob(D3GA_KNOB_TILE_ASSIGN); }
static inline int composite_bwd_split() { return debug_knob(D3GA_KNOB_BWD_SPLIT); }
// L1 image loss fused into the compositing backward: image (3,H,W) = the forward's colour output, target (or the device
// cell that holds its address: graph.TensorSlot), g_loss = dL/dloss (device scalar), inv_n = 1 / (3 H W); image == null: off
struct L1Source { const float *image, *target; const float *const *target_cell; const float *g_loss; float inv_n; };
// L1 VALUE fused into the compositing forward (d3ga_raster_composite_fwd_l1): every quadrant wavefront leaves
// sum |colour - target| * inv_n of its pixels in partials[4 * tile + quadrant]; partials == null: off
struct L1Value { const float *target; const float *const *target_cell; float *partials; float inv_n; };
// loss.hip: out[0] = sum of np partials, added in a fixed order by one workgroup
void launch_sum_partials(int np, const float *partials, float *out, hipStream_t s);
int launch_composite_bwd_scan(const d3ga_raster_params *prm, int gx, int gy, const BinBuf &bin, const GeomBuf &g,
                              const ImgBuf &im, int64_t d_capacity, const float *bg, const float *dL_dpix, float *acc,
                              bool ordered, const float *colors2, const float *bg2, const float *dL_dpix2, const L1Source &l1,
                              hipStream_t s, const float *dL_dinvd = nullptr);

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}

// ---- work item -> (tile, quadrant) with tile rows interleaved over the 8 XCDs ----
// View-batched launches (d3ga_raster_params::n_views = k > 1): the grid spans gx x (k gyv) tiles, tile row v gyv + ty is row ty of
// view v; pixel coordinates (and the splat centres they are compared with) stay LOCAL to the view -- every view computes exactly
// what a single-view launch computes -- and `view` selects the image planes.  One view: gyv = gy, view = 0.
struct Quad {
    bool valid;
    int tile, px, py;            // tile index (of the whole batch), this lane's pixel (local to its view)
    int qx0, qy0;                // quadrant origin in pixels (local)
    int quad;                    // quadrant index inside the tile (0..3)
    int view;
};
__device__ __forceinline__ void quad_place(Quad &q, int tx, int ty, int quad, int gyv) {
    q.view = ty / gyv;
    const int tyl = ty - q.view * gyv;
    q.quad = quad;
    q.qx0 = tx * kTile + ((quad & 1) << 3);
    q.qy0 = tyl * kTile + ((quad >> 1) << 3);
    const int lane = threadIdx.x & 63;
    q.px = q.qx0 + (lane & 7);
    q.py = q.qy0 + (lane >> 3);
}
__device__ __forceinline__ Quad quad_of_block(int gx, int gy, int gyv) {
    Quad q;
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int per_row = gx * 4;
    const int k = slot / per_row, rem = slot - k * per_row;
    const int ty = xcd + 8 * k, tx = rem >> 2, quad = rem & 3;
    q.valid = ty < gy;
    q.tile = ty * gx + tx;
    quad_place(q, tx, q.valid ? ty : 0, quad, gyv);
    return q;
}
static inline int quad_grid(int gx, int gy) { return 8 * ((gy + 7) / 8) * gx * 4; }
// Work-ordered mapping: tile rank k (tile_order: descending list length) -> blocks b, b+8, b+16, b+24 of one XCD (the four
// quadrants of a tile keep sharing an L2), ranks dealt round-robin over the XCDs.
__device__ __forceinline__ Quad quad_of_block_ordered(int gx, int tiles, int gyv, const uint32_t *__restrict__ order) {
    Quad q;
    const int b = blockIdx.x;
    const int k = (b & 7) + 8 * (b >> 5), quad = (b >> 3) & 3;
    q.valid = k < tiles;
    q.tile = q.valid ? (int)order[k] : 0;
    const int ty = q.tile / gx, tx = q.tile - ty * gx;
    quad_place(q, tx, ty, quad, gyv);
    return q;
}
static inline int quad_grid_ordered(int tiles) { return 32 * ((tiles + 7) / 8); }

// alpha of one splat on one pixel, branch-free: ok <=> the splat touches the pixel (power <= 0 and alpha >= 1/255).
// The constants of  G = exp(-1/2 (a dx^2 + c dy^2) - b dx dy)  are folded into the conic once per entry:
//   q = (-1/2 log2(e) a, -log2(e) b, -1/2 log2(e) c),   G = exp2(dx (q.a dx + q.b dy) + q.c dy^2)
// -> two multiplies and two FMAs in front of v_exp_f32 instead of seven multiplies/FMAs and the log2(e) scaling; callers
// whose lanes share dy (the entry-per-lane backward) hoist tb = q.b dy and tc = q.c dy^2 out of the pixel loop.  Forward and
// backward use THIS expression tree (explicit fmaf), so both see bit-identical alphas.
constexpr float kLog2e = 1.4426950408889634f;
struct ConicQ { float a, b, c; };
__device__ __forceinline__ ConicQ conic_q(float a, float b, float c) {
    return ConicQ{(-0.5f * kLog2e) * a, (-kLog2e) * b, (-0.5f * kLog2e) * c};
}
__device__ __forceinline__ void splat_eval_q(float dx, float tb, float tc, float qa, float o, float &alpha, float &G, bool &ok) {
    const float p = fmaf(dx, fmaf(qa, dx, tb), tc);       // log2 of G
    G = __builtin_amdgcn_exp2f(p);
    alpha = fminf(kAlphaMax, o * G);
    ok = (p <= 0.0f) && (alpha >= kAlphaMin);
}
__device__ __forceinline__ void splat_eval_q(float dx, float dy, const ConicQ &q, float o, float &alpha, float &G, bool &ok) {
    splat_eval_q(dx, q.b * dy, (q.c * dy) * dy, q.a, o, alpha, G, ok);
}

struct RowGeom {
    int row, px, py;
    float x0, y0;     // sub-block origin
};
__device__ __forceinline__ RowGeom row_geom(const Quad &q, int lane) {
    RowGeom g;
    g.row = lane >> 4;
    const int l = lane & 15;
    const int sx = q.qx0 + ((g.row & 1) << 2), sy = q.qy0 + ((g.row >> 1) << 2);
    g.px = sx + (l & 3);
    g.py = sy + (l >> 2);
    g.x0 = (float)sx; g.y0 = (float)sy;
    return g;
}
// the four 4x4 sub-blocks of the quadrant at (bx0, by0) share their x / y range tests: 8 compares instead of 16
struct BlockHits { bool r0, r1, r2, r3; };
__device__ __forceinline__ BlockHits block_hits4(float cx, float cy, float hx, float hy, float bx0, float by0) {
    const bool vis = !(hx < 0.0f);
    const float xl = cx - hx, xr = cx + hx, yt = cy - hy, yb = cy + hy;
    const bool x0 = vis && !(xr < bx0) && !(xl > bx0 + 3.0f), x1 = vis && !(xr < bx0 + 4.0f) && !(xl > bx0 + 7.0f);
    const bool y0 = !(yb < by0) && !(yt > by0 + 3.0f), y1 = !(yb < by0 + 4.0f) && !(yt > by0 + 7.0f);
    BlockHits h;
    h.r0 = x0 && y0; h.r1 = x1 && y0; h.r2 = x0 && y1; h.r3 = x1 && y1;
    return h;
}
// Exact refinement of block_hits4 (kVariantExactCull).  The bounding box of the alpha >= 1/255 ellipse q(d) <= tau is loose for
// elongated, rotated splats; every (entry, block) pair that survives costs the backward 16 pixel steps and one atomic
// request.  A pair is kept iff  min over the block's rectangle of q  <= tau.  q = A dx^2 + 2B dx dy + C dy^2 is convex, so
// with (fx, fy) = the rectangle's point closest to the centre per axis (clamp of 0 into [xa,xb] / [ya,yb], all relative to
// the centre) the minimum is  min( min_dy q(fx, dy), min_dx q(dx, fy) )  with the inner minimiser clamped to the edge:
// centre inside the x range -> fx = 0 and the second term is never larger; inside both -> 0.  tau = 2 ln(255 o)
// (inflated 0.1 % + 1e-4; the box extents sqrt(tau C/det), sqrt(tau A/det) by 0.1 % + 0.02 px); pixels are a subset of the continuous rectangle, so the test stays conservative.
struct SplatCull { float tau, hx, hy, nbc, nba; };     // nbc = -B/C, nba = -B/A
__device__ __forceinline__ SplatCull splat_cull(float A, float B, float C, float o) {
    SplatCull c;
    if (o * 255.0f < 1.0f) { c.tau = -1.0f; c.hx = -1.0f; c.hy = -1.0f; c.nbc = 0.f; c.nba = 0.f; return c; }
    c.tau = 2.0f * __logf(255.0f * o) * 1.001f + 1e-4f;
    const float idet = __builtin_amdgcn_rcpf(A * C - B * B);
    c.hx = __builtin_amdgcn_sqrtf(c.tau * C * idet) * 1.001f + 0.02f;
    c.hy = __builtin_amdgcn_sqrtf(c.tau * A * idet) * 1.001f + 0.02f;
    c.nbc = -B * __builtin_amdgcn_rcpf(C);
    c.nba = -B * __builtin_amdgcn_rcpf(A);
    return c;
}
__device__ __forceinline__ float clamp3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
__device__ __forceinline__ BlockHits block_hits4_exact(float cx, float cy, float A, float B, float C, const SplatCull &c, float bx0,
                                                       float by0, const BlockHits &box) {
    // block column j = 0,1: x range [bx0 + 4j, bx0 + 4j + 3]; block line i = 0,1 likewise; everything relative to the centre
    const float xa0 = bx0 - cx, xb0 = xa0 + 3.0f, xa1 = xa0 + 4.0f, xb1 = xa0 + 7.0f;
    const float ya0 = by0 - cy, yb0 = ya0 + 3.0f, ya1 = ya0 + 4.0f, yb1 = ya0 + 7.0f;
    const float fx0 = clamp3(0.f, xa0, xb0), fx1 = clamp3(0.f, xa1, xb1);
    const float fy0 = clamp3(0.f, ya0, yb0), fy1 = clamp3(0.f, ya1, yb1);
    const float tB = 2.0f * B;
    // min over dy in [ya, yb] of q(fx, dy)
    // q is a sum of cancelling terms (a splat a thousand pixels long: terms ~1e6, tau ~11): what is compared is q minus a bound
    // on its float32 rounding error, 1e-6 x the sum of the terms' magnitudes, so that the test stays conservative for ANY
    // footprint (the fixed 0.1 % + 1e-4 inflation of tau alone covers |terms| up to ~1e4 only)
    auto qx = [&](float fx, float ya, float yb) {
        const float dy = clamp3(c.nbc * fx, ya, yb);
        const float t0 = A * fx * fx, t1 = tB * fx * dy, t2 = C * dy * dy;
        return (t0 + t1 + t2) - 1e-6f * (fabsf(t0) + fabsf(t1) + fabsf(t2));
    };
    auto qy = [&](float fy, float xa, float xb) {
        const float dx = clamp3(c.nba * fy, xa, xb);
        const float t0 = C * fy * fy, t1 = tB * fy * dx, t2 = A * dx * dx;
        return (t0 + t1 + t2) - 1e-6f * (fabsf(t0) + fabsf(t1) + fabsf(t2));
    };
    BlockHits h;
    // (!(q > tau): NaNs answer "relevant", like the box test)
    h.r0 = box.r0 && !(fminf(qx(fx0, ya0, yb0), qy(fy0, xa0, xb0)) > c.tau);
    h.r1 = box.r1 && !(fminf(qx(fx1, ya0, yb0), qy(fy0, xa1, xb1)) > c.tau);
    h.r2 = box.r2 && !(fminf(qx(fx0, ya1, yb1), qy(fy1, xa0, xb0)) > c.tau);
    h.r3 = box.r3 && !(fminf(qx(fx1, ya1, yb1), qy(fy1, xa1, xb1)) > c.tau);
    return h;
}
__device__ __forceinline__ int lanes_below(unsigned long long m) {   // popcount of m restricted to lower lanes
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ uint32_t row_max_u32(uint32_t v) {       // every lane <- max over its 16-lane row
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); v = max(v, t);
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); v = max(v, t);
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); v = max(v, t);
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true); v = max(v, t);
    return v;
}

// Builds the four per-row lists of one staged batch: list r holds the LDS byte offsets (16 * staged lane) of the entries
// that hit block r, in list order, and is padded with kNullRec -- the offset of a record with opacity 0, which never blends --
// so the blend loop needs neither a per-row count nor an index mask.  Returns the trip count (longest list); m[] are the
// four hit masks.  (LDS instructions of one wavefront execute in order: the padding lands before the entries.)
constexpr uint16_t kNullRec = 64 * 16;
constexpr int kListStride = 66;       // 64 entries + two null records: the blend loop reads its offsets one iteration ahead
__device__ __forceinline__ int build_row_lists(uint16_t (*s_list)[kListStride], bool r0, bool r1, bool r2, bool r3, int lane,
                                               unsigned long long (&m)[4]) {
    m[0] = __ballot(r0); m[1] = __ballot(r1); m[2] = __ballot(r2); m[3] = __ballot(r3);
    s_list[0][lane] = kNullRec; s_list[1][lane] = kNullRec; s_list[2][lane] = kNullRec; s_list[3][lane] = kNullRec;
    const uint16_t mine = (uint16_t)(lane * 16);
    if (r0) s_list[0][lanes_below(m[0])] = mine;
    if (r1) s_list[1][lanes_below(m[1])] = mine;
    if (r2) s_list[2][lanes_below(m[2])] = mine;
    if (r3) s_list[3][lanes_below(m[3])] = mine;
    return max(max(__popcll(m[0]), __popcll(m[1])), max(__popcll(m[2]), __popcll(m[3])));
}

}  // namespace d3ga

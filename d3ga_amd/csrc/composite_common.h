// composite_common.h -- device helpers shared by the compositing kernels (raster_composite.hip: forward;
// raster_composite_scan.hip: backward): the work item -> (tile, quadrant) maps, the alpha evaluation and the 4x4-block
// culling tests (bounding box, then exact ellipse / rectangle).
#pragma once
#include "d3ga_internal.h"

#include <stdlib.h>

namespace d3ga {

// D3GA_COMPOSITE_VARIANT (A/B knob, read once; the other bits selected kernels that no longer exist):
//   bit 5 (32)  work-ordered dispatch: quadrants are handed out heaviest tile first (tile_order of the bin stage);
//   bit 7 (128) exact ellipse / block-rectangle test behind the bounding-box test of the forward's culling.
constexpr int kVariantOrdered = 32, kVariantExactCull = 128;
constexpr int kDefaultCompositeVariant = kVariantOrdered | kVariantExactCull;
static inline int composite_variant() {
    static const int v = [] {
        const char *e = getenv("D3GA_COMPOSITE_VARIANT");
        return e ? atoi(e) : kDefaultCompositeVariant;
    }();
    return v;
}
constexpr int kDefaultMergeSlots = 512;
static inline int composite_merge_slots() {        // A/B knob: slots of the tile-level merge cache (256 | 512 | 1024)
    static const int v = [] {
        const char *e = getenv("D3GA_MERGE_SLOTS");
        return e ? atoi(e) : kDefaultMergeSlots;
    }();
    return v;
}
// block -> wavefront assignment of the backward's tile kernel: 0 quadrants, 1 interleaved (blocks 8 px apart), 2 by list
// length (the default: DESIGN.md sec. 4; d3ga_debug_defaults() reports what THIS library runs and a test pins it)
constexpr int kDefaultTileAssign = 2;
static inline int composite_tile_assign() {        // A/B knob (D3GA_TILE_ASSIGN)
    static const int v = [] {
        const char *e = getenv("D3GA_TILE_ASSIGN");
        return e ? atoi(e) : kDefaultTileAssign;
    }();
    return v;
}
// Backward: the heaviest tiles (in work order) get two workgroups each (raster_composite_scan.hip).  D3GA_BWD_SPLIT: -1 (default) = as many
// as the order kernel counts (a tenth of the non-empty tiles: D3GA_CNT_HEAVY), 0 = none, n > 0 = the n heaviest
constexpr int kDefaultBwdSplit = -1;
static inline int composite_bwd_split() {
    static const int v = [] { const char *e = getenv("D3GA_BWD_SPLIT"); return e ? atoi(e) : kDefaultBwdSplit; }();
    return v;
}
// A/B knob (D3GA_FWD_LDS_TOTAL / D3GA_BWD_LDS_TOTAL, bytes): pad a kernel's LDS allocation up to this total with dynamic shared
// memory -- limits the workgroups resident per CU (160 KB / total) without touching the code: fewer, faster waves per SIMD
// and more dispatch rounds (the hardware dispatcher hands out workgroups in launch order as slots free up).  0 / unset: no pad.
static inline unsigned lds_pad_bytes(const void *kernel, const char *env_name) {
    const char *e = getenv(env_name);
    const long total = e ? atol(e) : 0;
    if (total <= 0) return 0u;
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, kernel) != hipSuccess) return 0u;
    return total > (long)a.sharedSizeBytes ? (unsigned)(total - (long)a.sharedSizeBytes) : 0u;
}
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

// raster_composite_lists.hip: the two-launch forward (tile_cull_kernel + composite_fwd_lists_kernel)
int launch_composite_fwd_lists(const d3ga_raster_params *prm, int gx, int gy, const BinBuf &bin, const GeomBuf &g, const ImgBuf &im,
                               int64_t d_capacity, const float *bg, float *out_color, float *out_invdepth, const float *colors2,
                               const float *bg2, float *out_color2, bool ordered, bool exact, const L1Value &l1v, bool lists_ready,
                               hipStream_t s);
// D3GA_FWD_IMPL (A/B knob, read once): 0 the one-launch quadrant forward (raster_composite.hip); 1 the two-launch forward of
// raster_composite_lists.hip for renders that are followed by a backward (measured, not faster: its list pass is latency-bound --
// DESIGN.md sec. 4); 2 the lists blend of raster_composite_lists.hip over block lists the per-tile SORT emitted
// (d3ga_raster_bin_sort_lists + d3ga_raster_params::block_lists; a composite call whose params do not say so builds them with the
// list pass of 1).  Renders with forward_only always use the one-launch forward: no block lists are allocated.
constexpr int kDefaultFwdImpl = 0;
static inline int composite_fwd_impl_kind() {
    static const int v = [] {
        const char *e = getenv("D3GA_FWD_IMPL");
        return e ? atoi(e) : kDefaultFwdImpl;
    }();
    return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}

// ---- work item -> (tile, quadrant) with tile rows interleaved over the 8 XCDs ----
struct Quad {
    bool valid;
    int tile, px, py;            // tile index, this lane's pixel
    int qx0, qy0;                // quadrant origin in pixels
    int quad;                    // quadrant index inside the tile (0..3)
};
__device__ __forceinline__ Quad quad_of_block(int gx, int gy) {
    Quad q;
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int per_row = gx * 4;
    const int k = slot / per_row, rem = slot - k * per_row;
    const int ty = xcd + 8 * k, tx = rem >> 2, quad = rem & 3;
    q.valid = ty < gy;
    q.tile = ty * gx + tx;
    q.quad = quad;
    q.qx0 = tx * kTile + ((quad & 1) << 3);
    q.qy0 = ty * kTile + ((quad >> 1) << 3);
    const int lane = threadIdx.x & 63;
    q.px = q.qx0 + (lane & 7);
    q.py = q.qy0 + (lane >> 3);
    return q;
}
static inline int quad_grid(int gx, int gy) { return 8 * ((gy + 7) / 8) * gx * 4; }
// Work-ordered mapping: tile rank k (tile_order: descending list length) -> blocks b, b+8, b+16, b+24 of one XCD (the four
// quadrants of a tile keep sharing an L2), ranks dealt round-robin over the XCDs.
__device__ __forceinline__ Quad quad_of_block_ordered(int gx, int tiles, const uint32_t *__restrict__ order) {
    Quad q;
    const int b = blockIdx.x;
    const int k = (b & 7) + 8 * (b >> 5), quad = (b >> 3) & 3;
    q.valid = k < tiles;
    q.tile = q.valid ? (int)order[k] : 0;
    const int ty = q.tile / gx, tx = q.tile - ty * gx;
    q.quad = quad;
    q.qx0 = tx * kTile + ((quad & 1) << 3);
    q.qy0 = ty * kTile + ((quad >> 1) << 3);
    const int lane = threadIdx.x & 63;
    q.px = q.qx0 + (lane & 7);
    q.py = q.qy0 + (lane >> 3);
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
// ---- block spans of a splat (round 5) ----
// The 4x4-pixel blocks (global block grid: block column C covers pixel columns 4C .. 4C+3, block line R pixel lines 4R .. 4R+3)
// that the alpha >= 1/255 ellipse {q <= tau} can touch, as ONE column interval per block line.  q is convex, so the ellipse cut
// by the slab of a block line spans [l(dyl), r(dyr)] with  r(dy) = (-B dy + sqrt(tau A - det dy^2)) / A  (concave, maximiser
// dy* = -(B/C) hx: the rightmost point) taken at dy* clamped into the slab, and l likewise (convex, minimiser -dy*); a block is in
// iff its pixel columns meet that interval.  tau and the box extents are the inflated ones of splat_cull() (0.1 % + 1e-4; 0.1 % +
// 0.02 px), the interval is padded by 0.1 % of the extent + 0.02 px, everything is intersected with the bounding box: conservative
// -- the same set as block_hits4() + block_hits4_exact() up to rounding.  exact == false: the bounding box on every line.
// Written once per Gaussian by preprocess (GeomBuf::span) and decoded per (tile | quadrant, entry) with integer arithmetic by the
// compositing forward, instead of the geometric test per (quadrant, entry).  Layout (16 bytes):
//   x: block line R0 of the first line (i16; kSpanBigR0: see below) | block column C0 the intervals are relative to (i16) << 16
//   y, z, w: up to twelve lines, line k in BYTE k:  (lo + 1) | hi << 4  = columns C0 + lo .. C0 + hi (lo <= 14, hi <= 15); low
//      nibble 0: empty line (lines past the last one are stored empty)
//   R0 == kSpanBigR0: too tall (> 12 block lines: half height > 22 px) / wide (> 16 columns) / far off for this record -- the reader
//      falls back to the geometric test.  (At C3 2.4 % of the splats are taller than SIX block lines: a first version with 16-bit
//      fields and six lines sent 79 % of the forward's 64-survivor batches through the fallback.)
constexpr int kSpanBigR0 = 0x7fff, kSpanLines = 12;
__device__ __forceinline__ uint4 span_big() { return make_uint4((uint32_t)kSpanBigR0, 0u, 0u, 0u); }
__device__ __forceinline__ bool span_is_big(const uint4 &sp) { return (sp.x & 0xffffu) == (uint32_t)kSpanBigR0; }
__device__ __forceinline__ uint4 splat_spans(float cx, float cy, float A, float B, float C, float o, bool exact) {
    const SplatCull sc = splat_cull(A, B, C, o);
    if (sc.hx < 0.0f) return make_uint4(0u, 0u, 0u, 0u);
    if (!(sc.hx < 500.f) || !(sc.hy < 500.f) || !(fabsf(cx) < 100000.f) || !(fabsf(cy) < 100000.f)) return span_big();
    // block lines R with 4R <= cy + hy and 4R + 3 >= cy - hy; columns of the box likewise
    const int R0 = (int)ceilf((cy - sc.hy - 3.0f) * 0.25f), R1 = (int)floorf((cy + sc.hy) * 0.25f);
    const int C0 = (int)ceilf((cx - sc.hx - 3.0f) * 0.25f), C1 = (int)floorf((cx + sc.hx) * 0.25f);
    const int K = R1 - R0 + 1;
    if (K <= 0 || C1 < C0) return make_uint4(0u, 0u, 0u, 0u);
    if (K > kSpanLines || C1 - C0 > 15) return span_big();
    const float det = A * C - B * B, rA = __builtin_amdgcn_rcpf(A), tauA = sc.tau * A;
    const float dys = sc.nbc * sc.hx;
    const float pad = 1e-3f * sc.hx + 0.02f;
    uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < kSpanLines; ++k) {
        if (k >= 4 && __builtin_amdgcn_ballot_w64(k < K) == 0ull) break;     // (uniform: no lane of the wavefront has a line k)
        int lo = C0, hi = C1;
        if (exact) {
            const float ya = (float)(4 * (R0 + k)) - cy, yb = ya + 3.0f;
            const float dyr = clamp3(dys, ya, yb), dyl = clamp3(-dys, ya, yb);
            const float sr = __builtin_amdgcn_sqrtf(fmaxf(0.0f, tauA - det * dyr * dyr));
            const float sl = __builtin_amdgcn_sqrtf(fmaxf(0.0f, tauA - det * dyl * dyl));
            const float rmax = (sr - B * dyr) * rA + pad, lmin = (-sl - B * dyl) * rA - pad;
            const int l2 = (int)ceilf((cx + lmin - 3.0f) * 0.25f), h2 = (int)floorf((cx + rmax) * 0.25f);
            if (l2 > lo && lmin == lmin) lo = l2;                     // (NaN -> the box)
            if (h2 < hi && rmax == rmax) hi = h2;
        }
        const uint32_t f = (k < K && lo <= hi) ? ((uint32_t)(lo - C0 + 1) | ((uint32_t)(hi - C0) << 4)) : 0u;
        w[k >> 2] |= f << (8 * (k & 3));                              // (k is a constant here: no indexed access)
    }
    return make_uint4(((uint32_t)R0 & 0xffffu) | ((uint32_t)C0 << 16), w[0], w[1], w[2]);
}
// the 8-bit field of span line idx (empty outside 0 .. kSpanLines - 1).  Shifts, not a select over the record's words: the
// compiler turns `idx < 4 ? sp.y : ...` into an indexed load of the record, which puts the record into SCRATCH memory (measured:
// the forward went from 79 to 106 us)
__device__ __forceinline__ uint32_t span_line(const uint4 &sp, int idx) {
    const unsigned long long lo8 = (unsigned long long)sp.y | ((unsigned long long)sp.z << 32);
    const uint32_t a = (uint32_t)(lo8 >> (8 * (idx & 7))), b = sp.w >> (8 * (idx & 3));
    const uint32_t e = ((idx & 8) ? b : a) & 0xffu;
    return (unsigned)idx < (unsigned)kSpanLines ? e : 0u;
}
// column bits (bit i = block column Cq0 + i, i < ncols) of one span line field
__device__ __forceinline__ uint32_t span_cols(uint32_t e, int cb, int ncols) {
    const int lo = max(cb + (int)(e & 15u) - 1, 0), hi = min(cb + (int)(e >> 4), ncols - 1);
    return ((e & 15u) != 0u && lo <= hi) ? ((2u << hi) - (1u << lo)) : 0u;
}
// the 16-bit block mask (bit 4 * quadrant + block within it, as the block lists are numbered) of a span record inside the tile whose
// first block column / line are Ct0 / Rt0; the record must not be big
__device__ __forceinline__ uint32_t span_mask16(const uint4 &sp, int Ct0, int Rt0) {
    const int R0 = (int)(int16_t)(sp.x & 0xffffu), C0 = (int)sp.x >> 16;
    const int k0 = Rt0 - R0, cb = C0 - Ct0;
    uint32_t mask = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t cols = span_cols(span_line(sp, k0 + j), cb, 4) & 15u;
        const int p0 = 8 * (j >> 1) + 2 * (j & 1);                    // block (i, j) -> bit 8 (j >> 1) + 2 (j & 1) + {0, 1, 4, 5}[i]
        mask |= ((cols & 3u) << p0) | (((cols >> 2) & 3u) << (p0 + 4));
    }
    return mask;
}
// the 4-bit mask of the 2x2 blocks of one quadrant (bit = block column + 2 x block line inside the quadrant: the forward's row
// index) whose first block column / line are Cq0 / Rq0
__device__ __forceinline__ uint32_t span_mask4(const uint4 &sp, int Cq0, int Rq0) {
    const int R0 = (int)(int16_t)(sp.x & 0xffffu), C0 = (int)sp.x >> 16;
    const int k0 = Rq0 - R0, cb = C0 - Cq0;
    return (span_cols(span_line(sp, k0), cb, 2) & 3u) | ((span_cols(span_line(sp, k0 + 1), cb, 2) & 3u) << 2);
}

// (bx, by) = block column / line inside the tile -> the block index both directions use: 4 * quadrant + block within it
__device__ __forceinline__ int blk_of(int bx, int by) { return 4 * ((bx >> 1) + 2 * (by >> 1)) + ((bx & 1) + 2 * (by & 1)); }

// Blocks of the tile at (tx0, ty0) that the splat (centre, conic | opacity) can touch, as a 16-bit mask over blk_of().
// Box test = block_hits4() per block column / line; exact: per block LINE j the x interval of {q <= tau} inside the slab
// y in [4j, 4j+3] -- q(dx, dy) = A dx^2 + 2 B dx dy + C dy^2 is convex, so the intersection of the ellipse with the slab spans
// [l(dyl), r(dyr)] with  r(dy) = (-B dy + sqrt(tau A - det dy^2)) / A  (concave; its maximiser dy* = -(B/C) hx is the
// ellipse's rightmost point) taken at dy* clamped into the slab, likewise l (convex, minimiser -dy*); a block is kept iff its
// pixel columns meet that interval.  Same set as block_hits4_exact() up to rounding (tau is inflated by 0.1 % + 1e-4 there and
// here; the interval is padded by 0.1 % of the splat's extent + 0.02 px), so it stays conservative for ANY footprint.
__device__ __forceinline__ uint32_t block_mask16(float cx, float cy, float A, float B, float C, float o, float tx0, float ty0, bool exact) {
    const SplatCull sc = splat_cull(A, B, C, o);
    if (sc.hx < 0.0f) return 0u;                                      // alpha < 1/255 everywhere (NaN: falls through = relevant)
    const float xl = cx - sc.hx, xr = cx + sc.hx, yt = cy - sc.hy, yb = cy + sc.hy;
    uint32_t colbox = 0u, rowbox = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float c0 = tx0 + 4.0f * (float)i, r0 = ty0 + 4.0f * (float)i;
        colbox |= (!(xr < c0) && !(xl > c0 + 3.0f)) ? (1u << i) : 0u;
        rowbox |= (!(yb < r0) && !(yt > r0 + 3.0f)) ? (1u << i) : 0u;
    }
    if (colbox == 0u || rowbox == 0u) return 0u;
    const float det = A * C - B * B, rA = __builtin_amdgcn_rcpf(A), tauA = sc.tau * A;
    const float dys = sc.nbc * sc.hx;                                 // dy of the rightmost point (-dys: leftmost)
    const float pad = 1e-3f * sc.hx + 0.02f;
    uint32_t mask = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t cols = colbox;
        if (exact) {
            const float ya = (ty0 + 4.0f * (float)j) - cy, yb2 = ya + 3.0f;
            const float dyr = clamp3(dys, ya, yb2), dyl = clamp3(-dys, ya, yb2);
            const float sr = __builtin_amdgcn_sqrtf(fmaxf(0.0f, tauA - det * dyr * dyr));
            const float sl = __builtin_amdgcn_sqrtf(fmaxf(0.0f, tauA - det * dyl * dyl));
            const float rmax = (sr - B * dyr) * rA + pad, lmin = (-sl - B * dyl) * rA - pad;
            cols = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xa = (tx0 + 4.0f * (float)i) - cx;
                cols |= (!(rmax < xa) && !(lmin > xa + 3.0f)) ? (1u << i) : 0u;     // (NaN: relevant)
            }
            cols &= colbox;
        }
        cols = ((rowbox >> j) & 1u) ? cols : 0u;
        // block (i, j) -> bit 8 (j >> 1) + 2 (j & 1) + {0, 1, 4, 5}[i]
        const int p0 = 8 * (j >> 1) + 2 * (j & 1);
        mask |= ((cols & 3u) << p0) | (((cols >> 2) & 3u) << (p0 + 4));
    }
    return mask;
}

// the rare path of the mask decoders (tile_cull_kernel, tile_scatter_kernel, the huge-list emission: a splat too large for a span record): out of line, so that its ~250 instructions and their
// registers are not replicated into every sub-round of the kernel
static __device__ __attribute__((noinline)) uint32_t block_mask16_slow(uint32_t id, const float4 *__restrict__ xyh, const float4 *__restrict__ conic_o,
                                                                 float tx0, float ty0, bool exact) {
    const float4 h = xyh[id], co = conic_o[id];
    return block_mask16(h.x, h.y, co.x, co.y, co.z, co.w, tx0, ty0, exact);
}

// inclusive prefix sum over the 64 lanes (DPP: row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15 / :31 across them);
// used on four 8-bit counters packed in a dword (a lane contributes 0 or 1 per counter: no carry between the fields)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // rows 1, 3 += lane 15 of rows 0, 2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // rows 2, 3 += lane 31
    return v;
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

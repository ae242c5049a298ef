// composite_common.h -- device helpers shared by the compositing kernels (raster_composite.hip: forward and the
// row-segmented backward; raster_composite_scan.hip: the entry-per-lane backward): DPP reductions, the work item ->
// (tile, quadrant) maps, the alpha evaluation and the conservative 4x4-block culling test.
#pragma once
#include "d3ga_internal.h"

#include <stdlib.h>

namespace d3ga {

// D3GA_COMPOSITE_VARIANT (A/B knob, read once): bit 0 forward, bit 1 backward of the 64-lane kernels fetch entry records
// through a wave-private LDS slab instead of v_readlane broadcasts; bit 2 forward, bit 3 backward use the row-segmented
// kernels (four 4x4 blocks per wavefront); bit 4 unused; bit 5 work-ordered dispatch (tile_order); bit 6 (with bit 2) the
// forward writes the per-block culled lists and the backward is the entry-per-lane kernel of raster_composite_scan.hip.
// bit 7: exact ellipse / block-rectangle test behind the bounding-box test of the forward's culling.
constexpr int kDefaultCompositeVariant = 255;
static inline int composite_variant() {
    static const int v = [] {
        const char *e = getenv("D3GA_COMPOSITE_VARIANT");
        return e ? atoi(e) : kDefaultCompositeVariant;
    }();
    return v;
}
int launch_composite_bwd_scan(const d3ga_raster_params *prm, int gx, int gy, const BinBuf &bin, const GeomBuf &g,
                              const ImgBuf &im, int64_t d_capacity, const float *bg, const float *dL_dpix, float *acc,
                              bool ordered, const float *colors2, const float *bg2, const float *dL_dpix2, hipStream_t s);

// ---- wavefront (64 lanes) reductions through DPP ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}
// single value; total broadcast to all lanes
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);   // row_mirror        -> every lane holds its 16-lane row sum
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3 -> row 3 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// NV values at once: the chains are independent, so the scheduler interleaves them and the DPP wait states of one
// chain are filled by the others.  Totals end up in every lane of row 3 (lanes 48..63).
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
// ---- nine values at once: reduce-scatter with the gfx950 lane-swap instructions ----
// v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31 of the second; adding the two results
// leaves the half-wave sums of the first value in lanes 0..31 and of the second in lanes 32..63 (2 instructions retire
// one of two values).  v_permlane16_swap does the same for odd/even 16-lane rows.  Two levels take 8 values down to 2
// registers whose four rows each hold a different value; four row-local DPP adds finish them.
//   q0 rows 0..3 = totals of v[0], v[2], v[1], v[3]     q1 rows 0..3 = totals of v[4], v[6], v[5], v[7]
//   r8 row 3     = total of v[8] (plain six-step chain)
// 30 VALU instructions instead of 72 for nine independent six-step chains.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float swap32_add(float a, float b) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float swap16_add(float a, float b) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float row_sum16(float v) {       // every lane <- sum over its 16-lane row
    v = dpp_add<0xB1, 0xf>(v); v = dpp_add<0x4E, 0xf>(v); v = dpp_add<0x141, 0xf>(v); v = dpp_add<0x140, 0xf>(v);
    return v;
}
struct Reduced9 { float q0, q1, r8; };
__device__ __forceinline__ Reduced9 wave_reduce9(const float (&v)[9]) {
    Reduced9 r;
    const float p0 = swap32_add(v[0], v[1]), p1 = swap32_add(v[2], v[3]);
    const float p2 = swap32_add(v[4], v[5]), p3 = swap32_add(v[6], v[7]);
    r.q0 = row_sum16(swap16_add(p0, p1));
    r.q1 = row_sum16(swap16_add(p2, p3));
    float t = row_sum16(v[8]);
    t = dpp_add<0x142, 0xa>(t);
    r.r8 = dpp_add<0x143, 0xc>(t);
    return r;
}
// which of the nine totals does this lane publish?  (-1: none).  Lanes 0,16,32,48 -> q0; 1,17,33,49 -> q1; 50 -> r8.
__device__ __forceinline__ int reduce9_value_of_lane(int lane) {
    const int row = lane >> 4, c = lane & 15;
    const int perm = (row == 1) ? 2 : (row == 2) ? 1 : row;           // rows hold values 0,2,1,3
    if (c == 0) return perm;
    if (c == 1) return 4 + perm;
    if (lane == 50) return 8;
    return -1;
}
__device__ __forceinline__ float reduce9_pick(const Reduced9 &r, int lane) {
    const int c = lane & 15;
    return c == 0 ? r.q0 : (c == 1 ? r.q1 : r.r8);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}
__device__ __forceinline__ float bcast(float v, int lane) {   // lane is wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
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
    int b = blockIdx.x;
#ifdef D3GA_DIAG
    {   // diagnostic build: a grid launched k times too large runs every quadrant k times (throughput vs balance test)
        const int n = 8 * ((gy + 7) / 8) * gx * 4;
        b = b % n;
    }
#endif
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

// Conservative test: can the Gaussian reach alpha >= 1/255 on any pixel of the quadrant [x0,x0+7]x[y0,y0+7]?
// alpha = o*exp(-q/2) >= 1/255  <=>  q = A dx^2 + 2B dx dy + C dy^2 <= 2 ln(255 o) =: tau.  The ellipse q <= tau has the
// axis-aligned half extents sqrt(tau*C/det), sqrt(tau*A/det); they are inflated by 0.1 % + 0.02 px against rounding.
// Comparisons are written so that NaNs answer "relevant".
__device__ __forceinline__ bool quad_relevant(float cx, float cy, float A, float B, float C, float o, float x0, float y0) {
    if (o * 255.0f < 1.0f) return false;                  // o*G <= o < 1/255 for every G <= 1
    // hardware rcp / sqrt / log (1 ulp-ish) are fine here: the extents are inflated below
    const float tau = 2.0f * __logf(255.0f * o) * 1.001f + 1e-4f;
    const float idet = __builtin_amdgcn_rcpf(A * C - B * B);
    const float hx = __builtin_amdgcn_sqrtf(tau * C * idet) * 1.001f + 0.02f;
    const float hy = __builtin_amdgcn_sqrtf(tau * A * idet) * 1.001f + 0.02f;
    return !(cx + hx < x0) && !(cx - hx > x0 + 7.0f) && !(cy + hy < y0) && !(cy - hy > y0 + 7.0f);
}

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

__device__ __forceinline__ void splat_eval(float dx, float dy, float ca, float cb, float cc, float o, float &alpha,
                                           float &G, bool &ok) {
    const ConicQ q = conic_q(ca, cb, cc);
    splat_eval_q(dx, dy, q, o, alpha, G, ok);
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
// half extents of the alpha >= 1/255 ellipse (inflated); negative hx marks "never visible"
__device__ __forceinline__ void splat_extent(float A, float B, float C, float o, float &hx, float &hy) {
    if (o * 255.0f < 1.0f) { hx = -1.0f; hy = -1.0f; return; }
    const float tau = 2.0f * __logf(255.0f * o) * 1.001f + 1e-4f;
    const float idet = __builtin_amdgcn_rcpf(A * C - B * B);
    hx = __builtin_amdgcn_sqrtf(tau * C * idet) * 1.001f + 0.02f;
    hy = __builtin_amdgcn_sqrtf(tau * A * idet) * 1.001f + 0.02f;
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
// Exact refinement of block_hits4 (variant bit 7).  The bounding box of the alpha >= 1/255 ellipse q(d) <= tau is loose for
// elongated, rotated splats; every (entry, block) pair that survives costs the backward 16 pixel steps and one atomic
// request.  A pair is kept iff  min over the block's rectangle of q  <= tau.  q = A dx^2 + 2B dx dy + C dy^2 is convex, so
// with (fx, fy) = the rectangle's point closest to the centre per axis (clamp of 0 into [xa,xb] / [ya,yb], all relative to
// the centre) the minimum is  min( min_dy q(fx, dy), min_dx q(dx, fy) )  with the inner minimiser clamped to the edge:
// centre inside the x range -> fx = 0 and the second term is never larger; inside both -> 0.  tau as in splat_extent
// (inflated 0.1 % + 1e-4); pixels are a subset of the continuous rectangle, so the test stays conservative.
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
    auto qx = [&](float fx, float ya, float yb) { const float dy = clamp3(c.nbc * fx, ya, yb); return A * fx * fx + dy * (tB * fx + C * dy); };
    auto qy = [&](float fy, float xa, float xb) { const float dx = clamp3(c.nba * fy, xa, xb); return C * fy * fy + dx * (tB * fy + A * dx); };
    BlockHits h;
    // (!(q > tau): NaNs answer "relevant", like the box test)
    h.r0 = box.r0 && !(fminf(qx(fx0, ya0, yb0), qy(fy0, xa0, xb0)) > c.tau);
    h.r1 = box.r1 && !(fminf(qx(fx1, ya0, yb0), qy(fy0, xa1, xb1)) > c.tau);
    h.r2 = box.r2 && !(fminf(qx(fx0, ya1, yb1), qy(fy1, xa0, xb0)) > c.tau);
    h.r3 = box.r3 && !(fminf(qx(fx1, ya1, yb1), qy(fy1, xa1, xb1)) > c.tau);
    return h;
}
__device__ __forceinline__ bool block_hit(float cx, float cy, float hx, float hy, float x0, float y0, float ext) {
    return !(hx < 0.0f) && !(cx + hx < x0) && !(cx - hx > x0 + ext) && !(cy + hy < y0) && !(cy - hy > y0 + ext);
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

// builds the four per-row lists of one staged batch; returns the per-lane count of THIS lane's row and the trip count
__device__ __forceinline__ int build_row_lists(uint8_t (*s_list)[64], bool r0, bool r1, bool r2, bool r3, int lane, int row,
                                               int &trip) {
    const unsigned long long m0 = __ballot(r0), m1 = __ballot(r1), m2 = __ballot(r2), m3 = __ballot(r3);
    if (r0) s_list[0][lanes_below(m0)] = (uint8_t)lane;
    if (r1) s_list[1][lanes_below(m1)] = (uint8_t)lane;
    if (r2) s_list[2][lanes_below(m2)] = (uint8_t)lane;
    if (r3) s_list[3][lanes_below(m3)] = (uint8_t)lane;
    const int c0 = __popcll(m0), c1 = __popcll(m1), c2 = __popcll(m2), c3 = __popcll(m3);
    trip = max(max(c0, c1), max(c2, c3));
    return row == 0 ? c0 : (row == 1 ? c1 : (row == 2 ? c2 : c3));
}

}  // namespace d3ga

// raster_composite.hip -- alpha compositing forward / backward for gfx950.
//
// Wavefront-autonomous design.  The unit of work is ONE 64-lane wavefront = one 8x8-pixel quadrant of a 16x16 tile
// (workgroup = one wavefront: no barriers, independent early-out, 4x finer load balancing than a workgroup per
// tile).  A wavefront walks its tile's depth-ordered list 64 entries at a time:
//   1. every lane gathers ONE entry's record (xy, conic+opacity, rgb+1/depth: three 8/16-byte loads), parks it in a
//      wave-private LDS slab and tests it with a conservative bounding box of the alpha >= 1/255 ellipse;
//   2. DEFAULT ("rows" kernels): each of the four 16-lane DPP rows owns a 4x4 sub-block; the batch is tested against
//      the four sub-blocks (four ballots) and compacted into four per-row index lists in LDS; iteration i makes row r
//      process the i-th entry of its own list (records fetched with per-row broadcast ds_reads).
//      ALTERNATIVE (64-lane kernels, D3GA_COMPOSITE_VARIANT): one ballot per batch, all 64 lanes visit the set bits;
//      records come from the LDS slab or from v_readlane broadcasts;
//   3. forward: straight-line front-to-back blend, two list positions per iteration, stop when all 64 pixels are
//      saturated (T < 1e-4);
//      backward: back-to-front from the deepest contributor; the nine per-pixel partial derivatives are reduced inside
//      the row with four DPP steps (VALU only) and lanes 0..8 of every hit row issue ONE global_atomic_add_f32
//      instruction per iteration (64-lane variant: v_permlane32/16_swap reduce-scatter, 9 lanes publish).
// Culled entries provably contribute nothing (alpha < 1/255 on every pixel of the block), so the result is identical
// to walking the full tile list; list positions (n_contrib) are kept as positions in the FULL list.
//
// Work -> XCD mapping: the dispatcher places block b on XCD b % 8 (observed, speed only).  Tile ROW r is processed
// by XCD r % 8, so the four quadrants of a tile and its horizontal neighbours -- which share most of their
// Gaussians -- gather their records through the same 4 MiB L2, while the image's heavy rows stay interleaved
// across XCDs for balance.
#include <stdlib.h>

#include "composite_common.h"

namespace d3ga {

template <bool LDS>
__global__ __launch_bounds__(64) void composite_fwd_kernel(
    int W, int H, int gx, int gy, const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
    uint64_t dcap, const float2 *__restrict__ xy, const float4 *__restrict__ conic_o,
    const float4 *__restrict__ rgb_invd, const float *__restrict__ bg, float *__restrict__ final_T,
    uint32_t *__restrict__ n_contrib, float *__restrict__ out_color, float *__restrict__ out_invdepth) {
    const Quad q = quad_of_block(gx, gy);
    if (!q.valid || q.qx0 >= W || q.qy0 >= H) return;     // wave-uniform
    const int lane = threadIdx.x & 63;
    const bool inside = q.px < W && q.py < H;
    const float fx = (float)q.px, fy = (float)q.py, x0 = (float)q.qx0, y0 = (float)q.qy0;
    const uint32_t begin = (uint32_t)min((uint64_t)tile_start[q.tile], dcap);
    const uint32_t end = (uint32_t)min((uint64_t)tile_start[q.tile + 1], dcap);

    // LDS variant: the batch's records are parked in a wave-private LDS slab and each entry is fetched with three
    // uniform-address (broadcast) ds_reads instead of ten v_readlane (which occupy the VALU).
    __shared__ float2 s_xy[64];
    __shared__ float4 s_co[64];
    __shared__ float4 s_rgb[64];

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    // software pipeline: the next batch's records are in flight while the current batch is blended
    float2 nxy = make_float2(0.f, 0.f);
    float4 nco = make_float4(0.f, 0.f, 0.f, 0.f), nrgb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (begin + lane < end) {
        const uint32_t g = point_list[begin + lane];
        nxy = xy[g]; nco = conic_o[g]; nrgb = rgb_invd[g];
    }
    for (uint32_t base = begin; base < end; base += 64) {
        const float2 cxy = nxy;
        const float4 cco = nco, crgb = nrgb;
        const bool have = base + lane < end;
        const uint32_t nb = base + 64;
        if (nb + lane < end) {
            const uint32_t g = point_list[nb + lane];
            nxy = xy[g]; nco = conic_o[g]; nrgb = rgb_invd[g];
        }
        unsigned long long mask = __ballot(have && quad_relevant(cxy.x, cxy.y, cco.x, cco.y, cco.z, cco.w, x0, y0));
        if constexpr (LDS) {
            __builtin_amdgcn_wave_barrier();              // previous batch's reads are done (program order)
            s_xy[lane] = cxy; s_co[lane] = cco; s_rgb[lane] = crgb;
            __builtin_amdgcn_wave_barrier();
        }
        // Two entries per iteration, straight-line code: the alpha evaluations (readlane broadcasts, quadratic form,
        // exp) of the pair are independent and overlap; only the short T / done recurrence is serial.  No divergent
        // branches -> no exec-mask juggling on the (single, shared) scalar unit.
        while (mask) {
            const int j0 = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const bool two = mask != 0;
            const int j1 = two ? __ffsll((long long)mask) - 1 : j0;
            mask &= mask - 1;                                              // no-op when mask is already 0
            float al0, G0, al1, G1;
            bool ok0, ok1;
            float2 e0xy, e1xy;
            float4 e0co, e1co, e0rgb, e1rgb;
            if constexpr (LDS) {
                e0xy = s_xy[j0]; e0co = s_co[j0]; e0rgb = s_rgb[j0];
                e1xy = s_xy[j1]; e1co = s_co[j1]; e1rgb = s_rgb[j1];
            } else {
                e0xy = make_float2(bcast(cxy.x, j0), bcast(cxy.y, j0));
                e0co = make_float4(bcast(cco.x, j0), bcast(cco.y, j0), bcast(cco.z, j0), bcast(cco.w, j0));
                e0rgb = make_float4(bcast(crgb.x, j0), bcast(crgb.y, j0), bcast(crgb.z, j0), bcast(crgb.w, j0));
                e1xy = make_float2(bcast(cxy.x, j1), bcast(cxy.y, j1));
                e1co = make_float4(bcast(cco.x, j1), bcast(cco.y, j1), bcast(cco.z, j1), bcast(cco.w, j1));
                e1rgb = make_float4(bcast(crgb.x, j1), bcast(crgb.y, j1), bcast(crgb.z, j1), bcast(crgb.w, j1));
            }
            splat_eval(e0xy.x - fx, e0xy.y - fy, e0co.x, e0co.y, e0co.z, e0co.w, al0, G0, ok0);
            splat_eval(e1xy.x - fx, e1xy.y - fy, e1co.x, e1co.y, e1co.z, e1co.w, al1, G1, ok1);
            ok1 = ok1 && two;
            {   // entry j0
                const bool act = ok0 && !done;
                const float test_T = T * (1.0f - al0);
                const bool sat = act && (test_T < kTmin);
                const bool bl = act && !sat;
                const float w = bl ? al0 * T : 0.f;
                C0 += e0rgb.x * w; C1 += e0rgb.y * w; C2 += e0rgb.z * w; Dp += e0rgb.w * w;
                T = bl ? test_T : T;
                last = bl ? (base - begin + (uint32_t)j0 + 1u) : last;   // 1-based position in the FULL tile list
                done = done || sat;
            }
            {   // entry j1
                const bool act = ok1 && !done;
                const float test_T = T * (1.0f - al1);
                const bool sat = act && (test_T < kTmin);
                const bool bl = act && !sat;
                const float w = bl ? al1 * T : 0.f;
                C0 += e1rgb.x * w; C1 += e1rgb.y * w; C2 += e1rgb.z * w; Dp += e1rgb.w * w;
                T = bl ? test_T : T;
                last = bl ? (base - begin + (uint32_t)j1 + 1u) : last;
                done = done || sat;
            }
            if (__all(done)) { mask = 0; base = end; }                     // whole quadrant saturated
        }
    }
    if (inside) {
        const size_t pid = (size_t)q.py * W + q.px;
        const size_t hw = (size_t)H * W;
        final_T[pid] = T;
        n_contrib[pid] = last;
        out_color[pid] = C0 + T * bg[0];
        out_color[hw + pid] = C1 + T * bg[1];
        out_color[2 * hw + pid] = C2 + T * bg[2];
        if (out_invdepth) out_invdepth[pid] = Dp;
    }
}

constexpr int kNG = 9;   // partial derivatives per (pixel, Gaussian): mean2D x,y | conic a,b/2,c | opacity | r,g,b

template <bool LDS>
__global__ __launch_bounds__(64) void composite_bwd_kernel(
    int W, int H, int gx, int gy, const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
    uint64_t dcap, const float2 *__restrict__ xy, const float4 *__restrict__ conic_o,
    const float4 *__restrict__ rgb_invd, const float *__restrict__ bg, const float *__restrict__ final_T,
    const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix, float *__restrict__ acc) {
    const Quad q = quad_of_block(gx, gy);
    if (!q.valid || q.qx0 >= W || q.qy0 >= H) return;     // wave-uniform
    const int lane = threadIdx.x & 63;
    const bool inside = q.px < W && q.py < H;
    const float fx = (float)q.px, fy = (float)q.py, x0 = (float)q.qx0, y0 = (float)q.qy0;
    const uint32_t begin = (uint32_t)min((uint64_t)tile_start[q.tile], dcap);
    const uint32_t end = (uint32_t)min((uint64_t)tile_start[q.tile + 1], dcap);
    if (begin >= end) return;                              // uniform: empty tile

    const size_t pid = (size_t)q.py * W + q.px;
    const size_t hw = (size_t)H * W;
    const float T_final = inside ? final_T[pid] : 0.f;
    const uint32_t last = inside ? n_contrib[pid] : 0u;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (inside) { g0 = dL_dpix[pid]; g1 = dL_dpix[hw + pid]; g2 = dL_dpix[2 * hw + pid]; }
    const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    const uint32_t maxlast = wave_max_u32(last);           // deepest 1-based list position any pixel of the quadrant used
    if (maxlast == 0) return;

    __shared__ float2 s_xy[64];
    __shared__ float4 s_co[64];
    __shared__ float4 s_rgb[64];
    __shared__ uint32_t s_id[64];

    float T = T_final;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;                    // colour accumulated behind the current splat
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    const int slot = reduce9_value_of_lane(lane);          // which of the nine totals this lane publishes (-1: none)
    const int slot_off = slot < 2 ? slot : slot + 1;       // acc layout: 0,1 | 3,4,5 | 6 | 7,8,9

    // positions hi, hi-1, ... (1-based); lane l holds position hi - l, so ascending lanes = back-to-front
    float2 nxy = make_float2(0.f, 0.f);
    float4 nco = make_float4(0.f, 0.f, 0.f, 0.f), nrgb = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t nid = 0;
    if ((uint32_t)lane < maxlast) {
        nid = point_list[begin + (maxlast - 1 - lane)];
        nxy = xy[nid]; nco = conic_o[nid]; nrgb = rgb_invd[nid];
    }
    for (uint32_t hi = maxlast; hi > 0; hi = hi > 64 ? hi - 64 : 0) {
        const float2 cxy = nxy;
        const float4 cco = nco, crgb = nrgb;
        const uint32_t cid = nid;
        const bool have = (uint32_t)lane < hi;
        if (hi > 64 && (uint32_t)lane < hi - 64) {
            nid = point_list[begin + (hi - 64 - 1 - lane)];
            nxy = xy[nid]; nco = conic_o[nid]; nrgb = rgb_invd[nid];
        }
        unsigned long long mask = __ballot(have && quad_relevant(cxy.x, cxy.y, cco.x, cco.y, cco.z, cco.w, x0, y0));
        if constexpr (LDS) {
            __builtin_amdgcn_wave_barrier();
            s_xy[lane] = cxy; s_co[lane] = cco; s_rgb[lane] = crgb; s_id[lane] = cid;
            __builtin_amdgcn_wave_barrier();
        }
        while (mask) {
            const int j = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const uint32_t pos = hi - (uint32_t)j;
            float ex, ey, ea, eb, ec, eo;
            if constexpr (LDS) {
                const float2 t = s_xy[j]; const float4 u = s_co[j];
                ex = t.x; ey = t.y; ea = u.x; eb = u.y; ec = u.z; eo = u.w;
            } else {
                ex = bcast(cxy.x, j); ey = bcast(cxy.y, j);
                ea = bcast(cco.x, j); eb = bcast(cco.y, j); ec = bcast(cco.z, j); eo = bcast(cco.w, j);
            }
            const float dx = ex - fx, dy = ey - fy;
            float alpha = 0.f, G = 0.f;
            const bool hit = inside && pos <= last && splat_alpha(dx, dy, ea, eb, ec, eo, alpha, G);
            if (!__any(hit)) continue;                     // wave-uniform skip
            float cr, cg, cb;
            uint32_t gid;
            if constexpr (LDS) {
                const float4 t = s_rgb[j];
                cr = t.x; cg = t.y; cb = t.z;
                gid = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_id[j]);
            } else {
                cr = bcast(crgb.x, j); cg = bcast(crgb.y, j); cb = bcast(crgb.z, j);
                gid = (uint32_t)__builtin_amdgcn_readlane((int)cid, j);
            }
            float v[kNG];
#pragma unroll
            for (int k = 0; k < kNG; ++k) v[k] = 0.f;
            if (hit) {
                // hardware reciprocal (1 ulp) instead of an IEEE division: the ~50-step running product stays within
                // ~1e-5 relative of the forward's T, far inside the 1e-3 gradient bar
                const float inv1ma = __builtin_amdgcn_rcpf(1.0f - alpha);
                T = T * inv1ma;
                const float dch = alpha * T;
                a0 = last_alpha * lc0 + (1.f - last_alpha) * a0;
                a1 = last_alpha * lc1 + (1.f - last_alpha) * a1;
                a2 = last_alpha * lc2 + (1.f - last_alpha) * a2;
                lc0 = cr; lc1 = cg; lc2 = cb;
                float dL_dalpha = ((cr - a0) * g0 + (cg - a1) * g1 + (cb - a2) * g2) * T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * inv1ma) * bg_dot;
                const float dL_dG = eo * dL_dalpha;        // the 0.99 clamp passes the gradient through
                const float gdx = G * dx, gdy = G * dy;
                v[0] = dL_dG * (-gdx * ea - gdy * eb) * ddelx_dx;
                v[1] = dL_dG * (-gdy * ec - gdx * eb) * ddely_dy;
                v[2] = -0.5f * gdx * dx * dL_dG;
                v[3] = -0.5f * gdx * dy * dL_dG;            // half of dL/dB, doubled in the per-Gaussian backward
                v[4] = -0.5f * gdy * dy * dL_dG;
                v[5] = G * dL_dalpha;
                v[6] = dch * g0; v[7] = dch * g1; v[8] = dch * g2;
            }
            const Reduced9 red = wave_reduce9(v);
            if (slot >= 0) atomicAdd(acc + D3GA_ACC_STRIDE * (size_t)gid + slot_off, reduce9_pick(red, lane));   // one instruction, 9 lanes
        }
    }
}

// =========================================================================================================
// Row-segmented variant.  A wavefront still owns one 8x8 quadrant, but each of its four 16-lane DPP rows owns a 4x4
// sub-block and walks ITS OWN culled list: per batch the 64 staged entries are tested against the four sub-blocks
// (four ballots), compacted into four per-row index lists in LDS, and iteration i makes row r process the i-th
// entry of list r.  A 4x4 block is touched by ~1.6x fewer list entries than an 8x8 quadrant (measured at C3: 160 vs
// 251 iterations per quadrant), and the backward's cross-lane reduction shrinks to the four row-local DPP steps.
// =========================================================================================================

// DUAL: a second set of per-Gaussian colours (colors2, (P,3), read by Gaussian id) is blended with the same alphas into
// out_color2 over bg2 -- the reference's training step renders every package twice with identical geometry and opacities
// (RGB, then the silhouette colours on black: models/trainer.py:102-110); alpha, T, the culled lists and the early exit
// are shared, the second image costs three more FMAs per (pixel, entry).
template <bool DUAL>
__global__ __launch_bounds__(64) void composite_fwd_rows_kernel(
    int W, int H, int gx, int gy, const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
    uint64_t dcap, const float2 *__restrict__ xy, const float4 *__restrict__ conic_o,
    const float4 *__restrict__ rgb_invd, const float *__restrict__ bg, float *__restrict__ final_T,
    uint32_t *__restrict__ n_contrib, float *__restrict__ out_color, float *__restrict__ out_invdepth,
    const uint32_t *__restrict__ tile_order, const float *__restrict__ colors2, const float *__restrict__ bg2,
    float *__restrict__ out_color2, uint2 *__restrict__ blk_list, uint32_t *__restrict__ blk_count, bool exact_cull) {
    const Quad q = tile_order ? quad_of_block_ordered(gx, gx * gy, tile_order) : quad_of_block(gx, gy);
    if (!q.valid || q.qx0 >= W || q.qy0 >= H) return;     // wave-uniform
    const int lane = threadIdx.x & 63;
    const RowGeom rg = row_geom(q, lane);
    const bool inside = rg.px < W && rg.py < H;
    const float fx = (float)rg.px, fy = (float)rg.py;
    const float bx0 = (float)q.qx0, by0 = (float)q.qy0;
    const uint32_t begin = (uint32_t)min((uint64_t)tile_start[q.tile], dcap);
    const uint32_t end = (uint32_t)min((uint64_t)tile_start[q.tile + 1], dcap);
    // culled per-block lists for the backward (ImgBuf): block 4*quad + r of this tile, capacity end - begin each
    const uint32_t blk_cap = end - begin;
    uint2 *const blk_base = blk_list ? blk_list + 16 * (size_t)begin + (size_t)(4 * q.quad) * blk_cap : nullptr;
    uint32_t bc0 = 0, bc1 = 0, bc2 = 0, bc3 = 0;           // entries emitted per row so far (wave-uniform)

    __shared__ float2 s_xy[64];
    __shared__ float4 s_co[64];
    __shared__ float4 s_rgb[64];
    __shared__ float4 s_rgb2[DUAL ? 64 : 1];
    __shared__ uint8_t s_list[4][64];

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    float E0 = 0.f, E1 = 0.f, E2 = 0.f;                    // DUAL: the second image
    uint32_t last = 0;
    bool done = !inside;

    float2 nxy = make_float2(0.f, 0.f);
    float4 nco = make_float4(0.f, 0.f, 0.f, 0.f), nrgb = make_float4(0.f, 0.f, 0.f, 0.f), nrgb2 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t ng = 0;
    if (begin + lane < end) {
        const uint32_t g = point_list[begin + lane];
        ng = g;
        nxy = xy[g]; nco = conic_o[g]; nrgb = rgb_invd[g];
        if constexpr (DUAL) nrgb2 = make_float4(colors2[3 * (size_t)g], colors2[3 * (size_t)g + 1], colors2[3 * (size_t)g + 2], 0.f);
    }
    for (uint32_t base = begin; base < end; base += 64) {
        const float2 cxy = nxy;
        const float4 cco = nco, crgb = nrgb, crgb2 = nrgb2;
        const uint32_t cg = ng;
        const bool have = base + lane < end;
        const uint32_t nb = base + 64;
        if (nb + lane < end) {
            const uint32_t g = point_list[nb + lane];
            ng = g;
            nxy = xy[g]; nco = conic_o[g]; nrgb = rgb_invd[g];
            if constexpr (DUAL) nrgb2 = make_float4(colors2[3 * (size_t)g], colors2[3 * (size_t)g + 1], colors2[3 * (size_t)g + 2], 0.f);
        }
        const SplatCull sc = splat_cull(cco.x, cco.y, cco.z, cco.w);
        const float hx = have ? sc.hx : -1.0f, hy = sc.hy;
        __builtin_amdgcn_wave_barrier();                  // previous batch's LDS reads are done (program order)
        s_xy[lane] = cxy; s_rgb[lane] = crgb;
        {   // the blend reads the conic with its constants folded in (splat_eval_q)
            const ConicQ cq = conic_q(cco.x, cco.y, cco.z);
            s_co[lane] = make_float4(cq.a, cq.b, cq.c, cco.w);
        }
        if constexpr (DUAL) s_rgb2[lane] = crgb2;
        int trip;
        BlockHits bh = block_hits4(cxy.x, cxy.y, hx, hy, bx0, by0);
        if (exact_cull) bh = block_hits4_exact(cxy.x, cxy.y, cco.x, cco.y, cco.z, sc, bx0, by0, bh);
        const int my_cnt = build_row_lists(s_list, bh.r0, bh.r1, bh.r2, bh.r3, lane, rg.row, trip);
        if (blk_base) {
            // rows whose 16 pixels are all saturated (or outside the image) will not use this batch in the backward
            const unsigned long long dm = __ballot(done);
            const uint2 rec = make_uint2(base - begin + (uint32_t)lane + 1u, cg);          // 1-based list position, id
            const unsigned long long m0 = __ballot(bh.r0), m1 = __ballot(bh.r1), m2 = __ballot(bh.r2), m3 = __ballot(bh.r3);
            if ((dm & 0xffffull) != 0xffffull) {
                if (bh.r0) blk_base[bc0 + (uint32_t)lanes_below(m0)] = rec;
                bc0 += (uint32_t)__popcll(m0);
            }
            if (((dm >> 16) & 0xffffull) != 0xffffull) {
                if (bh.r1) blk_base[blk_cap + bc1 + (uint32_t)lanes_below(m1)] = rec;
                bc1 += (uint32_t)__popcll(m1);
            }
            if (((dm >> 32) & 0xffffull) != 0xffffull) {
                if (bh.r2) blk_base[2 * (size_t)blk_cap + bc2 + (uint32_t)lanes_below(m2)] = rec;
                bc2 += (uint32_t)__popcll(m2);
            }
            if ((dm >> 48) != 0xffffull) {
                if (bh.r3) blk_base[3 * (size_t)blk_cap + bc3 + (uint32_t)lanes_below(m3)] = rec;
                bc3 += (uint32_t)__popcll(m3);
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < trip; i += 2) {
            // two list positions per iteration, straight-line; rows whose list is exhausted idle (valid = false)
            const bool v0 = i < my_cnt, v1 = i + 1 < my_cnt;
            // (& 63: slots past a row's count hold stale bytes; they are only ever read with valid == false)
            const int j0 = s_list[rg.row][i] & 63, j1 = s_list[rg.row][(i + 1) & 63] & 63;
            const float2 e0xy = s_xy[j0], e1xy = s_xy[j1];
            const float4 e0co = s_co[j0], e1co = s_co[j1];
            const float4 e0rgb = s_rgb[j0], e1rgb = s_rgb[j1];
            float al0, G0, al1, G1;
            bool ok0, ok1;
            splat_eval_q(e0xy.x - fx, e0xy.y - fy, ConicQ{e0co.x, e0co.y, e0co.z}, e0co.w, al0, G0, ok0);
            splat_eval_q(e1xy.x - fx, e1xy.y - fy, ConicQ{e1co.x, e1co.y, e1co.z}, e1co.w, al1, G1, ok1);
            {
                const bool act = ok0 && v0 && !done;
                const float test_T = T * (1.0f - al0);
                const bool sat = act && (test_T < kTmin);
                const bool bl = act && !sat;
                const float w = bl ? al0 * T : 0.f;
                C0 += e0rgb.x * w; C1 += e0rgb.y * w; C2 += e0rgb.z * w; Dp += e0rgb.w * w;
                if constexpr (DUAL) { const float4 u = s_rgb2[j0]; E0 += u.x * w; E1 += u.y * w; E2 += u.z * w; }
                T = bl ? test_T : T;
                last = bl ? (base - begin + (uint32_t)j0 + 1u) : last;   // 1-based position in the FULL tile list
                done = done || sat;
            }
            {
                const bool act = ok1 && v1 && !done;
                const float test_T = T * (1.0f - al1);
                const bool sat = act && (test_T < kTmin);
                const bool bl = act && !sat;
                const float w = bl ? al1 * T : 0.f;
                C0 += e1rgb.x * w; C1 += e1rgb.y * w; C2 += e1rgb.z * w; Dp += e1rgb.w * w;
                if constexpr (DUAL) { const float4 u = s_rgb2[j1]; E0 += u.x * w; E1 += u.y * w; E2 += u.z * w; }
                T = bl ? test_T : T;
                last = bl ? (base - begin + (uint32_t)j1 + 1u) : last;
                done = done || sat;
            }
            if (__all(done)) { i = trip; base = end; }     // whole quadrant saturated
        }
    }
    if (inside) {
        const size_t pid = (size_t)rg.py * W + rg.px;
        const size_t hw = (size_t)H * W;
        final_T[pid] = T;
        n_contrib[pid] = last;
        out_color[pid] = C0 + T * bg[0];
        out_color[hw + pid] = C1 + T * bg[1];
        out_color[2 * hw + pid] = C2 + T * bg[2];
        if (out_invdepth) out_invdepth[pid] = Dp;
        if constexpr (DUAL) {
            out_color2[pid] = E0 + T * bg2[0];
            out_color2[hw + pid] = E1 + T * bg2[1];
            out_color2[2 * hw + pid] = E2 + T * bg2[2];
        }
    }
    if (blk_count && (lane & 15) == 0) {
        const int r = lane >> 4;
        blk_count[16 * (size_t)q.tile + 4 * q.quad + r] = r == 0 ? bc0 : (r == 1 ? bc1 : (r == 2 ? bc2 : bc3));
    }
}


// ---------------------------------------------------------------------------------------------------------
// Backward, third generation of the row-segmented kernel.  Same work decomposition as the forward above; what
// changed is the instruction stream of the inner loop (the kernel is VALU-issue bound):
//  * RAW MOMENTS: with w = o*G*dL/dalpha the five geometric gradients are linear in  S = sum w*{dx, dy, dx^2, dx*dy,
//    dy^2}; the per-entry constants (conic, -1/2, the NDC scale) are applied ONCE per entry when the batch accumulator
//    is flushed instead of once per (pixel, entry)  -> 10 instead of 22 multiplies in the hit body;
//  * TRANSPOSED REDUCTION: two quad_perm butterflies (18 DPP adds) leave quad sums in all four lanes of a quad; lane t
//    of every quad then keeps values {t, 4+t, 8} only, and two row_ror steps (t-preserving: rotate by 4, by 8) finish
//    them -> 18 + 6 selects + 6 DPP adds + 2 selects = 32 instructions instead of 53, and lane k of the row ends up
//    with total k, exactly where the ds_add_f32 of the batch accumulator wants it;
//  * ONE LDS RECORD per staged entry (48 B: conic+opacity | rgb+id | xy) and an accumulator with the same 48 B stride:
//    the per-row lists hold the BYTE OFFSET j*48 (u16), so one ds_read_u16 yields the address of everything.
// ---------------------------------------------------------------------------------------------------------
constexpr int kRecBytes = 48;
#ifdef D3GA_DIAG
__device__ unsigned long long g_diag[8];     // diagnostic build only (tools/diag_bwd.py): loop statistics of the kernel below
#endif
// DUAL: the gradient of a second image blended with the same alphas from constant colours (colors2 over bg2, see the
// forward) joins dL/dalpha; the nine moments, their reduction and the flush are unchanged (no gradient flows to colors2).
template <bool DUAL>
__global__ __launch_bounds__(64) void composite_bwd_rows3_kernel(
    int W, int H, int gx, int gy, const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
    uint64_t dcap, const float2 *__restrict__ xy, const float4 *__restrict__ conic_o,
    const float4 *__restrict__ rgb_invd, const float *__restrict__ bg, const float *__restrict__ final_T,
    const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix, float *__restrict__ acc,
    const uint32_t *__restrict__ tile_order, const float *__restrict__ colors2, const float *__restrict__ bg2,
    const float *__restrict__ dL_dpix2) {
    const Quad q = tile_order ? quad_of_block_ordered(gx, gx * gy, tile_order) : quad_of_block(gx, gy);
    if (!q.valid || q.qx0 >= W || q.qy0 >= H) return;     // wave-uniform
    const int lane = threadIdx.x & 63;
    const RowGeom rg = row_geom(q, lane);
    const bool inside = rg.px < W && rg.py < H;
    const float fx = (float)rg.px, fy = (float)rg.py;
    const float bx0 = (float)q.qx0, by0 = (float)q.qy0;
    const uint32_t begin = (uint32_t)min((uint64_t)tile_start[q.tile], dcap);
    const uint32_t end = (uint32_t)min((uint64_t)tile_start[q.tile + 1], dcap);
    if (begin >= end) return;                              // uniform: empty tile

    const size_t pid = (size_t)rg.py * W + rg.px;
    const size_t hw = (size_t)H * W;
    const float T_final = inside ? final_T[pid] : 0.f;
    const uint32_t last = inside ? n_contrib[pid] : 0u;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (inside) { g0 = dL_dpix[pid]; g1 = dL_dpix[hw + pid]; g2 = dL_dpix[2 * hw + pid]; }
    float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    float h0 = 0.f, h1 = 0.f, h2 = 0.f;                    // DUAL: dL/dpixel of the second image
    if constexpr (DUAL) {
        if (inside) { h0 = dL_dpix2[pid]; h1 = dL_dpix2[hw + pid]; h2 = dL_dpix2[2 * hw + pid]; }
        bg_dot += bg2[0] * h0 + bg2[1] * h1 + bg2[2] * h2;                   // both backgrounds sit behind the same T
    }
    const uint32_t rowlast = row_max_u32(last);            // deepest position used inside this lane's 4x4 block
    const uint32_t maxlast = wave_max_u32(rowlast);
    if (maxlast == 0) return;
    const uint32_t rl0 = (uint32_t)__builtin_amdgcn_readlane((int)rowlast, 0), rl1 = (uint32_t)__builtin_amdgcn_readlane((int)rowlast, 16);
    const uint32_t rl2 = (uint32_t)__builtin_amdgcn_readlane((int)rowlast, 32), rl3 = (uint32_t)__builtin_amdgcn_readlane((int)rowlast, 48);

    __shared__ __attribute__((aligned(16))) char s_rec[64 * kRecBytes];      // [0,16) conic+o  [16,32) rgb,id  [32,40) xy
    __shared__ __attribute__((aligned(16))) char s_accb[64 * kRecBytes];     // nine float sums per staged entry (+3 pad)
    __shared__ float4 s_rgb2[DUAL ? 64 : 1];                                 // DUAL: the entry's second colour
    __shared__ uint16_t s_list[5][64];                                       // byte offsets j*48; [4] = union of the four rows
    s_list[0][lane] = 0; s_list[1][lane] = 0; s_list[2][lane] = 0; s_list[3][lane] = 0;   // stale slots stay in range

#ifdef D3GA_DIAG_COUNTERS
    unsigned d_batches = 0, d_iter = 0, d_hit = 0, d_lanes = 0, d_rows = 0, d_ent = 0, d_staged = 0;
#endif
    float T = T_final;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, ld0 = 0.f, ld1 = 0.f, ld2 = 0.f;   // DUAL: suffix colour / last colour of image 2
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    const int l16 = lane & 15, t4 = lane & 3;
    const bool t_is0 = t4 == 0, t_is1 = t4 == 1, t_is2 = t4 == 2;
    const bool l_lt4 = l16 < 4, l_lt8 = l16 < 8, l_lt9 = l16 < kNG;
    const uint32_t my_off = (uint32_t)lane * kRecBytes;
    const uint32_t l16x4 = (uint32_t)l16 * 4u;
    const int fq = lane / 9, fk = lane - 9 * fq;           // flush: lane -> (entry within a group of 7, value)
    const int fk_off = fk < 2 ? fk : fk + 1;               // acc layout 0,1 | 3,4,5 | 6 | 7,8,9

    float2 nxy = make_float2(0.f, 0.f);
    float4 nco = make_float4(0.f, 0.f, 0.f, 0.f), nrgb = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 nrgb2 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t nid = 0;
    if ((uint32_t)lane < maxlast) {
        nid = point_list[begin + (maxlast - 1 - lane)];
        nxy = xy[nid]; nco = conic_o[nid]; nrgb = rgb_invd[nid];
        if constexpr (DUAL) nrgb2 = make_float4(colors2[3 * (size_t)nid], colors2[3 * (size_t)nid + 1], colors2[3 * (size_t)nid + 2], 0.f);
    }
    for (uint32_t hi = maxlast; hi > 0; hi = hi > 64 ? hi - 64 : 0) {
        const float2 cxy = nxy;
        const float4 cco = nco;
        float4 crgb = nrgb;
        crgb.w = __uint_as_float(nid);
        const float4 crgb2 = nrgb2;
        const bool have = (uint32_t)lane < hi;
        if (hi > 64 && (uint32_t)lane < hi - 64) {
            nid = point_list[begin + (hi - 64 - 1 - lane)];
            nxy = xy[nid]; nco = conic_o[nid]; nrgb = rgb_invd[nid];
            if constexpr (DUAL) nrgb2 = make_float4(colors2[3 * (size_t)nid], colors2[3 * (size_t)nid + 1], colors2[3 * (size_t)nid + 2], 0.f);
        }
        float hx, hy;
        splat_extent(cco.x, cco.y, cco.z, cco.w, hx, hy);
        if (!have) hx = -1.0f;
        const uint32_t mypos = hi - (uint32_t)lane;        // list position of the entry this lane staged
        // pos = hi - j <= last   <=>   j*48 >= 48*(hi - last): one compare on the list's byte offset
        const uint32_t off_min = hi > last ? (hi - last) * kRecBytes : 0u;
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<float4 *>(s_rec + my_off) = cco;
        *reinterpret_cast<float4 *>(s_rec + my_off + 16) = crgb;
        *reinterpret_cast<float2 *>(s_rec + my_off + 32) = cxy;
        if constexpr (DUAL) s_rgb2[lane] = crgb2;
        {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(s_accb + my_off) = z;
            *reinterpret_cast<float4 *>(s_accb + my_off + 16) = z;
            *reinterpret_cast<float4 *>(s_accb + my_off + 32) = z;
        }
        const BlockHits bh = block_hits4(cxy.x, cxy.y, hx, hy, bx0, by0);
        const bool r0 = mypos <= rl0 && bh.r0, r1 = mypos <= rl1 && bh.r1;
        const bool r2 = mypos <= rl2 && bh.r2, r3 = mypos <= rl3 && bh.r3;
        const unsigned long long m0 = __ballot(r0), m1 = __ballot(r1), m2 = __ballot(r2), m3 = __ballot(r3);
        if (r0) s_list[0][lanes_below(m0)] = (uint16_t)my_off;
        if (r1) s_list[1][lanes_below(m1)] = (uint16_t)my_off;
        if (r2) s_list[2][lanes_below(m2)] = (uint16_t)my_off;
        if (r3) s_list[3][lanes_below(m3)] = (uint16_t)my_off;
        const unsigned long long mu = m0 | m1 | m2 | m3;                    // entries some row will visit: the only
        if (r0 || r1 || r2 || r3) s_list[4][lanes_below(mu)] = (uint16_t)my_off;   // ones the flush has to look at
        const int n_u = __popcll(mu);
        const int c0 = __popcll(m0), c1 = __popcll(m1), c2 = __popcll(m2), c3 = __popcll(m3);
        const int trip = max(max(c0, c1), max(c2, c3));                     // wave-uniform (scalar)
        const int my_cnt = rg.row == 0 ? c0 : (rg.row == 1 ? c1 : (rg.row == 2 ? c2 : c3));
#ifdef D3GA_DIAG_COUNTERS
        d_batches += 1; d_iter += trip; d_ent += c0 + c1 + c2 + c3; d_staged += min(hi, 64u);
#endif
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < trip; ++i) {
            const uint32_t off = s_list[rg.row][i];        // ascending staged lane = back-to-front
            const float4 eco = *reinterpret_cast<const float4 *>(s_rec + off);
            const float2 exy = *reinterpret_cast<const float2 *>(s_rec + off + 32);
            const float dx = exy.x - fx, dy = exy.y - fy;
            float al, G;
            bool ok;
            splat_eval(dx, dy, eco.x, eco.y, eco.z, eco.w, al, G, ok);
            const bool hit = ok && (i < my_cnt) && inside && off >= off_min;
            const unsigned long long hm = __ballot(hit);
            if (hm == 0) continue;                         // wave-uniform skip
#ifdef D3GA_DIAG_COUNTERS
            d_hit += 1; d_lanes += __popcll(hm);
            d_rows += ((hm & 0xffffull) != 0) + ((hm & 0xffff0000ull) != 0) + ((hm & 0xffff00000000ull) != 0) + ((hm >> 48) != 0);
#endif
            const float4 ergb = *reinterpret_cast<const float4 *>(s_rec + off + 16);
            // NOTE: keep this body in the kernel (no helper functions / lambdas over m[]): hipcc then turns the
            // lane-indexed selects below into a scratch-memory table lookup.
            float m[kNG];
#pragma unroll
            for (int k = 0; k < kNG; ++k) m[k] = 0.f;
            if (hit) {
                const float inv1ma = __builtin_amdgcn_rcpf(1.0f - al);
                T = T * inv1ma;
                const float dch = al * T;
                a0 = last_alpha * lc0 + (1.f - last_alpha) * a0;
                a1 = last_alpha * lc1 + (1.f - last_alpha) * a1;
                a2 = last_alpha * lc2 + (1.f - last_alpha) * a2;
                lc0 = ergb.x; lc1 = ergb.y; lc2 = ergb.z;
                float dL_dalpha = ((ergb.x - a0) * g0 + (ergb.y - a1) * g1 + (ergb.z - a2) * g2) * T;
                if constexpr (DUAL) {
                    const float4 u = s_rgb2[off / kRecBytes];
                    b0 = last_alpha * ld0 + (1.f - last_alpha) * b0;
                    b1 = last_alpha * ld1 + (1.f - last_alpha) * b1;
                    b2 = last_alpha * ld2 + (1.f - last_alpha) * b2;
                    ld0 = u.x; ld1 = u.y; ld2 = u.z;
                    dL_dalpha += ((u.x - b0) * h0 + (u.y - b1) * h1 + (u.z - b2) * h2) * T;
                }
                last_alpha = al;
                dL_dalpha += (-T_final * inv1ma) * bg_dot;
                const float gop = G * dL_dalpha;           // dL/dopacity term
                const float w = eco.w * gop;               // o * G * dL/dalpha   (the 0.99 clamp passes the gradient)
                const float wx = w * dx, wy = w * dy;
                m[0] = wx; m[1] = wy; m[2] = wx * dx; m[3] = wx * dy; m[4] = wy * dy;
                m[5] = gop;
                m[6] = dch * g0; m[7] = dch * g1; m[8] = dch * g2;
            }
#pragma unroll
            for (int k = 0; k < kNG; ++k) m[k] = dpp_add<0xB1, 0xf>(m[k]);      // quad_perm [1,0,3,2]
#pragma unroll
            for (int k = 0; k < kNG; ++k) m[k] = dpp_add<0x4E, 0xf>(m[k]);      // quad_perm [2,3,0,1]: quad sums
            float s0 = t_is0 ? m[0] : (t_is1 ? m[1] : (t_is2 ? m[2] : m[3]));
            float s1 = t_is0 ? m[4] : (t_is1 ? m[5] : (t_is2 ? m[6] : m[7]));
            float s2 = m[8];
            s0 = dpp_add<0x124, 0xf>(s0); s1 = dpp_add<0x124, 0xf>(s1); s2 = dpp_add<0x124, 0xf>(s2);   // row_ror:4
            s0 = dpp_add<0x128, 0xf>(s0); s1 = dpp_add<0x128, 0xf>(s1); s2 = dpp_add<0x128, 0xf>(s2);   // row_ror:8
            const float mine = l_lt4 ? s0 : (l_lt8 ? s1 : s2);              // lane k of the row: total of value k
            const bool row_any = ((hm >> (rg.row << 4)) & 0xffffull) != 0;
            if (l_lt9 && row_any) atomicAdd(reinterpret_cast<float *>(s_accb + off + l16x4), mine);   // ds_add_f32
        }
        __builtin_amdgcn_wave_barrier();
        // flush: nine consecutive lanes publish one visited entry (36 contiguous bytes = 2 memory-side requests), seven
        // entries per instruction; the per-entry constants of the raw moments are applied here
#pragma unroll 1
        for (int e0 = 0; e0 < n_u; e0 += 7) {
            const int idx = e0 + fq;
            if (fq < 7 && idx < n_u) {
                const uint32_t eoff = s_list[4][idx];
                const float *sa = reinterpret_cast<const float *>(s_accb + eoff);
                const float S = sa[fk];
                if (S != 0.f || fk < 2) {
                    const float4 co = *reinterpret_cast<const float4 *>(s_rec + eoff);
                    const float O = sa[fk ^ 1];
                    float val = S;
                    if (fk == 0) val = -(co.x * S + co.y * O) * ddelx_dx;
                    else if (fk == 1) val = -(co.z * S + co.y * O) * ddely_dy;
                    else if (fk < 5) val = -0.5f * S;
                    if (val != 0.f) {
                        const uint32_t gid = __float_as_uint(*reinterpret_cast<const float *>(s_rec + eoff + 28));
                        atomicAdd(acc + D3GA_ACC_STRIDE * (size_t)gid + fk_off, val);
                    }
                }
            }
        }
    }
#ifdef D3GA_DIAG_COUNTERS
    if (lane == 0) {
        atomicAdd(&g_diag[0], 1ull); atomicAdd(&g_diag[1], (unsigned long long)d_batches);
        atomicAdd(&g_diag[2], (unsigned long long)d_iter); atomicAdd(&g_diag[3], (unsigned long long)d_hit);
        atomicAdd(&g_diag[4], (unsigned long long)d_lanes); atomicAdd(&g_diag[5], (unsigned long long)d_rows);
        atomicAdd(&g_diag[6], (unsigned long long)d_ent); atomicMax(&g_diag[7], (unsigned long long)d_iter);
        (void)d_staged;
    }
#endif
}

// self-test kernel for the cross-lane reductions (tests/).  Per wave w with inputs x[0..63]:
//   out[10 w + k] = sum_l (k+1) x[l] + k   for k = 0..8, through wave_reduce9 and the lane mapping the backward uses;
//   out[10 w + 9] = sum_l x[l]              through wave_sum / wave_sum_multi (or -1e30 if those two disagree).
__global__ void wave_sum_selftest_kernel(const float *__restrict__ in, float *__restrict__ out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, w = gid >> 6;
    const float x = in[gid];
    float v9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v9[k] = (float)(k + 1) * x + (float)k / 64.0f;
    const Reduced9 red = wave_reduce9(v9);
    const int slot = reduce9_value_of_lane(lane);
    if (slot >= 0) out[10 * w + slot] = reduce9_pick(red, lane);
    float v[2] = {x, 2.0f * x};
    wave_sum_multi<2>(v);
    const float s = wave_sum(x);
    const float m0 = bcast(v[0], 50), m1 = bcast(v[1], 63);
    if (lane == 0) out[10 * w + 9] = (m0 == s && m1 == 2.0f * s) ? s : -1e30f;
}

}  // namespace d3ga

using namespace d3ga;

static int composite_fwd_impl(const d3ga_raster_params *prm, const float *bg, const void *geom, const void *binning,
                              int64_t d_capacity, void *img, float *out_color, float *out_invdepth, const float *colors2,
                              const float *bg2, float *out_color2, d3ga_stream_t stream) {
    if (!prm || !bg || !geom || !binning || !img || !out_color) return D3GA_E_NULL;
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0 || d_capacity < 0) return D3GA_E_SIZE;
    if (colors2 && (!bg2 || !out_color2)) return D3GA_E_NULL;
    if (colors2 && !(composite_variant() & 4)) return D3GA_E_CONFIG;          // only the row-segmented kernels blend two images
    hipStream_t s = (hipStream_t)stream;
    const int gx = tiles_x(prm->W), gy = tiles_y(prm->H);
    const BinBuf bin = carve_bin(const_cast<void *>(binning), (int64_t)gx * gy, d_capacity);
    const GeomBuf g = carve_geom(const_cast<void *>(geom), prm->P);
    const ImgBuf im = carve_img(img, prm->W, prm->H, (int64_t)gx * gy);
    const bool emit = (composite_variant() & 64) != 0;      // the entry-per-lane backward consumes the per-block lists
    if (composite_variant() & 4) {
        const bool ordered = (composite_variant() & 32) != 0;
        if (colors2)
            hipLaunchKernelGGL(composite_fwd_rows_kernel<true>, dim3(ordered ? quad_grid_ordered(gx * gy) : quad_grid(gx, gy)),
                               dim3(64), 0, s, prm->W, prm->H, gx, gy, bin.tile_start, bin.point_list, (uint64_t)d_capacity,
                               g.xy, g.conic_o, g.rgb_invd, bg, im.final_T, im.n_contrib, out_color, out_invdepth,
                               ordered ? (const uint32_t *)bin.tile_order : (const uint32_t *)nullptr, colors2, bg2, out_color2,
                               emit ? im.blk_list : (uint2 *)nullptr, emit ? im.blk_count : (uint32_t *)nullptr, (composite_variant() & 128) != 0);
        else
            hipLaunchKernelGGL(composite_fwd_rows_kernel<false>, dim3(ordered ? quad_grid_ordered(gx * gy) : quad_grid(gx, gy)),
                               dim3(64), 0, s, prm->W, prm->H, gx, gy, bin.tile_start, bin.point_list, (uint64_t)d_capacity,
                               g.xy, g.conic_o, g.rgb_invd, bg, im.final_T, im.n_contrib, out_color, out_invdepth,
                               ordered ? (const uint32_t *)bin.tile_order : (const uint32_t *)nullptr,
                               (const float *)nullptr, (const float *)nullptr, (float *)nullptr,
                               emit ? im.blk_list : (uint2 *)nullptr, emit ? im.blk_count : (uint32_t *)nullptr, (composite_variant() & 128) != 0);
    }
    if (composite_variant() & 4) return check_launch(s, prm->debug);
    if (composite_variant() & 1)
        hipLaunchKernelGGL(composite_fwd_kernel<true>, dim3(quad_grid(gx, gy)), dim3(64), 0, s, prm->W, prm->H, gx, gy,
                           bin.tile_start, bin.point_list, (uint64_t)d_capacity, g.xy, g.conic_o, g.rgb_invd, bg,
                           im.final_T, im.n_contrib, out_color, out_invdepth);
    else
        hipLaunchKernelGGL(composite_fwd_kernel<false>, dim3(quad_grid(gx, gy)), dim3(64), 0, s, prm->W, prm->H, gx, gy,
                           bin.tile_start, bin.point_list, (uint64_t)d_capacity, g.xy, g.conic_o, g.rgb_invd, bg,
                           im.final_T, im.n_contrib, out_color, out_invdepth);
    return check_launch(s, prm->debug);
}

extern "C" int d3ga_raster_composite_fwd(const d3ga_raster_params *prm, const float *bg, const void *geom,
                                         const void *binning, int64_t d_capacity, void *img, float *out_color,
                                         float *out_invdepth, d3ga_stream_t stream) {
    return composite_fwd_impl(prm, bg, geom, binning, d_capacity, img, out_color, out_invdepth, nullptr, nullptr, nullptr, stream);
}

extern "C" int d3ga_raster_composite_fwd2(const d3ga_raster_params *prm, const float *bg, const float *bg2, const void *geom,
                                          const float *colors2, const void *binning, int64_t d_capacity, void *img,
                                          float *out_color, float *out_color2, float *out_invdepth, d3ga_stream_t stream) {
    if (!colors2) return D3GA_E_NULL;
    return composite_fwd_impl(prm, bg, geom, binning, d_capacity, img, out_color, out_invdepth, colors2, bg2, out_color2, stream);
}

#ifdef D3GA_DIAG
extern "C" int d3ga_diag_read(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_diag), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

static int composite_bwd_impl(const d3ga_raster_params *prm, const float *bg, const void *geom, const void *binning,
                              int64_t d_capacity, const void *img, const float *dL_dpix, float *acc, const float *colors2,
                              const float *bg2, const float *dL_dpix2, d3ga_stream_t stream) {
    if (!prm) return D3GA_E_NULL;
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0 || d_capacity < 0) return D3GA_E_SIZE;
    if (prm->P == 0) return D3GA_OK;
    if (!bg || !geom || !binning || !img || !dL_dpix || !acc) return D3GA_E_NULL;
    if (colors2 && (!bg2 || !dL_dpix2)) return D3GA_E_NULL;
    if (colors2 && !(composite_variant() & 8)) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const int gx = tiles_x(prm->W), gy = tiles_y(prm->H);
    const BinBuf bin = carve_bin(const_cast<void *>(binning), (int64_t)gx * gy, d_capacity);
    const GeomBuf g = carve_geom(const_cast<void *>(geom), prm->P);
    const ImgBuf im = carve_img(const_cast<void *>(img), prm->W, prm->H, (int64_t)gx * gy);
#ifdef D3GA_DIAG
    if (composite_variant() & 512) {
        hipLaunchKernelGGL(composite_bwd_rows3_kernel<false>, dim3(2 * quad_grid(gx, gy)), dim3(64), 0, s, prm->W, prm->H, gx,
                           gy, bin.tile_start, bin.point_list, (uint64_t)d_capacity, g.xy, g.conic_o, g.rgb_invd, bg,
                           im.final_T, im.n_contrib, dL_dpix, acc, (const uint32_t *)nullptr, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr);
        return check_launch(s, prm->debug);
    }
#endif
    if ((composite_variant() & 64) && (composite_variant() & 4))       // entry-per-lane backward over the forward's block lists
        return launch_composite_bwd_scan(prm, gx, gy, bin, g, im, d_capacity, bg, dL_dpix, acc, (composite_variant() & 32) != 0,
                                         colors2, bg2, dL_dpix2, s);
    if (composite_variant() & 8) {
        const bool ordered = (composite_variant() & 32) != 0;
        if (colors2)
            hipLaunchKernelGGL(composite_bwd_rows3_kernel<true>, dim3(ordered ? quad_grid_ordered(gx * gy) : quad_grid(gx, gy)),
                               dim3(64), 0, s, prm->W, prm->H, gx, gy, bin.tile_start, bin.point_list, (uint64_t)d_capacity,
                               g.xy, g.conic_o, g.rgb_invd, bg, im.final_T, im.n_contrib, dL_dpix, acc,
                               ordered ? (const uint32_t *)bin.tile_order : (const uint32_t *)nullptr, colors2, bg2, dL_dpix2);
        else
            hipLaunchKernelGGL(composite_bwd_rows3_kernel<false>, dim3(ordered ? quad_grid_ordered(gx * gy) : quad_grid(gx, gy)),
                               dim3(64), 0, s, prm->W, prm->H, gx, gy, bin.tile_start, bin.point_list, (uint64_t)d_capacity,
                               g.xy, g.conic_o, g.rgb_invd, bg, im.final_T, im.n_contrib, dL_dpix, acc,
                               ordered ? (const uint32_t *)bin.tile_order : (const uint32_t *)nullptr, (const float *)nullptr,
                               (const float *)nullptr, (const float *)nullptr);
    }
    else if (composite_variant() & 2)
        hipLaunchKernelGGL(composite_bwd_kernel<true>, dim3(quad_grid(gx, gy)), dim3(64), 0, s, prm->W, prm->H, gx, gy,
                           bin.tile_start, bin.point_list, (uint64_t)d_capacity, g.xy, g.conic_o, g.rgb_invd, bg,
                           im.final_T, im.n_contrib, dL_dpix, acc);
    else
        hipLaunchKernelGGL(composite_bwd_kernel<false>, dim3(quad_grid(gx, gy)), dim3(64), 0, s, prm->W, prm->H, gx, gy,
                           bin.tile_start, bin.point_list, (uint64_t)d_capacity, g.xy, g.conic_o, g.rgb_invd, bg,
                           im.final_T, im.n_contrib, dL_dpix, acc);
    return check_launch(s, prm->debug);
}

extern "C" int d3ga_raster_composite_bwd(const d3ga_raster_params *prm, const float *bg, const void *geom,
                                         const void *binning, int64_t d_capacity, const void *img,
                                         const float *dL_dpix, float *acc, d3ga_stream_t stream) {
    return composite_bwd_impl(prm, bg, geom, binning, d_capacity, img, dL_dpix, acc, nullptr, nullptr, nullptr, stream);
}

extern "C" int d3ga_raster_composite_bwd2(const d3ga_raster_params *prm, const float *bg, const float *bg2, const void *geom,
                                          const float *colors2, const void *binning, int64_t d_capacity, const void *img,
                                          const float *dL_dpix, const float *dL_dpix2, float *acc, d3ga_stream_t stream) {
    if (!colors2) return D3GA_E_NULL;
    return composite_bwd_impl(prm, bg, geom, binning, d_capacity, img, dL_dpix, acc, colors2, bg2, dL_dpix2, stream);
}

// test hook (not part of the drop-in surface): n multiple of 256, in (n) -> out (10 * n/64)
extern "C" int d3ga_selftest_wave_sum(int n, const float *in, float *out, d3ga_stream_t stream) {
    if (n <= 0 || (n % 256) != 0) return D3GA_E_SIZE;
    if (!in || !out) return D3GA_E_NULL;
    hipLaunchKernelGGL(wave_sum_selftest_kernel, dim3(n / 256), dim3(256), 0, (hipStream_t)stream, in, out);
    return check_launch((hipStream_t)stream, 1);
}

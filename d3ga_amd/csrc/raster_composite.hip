// raster_composite.hip -- alpha compositing forward for gfx950 (composite_fwd_q_kernel) and the C entry points of both
// directions (the backward kernel is raster_composite_scan.hip).
//
// Unit of work: ONE 64-lane wavefront = one 8x8-pixel quadrant of a 16x16 tile (workgroup = one wavefront: no barriers,
// independent early-out); each of its four 16-lane DPP rows owns a 4x4-pixel block.  Two stages:
//   1. the tile's depth-ordered list is walked 128 entries at a time with ONE 16-byte gather per entry (GeomBuf::xyh: centre |
//      half extents of the alpha >= 1/255 box, written by preprocess with the same splat_cull()) and a box test against the
//      quadrant; the survivors' (1-based list position, Gaussian id) go into a 256-entry ring in LDS;
//   2. batches of 64 SURVIVORS: every lane gathers one entry's records (xy, conic + opacity, colour + 1/depth; the next
//      batch and the next stage-one chunk are in flight while the current batch is blended), parks them in a wave-private
//      LDS slab, tests the entry against the four blocks (box, then the exact ellipse / rectangle test), the hits are
//      compacted into four per-row lists of slab offsets (four ballots) AND appended to the blocks' culled lists in the
//      ImgBuf -- the backward walks exactly these; then iteration i makes row r blend the i-th and (i+1)-th entry of ITS
//      list, straight-line and front to back (records through per-row broadcast ds_reads, offsets read one iteration ahead);
//      the wavefront stops when all 64 pixels are saturated (T < 1e-4).
// Culled entries provably contribute nothing (alpha < 1/255 on every pixel of the block), so the result is identical to
// walking the full tile list; list positions (n_contrib) stay positions in the FULL list.
// Dispatch: quad_of_block_ordered() hands the quadrants out heaviest tile first (tile_order of the bin stage), the four
// quadrants of a tile on one XCD (workgroup b lands on XCD b % 8: observed, speed only).
// Numbers, history and the negative results (persistent workers, dispatch orders, occupancy): DESIGN.md sec. 4.
#include "composite_common.h"

// wavefronts per SIMD the register budget of the two-stage forward targets (A/B: build.py D3GA_SCAN_ABL=0f6)
#ifndef D3GA_FWD_WAVES
#define D3GA_FWD_WAVES 5
#endif
// (round 5) the register budget of the DUAL instantiations (render_pair) separately: at 5 wavefronts per SIMD they spill
// (7-10 VGPRs, 32-44 bytes of scratch per lane)
#ifndef D3GA_FWD_DUAL_WAVES
#define D3GA_FWD_DUAL_WAVES 5
#endif
// A/B (build.py D3GA_VARIANT): 0 = blk_count is the number of entries emitted (rounds 2-4), 1 = the prefix up to the last blended entry
#ifndef D3GA_FWD_USED
#define D3GA_FWD_USED 1
#endif

namespace d3ga {

#ifdef D3GA_DIAG
// diagnostic build only (tools/diag_fwd.py): [0] active waves, [1] stage-one chunks, [2] stage-two batches, [3] blend-loop
// iterations (two list positions each), [4] sum of the row lists' lengths = (entry, block) pairs, [5] blended (entry, pixel) pairs,
// [6] survivors of stage one, [7] tile-list entries looked at
__device__ unsigned long long g_diag_fwd[8];
__device__ unsigned long long g_diag_fwd_waves[32768 * 4];    // per active wave: start, end (s_memrealtime), iterations | batches << 32, list length
#endif

// DUAL: a second set of per-Gaussian colours (colors2, (P,3), read by Gaussian id) is blended with the same alphas into
// out_color2 over bg2 -- the reference's training step renders every package twice with identical geometry and opacities
// (RGB, then the silhouette colours on black: models/trainer.py:102-110); alpha, T, the culled lists and the early exit
// are shared, the second image costs three more FMAs per (pixel, entry).
// DEPTH: the inverse-depth image of branch dr_aa is accumulated and written (the D3GA renderer uses the colour only:
// renderer.py:141 takes [0]; without it the blend loop is one FMA per entry shorter and 4 B per pixel are not written)
// L1V: the L1 loss value against a target image is formed here as well (d3ga_raster_composite_fwd_l1; never with DUAL)
template <bool DUAL, bool DEPTH, bool L1V>
__global__ __launch_bounds__(64, (DUAL ? D3GA_FWD_DUAL_WAVES : D3GA_FWD_WAVES)) void composite_fwd_q_kernel(
    int W, int H, int gx, int gy, int gyv /* tile rows per view (= gy for one view) */, const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
    uint64_t dcap, const float2 *xy /* = xyh viewed as float2: the centre is record[0..1], stride 2 (round 4: no separate xy array) */, const float4 *__restrict__ conic_o,
    const float4 *__restrict__ rgb_invd, const float4 *xyh, const float *__restrict__ bg,
    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, float *__restrict__ out_color,
    float *__restrict__ out_invdepth, const uint32_t *__restrict__ tile_order, const float *__restrict__ colors2,
    const float *__restrict__ bg2, float *__restrict__ out_color2, uint2 *__restrict__ blk_list,
    uint32_t *__restrict__ blk_count, bool exact_cull, L1Value l1v, int P /* Gaussians per view: colors2 is indexed by Gaussian, the lists by (view, Gaussian) */) {
    const Quad q = tile_order ? quad_of_block_ordered(gx, gx * gy, gyv, tile_order) : quad_of_block(gx, gy, gyv);
    if (!q.valid) return;                                 // wave-uniform
    if (q.qx0 >= W || q.qy0 >= H) {                       // a quadrant without pixels: its L1 partial is zero
        if (L1V && threadIdx.x == 0) l1v.partials[4 * (size_t)q.tile + q.quad] = 0.f;
        return;
    }
    const int lane = threadIdx.x & 63;
    const RowGeom rg = row_geom(q, lane);
    const bool inside = rg.px < W && rg.py < H;
    const float fx = (float)rg.px, fy = (float)rg.py;
    const float bx0 = (float)q.qx0, by0 = (float)q.qy0;
    const uint32_t begin = (uint32_t)min((uint64_t)tile_start[q.tile], dcap);
    const uint32_t end = (uint32_t)min((uint64_t)tile_start[q.tile + 1], dcap);
    const uint32_t blk_cap = end - begin;
    // fused L1 value: the target's pixel is requested NOW and used after the blend loop.  (Measured at C3: the launch grows
    // from 73.8 to 80.6 us, the separate pass it replaces took 10.3 us.  Letting quadrant 0 of an EMPTY tile -- 27 000 of the
    // 32 640 quadrants of an avatar frame -- take the whole tile with 16-byte loads: 86.7 us, slower; loads at the end of the
    // wavefront instead of here: the same.)
    float tg0 = 0.f, tg1 = 0.f, tg2 = 0.f;
    if (L1V && inside) {
        const float *tg = l1v.target_cell ? *l1v.target_cell : l1v.target;
        const size_t hw = (size_t)H * W, pid = 3 * hw * q.view + (size_t)rg.py * W + rg.px;
        tg0 = tg[pid]; tg1 = tg[hw + pid]; tg2 = tg[2 * hw + pid];
    }
    uint2 *const blk_base = blk_list ? blk_list + 16 * (size_t)begin + (size_t)(4 * q.quad) * blk_cap : nullptr;
    uint32_t bc0 = 0, bc1 = 0, bc2 = 0, bc3 = 0;
    // round 5: what the backward walks is the PREFIX of a block's list up to the last entry some pixel of the block blended (a
    // row keeps emitting the hits of the batch in which its last pixel saturates, and of every batch to the end of the tile's list
    // when it never saturates): used r = entries of block r's list up to that one.  Formed once per batch from the pixels' `last`
    // (a row maximum), the batch's positions and the row's hit mask -- nothing per blended entry.
    uint32_t used0 = 0, used1 = 0, used2 = 0, used3 = 0;

    __shared__ float4 s_co[65];
    __shared__ float4 s_rgb[65];
    __shared__ float4 s_xyp[65];
    __shared__ float4 s_rgb2[DUAL ? 65 : 1];
    __shared__ uint16_t s_list[4][kListStride];
    constexpr int kRing = 256;                            // survivors of stage one: (1-based list position, Gaussian id)
    __shared__ uint2 s_ring[kRing];
    if (threadIdx.x == 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        s_co[64] = z; s_rgb[64] = z; s_xyp[64] = z;
        if constexpr (DUAL) s_rgb2[64] = z;
    }
    if (threadIdx.x < 8) s_list[threadIdx.x >> 1][64 + (threadIdx.x & 1)] = kNullRec;

#ifdef D3GA_DIAG_COUNTERS
    unsigned long long df_chunks = 0, df_pairs = 0, df_blend = 0, df_surv = 0;
#endif
#ifdef D3GA_DIAG_TIMELINE
    unsigned long long df_batches = 0, df_iters = 0, df_blend_ticks = 0;
    const unsigned long long df_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    float E0 = 0.f, E1 = 0.f, E2 = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    // ---- stage one state: the list is consumed 128 entries at a time, lane l looks at entries sbase + l and sbase + 64 + l ----
    uint32_t sbase = begin;                               // first list entry of the chunk whose records are in flight
    uint32_t qhead = 0, qcount = 0;                       // ring: entries (qhead + i) % kRing, i < qcount (wave-uniform)
    uint32_t g0 = 0, g1 = 0;                              // ids of the chunk in flight
    float4 h0 = make_float4(0.f, 0.f, -1.f, 0.f), h1 = h0; // its xyh records
    // round 5: the ids are fetched TWO chunks ahead (gn0 / gn1), the records one chunk ahead: at the top of a chunk its records'
    // gathers find their ids in registers instead of waiting for them (the gathers used to be issued right behind the id loads
    // they depend on: one exposed L2 round trip per chunk, ~7 chunks per wavefront)
    uint32_t gn0 = 0, gn1 = 0;
    auto chunk_ids = [&](uint32_t base) {                 // 1st level: the ids (coalesced) of the chunk at `base` -> gn
        gn0 = base + lane < end ? point_list[base + lane] : 0u;
        gn1 = base + 64 + lane < end ? point_list[base + 64 + lane] : 0u;
    };
    auto chunk_recs = [&]() { g0 = gn0; g1 = gn1; h0 = xyh[g0]; h1 = xyh[g1]; };       // 2nd level: the gathers of gn's chunk (id 0 is always readable)
    auto quad_hit = [&](const float4 &r) {                // NaN extents answer "relevant", like block_hits4
        return !(r.z < 0.0f) && !(r.x + r.z < bx0) && !(r.x - r.z > bx0 + 7.0f) && !(r.y + r.w < by0) && !(r.y - r.w > by0 + 7.0f);
    };
    auto consume_chunk = [&]() {                          // test the chunk in flight, append its survivors in list order
        const bool v0 = sbase + lane < end, v1 = sbase + 64 + lane < end;
        const bool k0 = v0 && quad_hit(h0), k1 = v1 && quad_hit(h1);
        const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
        const uint32_t n0 = (uint32_t)__popcll(m0);
        if (k0) s_ring[(qhead + qcount + (uint32_t)lanes_below(m0)) % kRing] = make_uint2(sbase - begin + (uint32_t)lane + 1u, g0);
        if (k1) s_ring[(qhead + qcount + n0 + (uint32_t)lanes_below(m1)) % kRing] = make_uint2(sbase - begin + 64u + (uint32_t)lane + 1u, g1);
        qcount += n0 + (uint32_t)__popcll(m1);
#ifdef D3GA_DIAG_COUNTERS
        df_chunks += 1; df_surv += n0 + (uint32_t)__popcll(m1);
#endif
        sbase = min(sbase + 128u, end);
    };
    // ---- stage two state: the batch whose gathers are in flight ----
    uint32_t bn = 0;                                      // its size (0: none)
    uint2 bpg = make_uint2(0u, 0u);
    float2 nxy = make_float2(0.f, 0.f);
    float4 nco = make_float4(0.f, 0.f, 0.f, 0.f), nrgb = nco, nrgb2 = nco;
    auto issue_batch = [&]() {                            // dequeue up to 64 survivors and gather their records
        bn = min(qcount, 64u);
        bpg = make_uint2(0u, 0u);
        if ((uint32_t)lane < bn) {
            bpg = s_ring[(qhead + (uint32_t)lane) % kRing];
            nxy = xy[2 * (size_t)bpg.y]; nco = conic_o[bpg.y]; nrgb = rgb_invd[bpg.y];
            if constexpr (DUAL) {
                const size_t g2 = 3 * ((size_t)bpg.y - (size_t)q.view * (size_t)P);
                nrgb2 = make_float4(colors2[g2], colors2[g2 + 1], colors2[g2 + 2], 0.f);
            }
        }
        qhead = (qhead + bn) % kRing;
        qcount -= bn;
    };
    auto fill = [&]() {                                   // stage one until a full batch is queued or the list is exhausted
        while (qcount < 64u && sbase < end) {
            consume_chunk();                              // (the ring holds < 64 + 128 entries)
            if (sbase < end) { chunk_recs(); chunk_ids(min(sbase + 128u, end)); }      // records of the chunk at sbase (ids arrived a chunk ago), ids of the next
        }
    };

    if (begin < end) { chunk_ids(begin); chunk_recs(); chunk_ids(min(begin + 128u, end)); }
    __builtin_amdgcn_wave_barrier();
    fill();
    __builtin_amdgcn_wave_barrier();
    issue_batch();
    while (bn != 0u) {
        // ---- stage two on the batch in flight ----
        const float2 cxy = nxy;
        const float4 cco = nco, crgb = nrgb, crgb2 = nrgb2;
        const uint2 cpg = bpg;
        const bool have = (uint32_t)lane < bn;
        const SplatCull sc = splat_cull(cco.x, cco.y, cco.z, cco.w);
        const float hx = have ? sc.hx : -1.0f, hy = sc.hy;
        __builtin_amdgcn_wave_barrier();                  // previous batch's LDS reads are done (program order)
        {
            const ConicQ cq = conic_q(cco.x, cco.y, cco.z);
            s_co[lane] = make_float4(cq.a, cq.b, cq.c, have ? cco.w : 0.f);
            s_rgb[lane] = crgb;
            s_xyp[lane] = make_float4(cxy.x, cxy.y, __uint_as_float(cpg.x), 0.f);
            if constexpr (DUAL) s_rgb2[lane] = crgb2;
        }
        BlockHits bh = block_hits4(cxy.x, cxy.y, hx, hy, bx0, by0);
        if (exact_cull) bh = block_hits4_exact(cxy.x, cxy.y, cco.x, cco.y, cco.z, sc, bx0, by0, bh);
        unsigned long long m[4];
        const int trip = build_row_lists(s_list, bh.r0, bh.r1, bh.r2, bh.r3, lane, m);
#ifdef D3GA_DIAG_TIMELINE
        df_batches += 1;
#endif
#ifdef D3GA_DIAG_COUNTERS
        df_pairs += __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
#endif
        const uint32_t bcb0 = bc0, bcb1 = bc1, bcb2 = bc2, bcb3 = bc3;      // the blocks' counts in front of this batch
        bool em0 = false, em1 = false, em2 = false, em3 = false;            // the rows that emit this batch's hits
        if (blk_base) {
            const unsigned long long dm = __builtin_amdgcn_ballot_w64(done);
            const uint2 rec = cpg;                        // 1-based list position, id
            em0 = (dm & 0xffffull) != 0xffffull; em1 = ((dm >> 16) & 0xffffull) != 0xffffull;
            em2 = ((dm >> 32) & 0xffffull) != 0xffffull; em3 = (dm >> 48) != 0xffffull;
            if ((dm & 0xffffull) != 0xffffull) {
                if (bh.r0) blk_base[bc0 + (uint32_t)lanes_below(m[0])] = rec;
                bc0 += (uint32_t)__popcll(m[0]);
            }
            if (((dm >> 16) & 0xffffull) != 0xffffull) {
                if (bh.r1) blk_base[blk_cap + bc1 + (uint32_t)lanes_below(m[1])] = rec;
                bc1 += (uint32_t)__popcll(m[1]);
            }
            if (((dm >> 32) & 0xffffull) != 0xffffull) {
                if (bh.r2) blk_base[2 * (size_t)blk_cap + bc2 + (uint32_t)lanes_below(m[2])] = rec;
                bc2 += (uint32_t)__popcll(m[2]);
            }
            if ((dm >> 48) != 0xffffull) {
                if (bh.r3) blk_base[3 * (size_t)blk_cap + bc3 + (uint32_t)lanes_below(m[3])] = rec;
                bc3 += (uint32_t)__popcll(m[3]);
            }
        }
        // ---- the next batch: stage one from the chunk in flight, then its gathers and the next chunk's loads ----
        fill();
        __builtin_amdgcn_wave_barrier();                  // ring writes before the dequeue reads (program order)
        issue_batch();
        __builtin_amdgcn_wave_barrier();
        // ---- blend ----
        const uint16_t *const my_list = s_list[rg.row];
        bool all_done = false;
#ifdef D3GA_DIAG_TIMELINE
        const unsigned long long df_tb = __builtin_amdgcn_s_memrealtime();
#endif
        uint32_t p0 = my_list[0], p1 = my_list[1];         // list offsets are read one iteration ahead: off the dependent chain
        for (int i = 0; i < trip; i += 2) {
            const uint32_t o0 = p0, o1 = p1;
            p0 = my_list[i + 2]; p1 = my_list[i + 3];      // (i + 3 <= 65: two null records behind the 64 entries)
            const float4 e0xy = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_xyp) + o0);
            const float4 e1xy = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_xyp) + o1);
            const float4 e0co = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_co) + o0);
            const float4 e1co = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_co) + o1);
            const float4 e0rgb = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rgb) + o0);
            const float4 e1rgb = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rgb) + o1);
            float al0, G0, al1, G1;
            bool ok0, ok1;
            splat_eval_q(e0xy.x - fx, e0xy.y - fy, ConicQ{e0co.x, e0co.y, e0co.z}, e0co.w, al0, G0, ok0);
            splat_eval_q(e1xy.x - fx, e1xy.y - fy, ConicQ{e1co.x, e1co.y, e1co.z}, e1co.w, al1, G1, ok1);
            {
                const bool act = ok0 && !done;
                const float test_T = T * (1.0f - al0);
                const bool keep = !(test_T < kTmin);
                const bool bl = act && keep;
                const float w = bl ? al0 * T : 0.f;
                C0 += e0rgb.x * w; C1 += e0rgb.y * w; C2 += e0rgb.z * w;
                if constexpr (DEPTH) Dp += e0rgb.w * w;
                if constexpr (DUAL) {
                    const float4 u = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rgb2) + o0);
                    E0 += u.x * w; E1 += u.y * w; E2 += u.z * w;
                }
                T = bl ? test_T : T;
                last = bl ? __float_as_uint(e0xy.z) : last;
                done = done || (act != bl);
#ifdef D3GA_DIAG_COUNTERS
                df_blend += __popcll(__builtin_amdgcn_ballot_w64(bl));
#endif
            }
            {
                const bool act = ok1 && !done;
                const float test_T = T * (1.0f - al1);
                const bool keep = !(test_T < kTmin);
                const bool bl = act && keep;
                const float w = bl ? al1 * T : 0.f;
                C0 += e1rgb.x * w; C1 += e1rgb.y * w; C2 += e1rgb.z * w;
                if constexpr (DEPTH) Dp += e1rgb.w * w;
                if constexpr (DUAL) {
                    const float4 u = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rgb2) + o1);
                    E0 += u.x * w; E1 += u.y * w; E2 += u.z * w;
                }
                T = bl ? test_T : T;
                last = bl ? __float_as_uint(e1xy.z) : last;
                done = done || (act != bl);
#ifdef D3GA_DIAG_COUNTERS
                df_blend += __popcll(__builtin_amdgcn_ballot_w64(bl));
#endif
#ifdef D3GA_DIAG_TIMELINE
                df_iters += 1;
#endif
            }
            if (__builtin_amdgcn_ballot_w64(done) == ~0ull) { i = trip; all_done = true; }      // whole quadrant saturated
        }
#ifdef D3GA_DIAG_TIMELINE
        df_blend_ticks += __builtin_amdgcn_s_memrealtime() - df_tb;
#endif
        if (D3GA_FWD_USED && blk_base) {
            // the row's latest contributor so far (1-based position in the tile list, 0: none) against the positions of this batch
            const uint32_t rowlast = row_max_u32(last);
            const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)rowlast, 0), l1 = (uint32_t)__builtin_amdgcn_readlane((int)rowlast, 16);
            const uint32_t l2 = (uint32_t)__builtin_amdgcn_readlane((int)rowlast, 32), l3 = (uint32_t)__builtin_amdgcn_readlane((int)rowlast, 48);
            const uint32_t n0 = (uint32_t)__popcll(m[0] & __ballot(cpg.x <= l0)), n1 = (uint32_t)__popcll(m[1] & __ballot(cpg.x <= l1));
            const uint32_t n2 = (uint32_t)__popcll(m[2] & __ballot(cpg.x <= l2)), n3 = (uint32_t)__popcll(m[3] & __ballot(cpg.x <= l3));
            if (em0 && n0) used0 = bcb0 + n0;
            if (em1 && n1) used1 = bcb1 + n1;
            if (em2 && n2) used2 = bcb2 + n2;
            if (em3 && n3) used3 = bcb3 + n3;
        }
        if (all_done) break;
    }
    if constexpr (L1V) {
        // fused L1 value (d3ga_raster_composite_fwd_l1): every quadrant leaves sum |colour - target| / n of its pixels; the
        // launcher's second kernel adds the partials in index order (reproducible), no pass over the finished image
        float d = inside ? fabsf(C0 + T * bg[0] - tg0) + fabsf(C1 + T * bg[1] - tg1) + fabsf(C2 + T * bg[2] - tg2) : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off);
        if (lane == 0) l1v.partials[4 * (size_t)q.tile + q.quad] = d * l1v.inv_n;
    }
    if (inside) {
        const size_t hw = (size_t)H * W;
        const size_t pid1 = hw * q.view + (size_t)rg.py * W + rg.px, pid = pid1 + 2 * hw * q.view;      // one-plane / three-plane images of the view
        final_T[pid1] = T;
        n_contrib[pid1] = last;
        out_color[pid] = C0 + T * bg[0];
        out_color[hw + pid] = C1 + T * bg[1];
        out_color[2 * hw + pid] = C2 + T * bg[2];
        if constexpr (DEPTH) out_invdepth[pid1] = Dp;
        if constexpr (DUAL) {
            out_color2[pid] = E0 + T * bg2[0];
            out_color2[hw + pid] = E1 + T * bg2[1];
            out_color2[2 * hw + pid] = E2 + T * bg2[2];
        }
    }
    if (blk_count && (lane & 15) == 0) {
        const int r = lane >> 4;
        if (D3GA_FWD_USED) blk_count[16 * (size_t)q.tile + 4 * q.quad + r] = r == 0 ? used0 : (r == 1 ? used1 : (r == 2 ? used2 : used3));
        else blk_count[16 * (size_t)q.tile + 4 * q.quad + r] = r == 0 ? bc0 : (r == 1 ? bc1 : (r == 2 ? bc2 : bc3));
    }
#ifdef D3GA_DIAG_COUNTERS
    if (lane == 0 && end > begin) {
        atomicAdd(&g_diag_fwd[1], df_chunks); atomicAdd(&g_diag_fwd[4], df_pairs); atomicAdd(&g_diag_fwd[5], df_blend);
        atomicAdd(&g_diag_fwd[6], df_surv); atomicAdd(&g_diag_fwd[7], (unsigned long long)(sbase - begin));
    }
#endif
#ifdef D3GA_DIAG_TIMELINE
    if (lane == 0 && end > begin && blockIdx.x < 32768) {   // no atomics: a returning same-address atomic per wave serialises at the memory side and would BE the timeline
        const unsigned long long df_t1 = __builtin_amdgcn_s_memrealtime();
        const size_t slot = blockIdx.x;
        g_diag_fwd_waves[4 * slot] = df_t0; g_diag_fwd_waves[4 * slot + 1] = df_t1;
        g_diag_fwd_waves[4 * slot + 2] = df_iters | (df_batches << 32); g_diag_fwd_waves[4 * slot + 3] = (unsigned long long)min(end - begin, 0xfffffu) | ((df_blend_ticks & 0xfffffull) << 20) |
                                         ((unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xffff) << 40) | ((unsigned long long)(__builtin_amdgcn_s_getreg(6164) & 15) << 56);
    }
#endif
}

// test hook (d3ga_selftest_alpha): the forward's OWN evaluation of alpha for listed (Gaussian, pixel) pairs -- conic_q() of the
// geometry record, splat_eval_q() on (centre - pixel) exactly as the blend loop forms them -- so that a parity test can hand
// the product's alpha >= 1/255 decisions to the oracle at the pairs where two float32 exps may disagree
__global__ void alpha_selftest_kernel(int n, const float2 *__restrict__ xy, const float4 *__restrict__ conic_o,
                                      const int32_t *__restrict__ gid, const int32_t *__restrict__ px, const int32_t *__restrict__ py,
                                      uint8_t *__restrict__ ok_out, float *__restrict__ alpha_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 c = xy[2 * (size_t)gid[i]];
    const float4 co = conic_o[gid[i]];
    const ConicQ q = conic_q(co.x, co.y, co.z);
    float al, G;
    bool ok;
    splat_eval_q(c.x - (float)px[i], c.y - (float)py[i], q, co.w, al, G, ok);
    ok_out[i] = ok ? 1 : 0;
    if (alpha_out) alpha_out[i] = al;
}

}  // namespace d3ga

using namespace d3ga;

extern "C" int d3ga_selftest_alpha(int32_t P, const void *geom, int n, const int32_t *gid, const int32_t *px, const int32_t *py,
                                   uint8_t *ok, float *alpha, d3ga_stream_t stream) {
    if (P <= 0 || n < 0) return D3GA_E_SIZE;
    if (n == 0) return D3GA_OK;
    if (!geom || !gid || !px || !py || !ok) return D3GA_E_NULL;
    const GeomBuf g = carve_geom(const_cast<void *>(geom), P);
    hipLaunchKernelGGL(d3ga::alpha_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, reinterpret_cast<const float2 *>(g.xyh), g.conic_o, gid, px, py, ok, alpha);
    return check_launch((hipStream_t)stream, 0);
}

#ifdef D3GA_DIAG
extern "C" __attribute__((visibility("default"))) int d3ga_diag_fwd_read(unsigned long long *out8, unsigned long long *waves, int n, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_diag_fwd), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (waves && n > 0 && hipMemcpyFromSymbol(waves, HIP_SYMBOL(g_diag_fwd_waves), sizeof(unsigned long long) * 4 * (size_t)n) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag_fwd), z, sizeof(z)) != hipSuccess) return 1;
        void *w = nullptr;
        if (hipGetSymbolAddress(&w, HIP_SYMBOL(g_diag_fwd_waves)) != hipSuccess || hipMemset(w, 0, sizeof(unsigned long long) * 4 * 32768) != hipSuccess) return 1;
    }
    return 0;
}
#endif

static const L1Value kNoL1Value = {nullptr, nullptr, nullptr, 0.f};
static int composite_fwd_impl(const d3ga_raster_params *prm, const float *bg, const void *geom, const void *binning,
                              int64_t d_capacity, void *img, float *out_color, float *out_invdepth, const float *colors2,
                              const float *bg2, float *out_color2, d3ga_stream_t stream, const L1Value &l1v = kNoL1Value) {
    if (!prm || !bg || !geom || !binning || !img || !out_color) return D3GA_E_NULL;
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0 || d_capacity < 0) return D3GA_E_SIZE;
    if (colors2 && (!bg2 || !out_color2)) return D3GA_E_NULL;
    const int views = n_views_of(prm);
    hipStream_t s = (hipStream_t)stream;
    const int gx = tiles_x(prm->W), gyv = tiles_y(prm->H), gy = gyv * views;
    const BinBuf bin = carve_bin(const_cast<void *>(binning), (int64_t)gx * gy, d_capacity);
    const GeomBuf g = carve_geom(const_cast<void *>(geom), (int64_t)prm->P * views);
    ImgBuf im = carve_img(img, prm->W, prm->H, (int64_t)gx * gy, views);
    if (prm->forward_only) { im.blk_list = nullptr; im.blk_count = nullptr; }     // the buffer ends behind n_contrib
    const bool ordered = (composite_variant() & kVariantOrdered) != 0, exact = (composite_variant() & kVariantExactCull) != 0;
    const dim3 grid(ordered ? quad_grid_ordered(gx * gy) : quad_grid(gx, gy));
    const uint32_t *order = ordered ? (const uint32_t *)bin.tile_order : (const uint32_t *)nullptr;
#define D3GA_LAUNCH_FWD(DUALV, DEPTHV, L1VV)                                                                                    \
    hipLaunchKernelGGL((composite_fwd_q_kernel<DUALV, DEPTHV, L1VV>), grid, dim3(64), 0, s, prm->W, prm->H, gx, gy, gyv, bin.tile_start, \
                       bin.point_list, (uint64_t)d_capacity, reinterpret_cast<const float2 *>(g.xyh), g.conic_o, g.rgb_invd, g.xyh, bg, im.final_T, im.n_contrib,     \
                       out_color, out_invdepth, order, colors2, bg2, out_color2, im.blk_list, im.blk_count, exact, l1v, prm->P)
    if (colors2 && l1v.partials) return D3GA_E_CONFIG;
    if (colors2) { if (out_invdepth) D3GA_LAUNCH_FWD(true, true, false); else D3GA_LAUNCH_FWD(true, false, false); }
    else if (l1v.partials) { if (out_invdepth) D3GA_LAUNCH_FWD(false, true, true); else D3GA_LAUNCH_FWD(false, false, true); }
    else { if (out_invdepth) D3GA_LAUNCH_FWD(false, true, false); else D3GA_LAUNCH_FWD(false, false, false); }
#undef D3GA_LAUNCH_FWD
    return check_launch(s, prm->debug);
}

extern "C" int d3ga_raster_composite_fwd(const d3ga_raster_params *prm, const float *bg, const void *geom,
                                         const void *binning, int64_t d_capacity, void *img, float *out_color,
                                         float *out_invdepth, d3ga_stream_t stream) {
    return composite_fwd_impl(prm, bg, geom, binning, d_capacity, img, out_color, out_invdepth, nullptr, nullptr, nullptr, stream);
}

extern "C" int d3ga_raster_composite_fwd_l1(const d3ga_raster_params *prm, const float *bg, const void *geom,
                                            const void *binning, int64_t d_capacity, void *img, float *out_color,
                                            float *out_invdepth, const float *target, const void *target_cell,
                                            float *loss, float *partials, d3ga_stream_t stream) {
    if (!prm || !loss || !partials || (!target && !target_cell)) return D3GA_E_NULL;
    if (prm->W <= 0 || prm->H <= 0) return D3GA_E_SIZE;
    const int views = n_views_of(prm);
    const L1Value l1v = {target, (const float *const *)target_cell, partials, 1.0f / (3.0f * (float)prm->W * (float)prm->H * (float)views)};
    D3GA_TRY(composite_fwd_impl(prm, bg, geom, binning, d_capacity, img, out_color, out_invdepth, nullptr, nullptr, nullptr,
                                stream, l1v));
    launch_sum_partials(4 * tiles_x(prm->W) * tiles_y(prm->H) * views, partials, loss, (hipStream_t)stream);
    return check_launch((hipStream_t)stream, prm->debug);
}

extern "C" int d3ga_raster_composite_fwd2(const d3ga_raster_params *prm, const float *bg, const float *bg2, const void *geom,
                                          const float *colors2, const void *binning, int64_t d_capacity, void *img,
                                          float *out_color, float *out_color2, float *out_invdepth, d3ga_stream_t stream) {
    if (!colors2) return D3GA_E_NULL;
    return composite_fwd_impl(prm, bg, geom, binning, d_capacity, img, out_color, out_invdepth, colors2, bg2, out_color2, stream);
}

static int composite_bwd_impl(const d3ga_raster_params *prm, const float *bg, const void *geom, const void *binning,
                              int64_t d_capacity, const void *img, const float *dL_dpix, float *acc, const float *colors2,
                              const float *bg2, const float *dL_dpix2, const L1Source &l1, d3ga_stream_t stream,
                              const float *dL_dinvd = nullptr) {
    if (!prm) return D3GA_E_NULL;
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0 || d_capacity < 0) return D3GA_E_SIZE;
    if (prm->forward_only) return D3GA_E_CONFIG;          // the forward did not write the per-block lists
    if (prm->P == 0) return D3GA_OK;
    if (!bg || !geom || !binning || !img || !acc) return D3GA_E_NULL;
    if (!dL_dpix && !l1.image && !dL_dinvd) return D3GA_E_NULL;        // some incoming gradient: an image, the fused L1 term, the inverse depth
    if (l1.image && (!(l1.target || l1.target_cell) || !l1.g_loss)) return D3GA_E_NULL;
    if (colors2 && (!bg2 || !dL_dpix2)) return D3GA_E_NULL;
    const int views = n_views_of(prm);
    const int gx = tiles_x(prm->W), gy = tiles_y(prm->H) * views;
    const BinBuf bin = carve_bin(const_cast<void *>(binning), (int64_t)gx * gy, d_capacity);
    const GeomBuf g = carve_geom(const_cast<void *>(geom), (int64_t)prm->P * views);
    const ImgBuf im = carve_img(const_cast<void *>(img), prm->W, prm->H, (int64_t)gx * gy, views);
    return launch_composite_bwd_scan(prm, gx, gy, bin, g, im, d_capacity, bg, dL_dpix, acc,
                                     (composite_variant() & kVariantOrdered) != 0, colors2, bg2, dL_dpix2, l1, (hipStream_t)stream, dL_dinvd);
}

static const L1Source kNoL1 = {nullptr, nullptr, nullptr, nullptr, 0.f};

extern "C" int d3ga_raster_composite_bwd(const d3ga_raster_params *prm, const float *bg, const void *geom,
                                         const void *binning, int64_t d_capacity, const void *img,
                                         const float *dL_dpix, float *acc, d3ga_stream_t stream) {
    if (!dL_dpix) return prm && prm->P == 0 ? D3GA_OK : D3GA_E_NULL;
    return composite_bwd_impl(prm, bg, geom, binning, d_capacity, img, dL_dpix, acc, nullptr, nullptr, nullptr, kNoL1, stream);
}

extern "C" int d3ga_raster_composite_bwd_depth(const d3ga_raster_params *prm, const float *bg, const void *geom,
                                               const void *binning, int64_t d_capacity, const void *img,
                                               const float *dL_dpix, const float *dL_dinvdepth, float *acc,
                                               d3ga_stream_t stream) {
    if (!dL_dpix && !dL_dinvdepth) return prm && prm->P == 0 ? D3GA_OK : D3GA_E_NULL;
    return composite_bwd_impl(prm, bg, geom, binning, d_capacity, img, dL_dpix, acc, nullptr, nullptr, nullptr, kNoL1, stream, dL_dinvdepth);
}

extern "C" int d3ga_raster_composite_bwd2(const d3ga_raster_params *prm, const float *bg, const float *bg2, const void *geom,
                                          const float *colors2, const void *binning, int64_t d_capacity, const void *img,
                                          const float *dL_dpix, const float *dL_dpix2, float *acc, d3ga_stream_t stream) {
    if (!colors2 || !dL_dpix) return D3GA_E_NULL;
    return composite_bwd_impl(prm, bg, geom, binning, d_capacity, img, dL_dpix, acc, colors2, bg2, dL_dpix2, kNoL1, stream);
}

extern "C" int d3ga_raster_composite_bwd_l1(const d3ga_raster_params *prm, const float *bg, const void *geom,
                                            const void *binning, int64_t d_capacity, const void *img, const float *image,
                                            const float *target, const void *target_cell, const float *g_loss,
                                            const float *dL_dpix, float *acc, d3ga_stream_t stream) {
    if (!prm) return D3GA_E_NULL;
    if (!image) return D3GA_E_NULL;
    const L1Source l1 = {image, target, (const float *const *)target_cell, g_loss, 1.0f / (3.0f * (float)prm->W * (float)prm->H * (float)n_views_of(prm))};
    return composite_bwd_impl(prm, bg, geom, binning, d_capacity, img, dL_dpix, acc, nullptr, nullptr, nullptr, l1, stream);
}

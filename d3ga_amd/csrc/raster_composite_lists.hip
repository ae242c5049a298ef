// raster_composite_lists.hip -- alpha compositing forward as TWO launches (round 5):
//
//   tile_cull_kernel            one 256-thread workgroup per tile walks the tile's depth-sorted list ONCE, 256 entries at a time
//                               (one entry per thread: two 16-byte gathers), finds the 4x4-pixel blocks of the tile the entry's
//                               alpha >= 1/255 ellipse can touch (bounding box, then the exact ellipse / block-line intersection,
//                               four block lines per entry, no divergent loop) and appends {1-based list position, Gaussian id}
//                               to those blocks' lists in ImgBuf::blk_list IN LIST ORDER (16 ballots per wavefront, the four
//                               wavefronts' offsets through 128 bytes of LDS, one barrier per 256 entries);
//   composite_fwd_lists_kernel  one wavefront = four blocks of one tile (the tile's 16 blocks sorted by list length, ranks
//                               4g .. 4g+3 per wavefront); each 16-lane row walks ITS block's list 16 entries at a time:
//                               lane l of the row gathers entry 16 g + l (records of the next group in flight while the current
//                               one is blended), folds the conic and forms, ONCE per (entry, pixel line), the two terms of the
//                               exponent that depend on the line only; the row then blends the 16 entries front to back out
//                               of a wave-private LDS slab with immediate-offset broadcast reads (two 16-byte reads and 18 VALU
//                               instructions per entry and 16 pixels; no list indirection, no ballots, no scalar bookkeeping).
//
// Against the one-launch forward it replaces on the training path (raster_composite.hip: composite_fwd_q_kernel, still the
// forward of renders that need no backward -- it allocates no block lists): a tile's list is scanned once instead of once per
// quadrant, the culling leaves the blend's instruction stream, the rows' work is known before the blend is launched (blocks of
// similar length share a wavefront), and blk_count shrinks to the prefix of a block's list that actually blended somewhere
// (the backward walks exactly that).  The (pixel, entry) pairs blended, their order and their alphas are those of the
// one-launch forward: culled entries provably have alpha < 1/255 on every pixel of the block; list positions (n_contrib) stay
// positions in the FULL tile list.  Numbers: DESIGN.md sec. 4.
#include "composite_common.h"

#ifndef D3GA_LISTS_WAVES
#define D3GA_LISTS_WAVES 5
#endif
// A/B (build.py D3GA_VARIANT): 1 = wavefronts with many groups ahead of them raise their issue priority (s_setprio): the launch
// is one round of resident wavefronts, its span is the heaviest wavefront's path under fair sharing of its SIMD
#ifndef D3GA_LISTS_PRIO
#define D3GA_LISTS_PRIO 0
#endif
// timing ablations of tile_cull_kernel (tools/_build variants, WRONG results): 1 no list stores (blk_total = 0: the blend sees empty
// lists), 2 no span gather (mask from the id's bits), 3 both
#ifndef D3GA_CULL_ABL
#define D3GA_CULL_ABL 0
#endif

namespace d3ga {

#ifdef D3GA_DIAG
// diagnostic build only (tools/diag_lists.py): per blend wavefront / per cull workgroup, indexed by workgroup (no atomics):
// start, end (s_memrealtime), groups | longest row list << 16 | tile list length << 32, HW_ID | XCC_ID << 32
__device__ unsigned long long g_diag_lists_blend[32768 * 4];
__device__ unsigned long long g_diag_lists_cull[16384 * 4];
__device__ unsigned long long g_diag_lists_cnt[8];          // [0] blend steps issued (wave x entry), [1] blended (entry, pixel) pairs, [2] (entry, block) pairs written
#endif

#ifndef D3GA_CULL_THREADS
#define D3GA_CULL_THREADS 256
#endif
#ifndef D3GA_CULL_SUB
#define D3GA_CULL_SUB 8
#endif
constexpr int kCullThreads = D3GA_CULL_THREADS, kCullSub = D3GA_CULL_SUB, kCullWaves = kCullThreads / 64, kCullChunks = kCullWaves * kCullSub;          // a round = up to kCullSub x 256 entries of the tile's list, kCullSub per thread

// History of this kernel (C3, rocprofv3): 256 entries per round, the geometric test per (tile, entry), 16 ballots per wavefront
// and round: 32 us -- ten dependent rounds of list -> records -> barrier -> stores for the longest lists.  1024 entries per round
// (four per thread, loads issued together), one packed prefix scan per chunk instead of the ballots: 39 us -- 12.2 M VALU
// wave-instructions, 62 % of the wave cycles waiting: the 1334 workgroups all start together and march through load / compute /
// store phases in step.  Now the geometric test is done ONCE PER GAUSSIAN by preprocess (GeomBuf::span, one column interval per
// block line: splat_spans) and a thread decodes it for its tile with integer arithmetic (one 16-byte gather per entry instead of
// two, ~55 instead of ~250 instructions), and a round takes up to 2048 entries: one round for nearly every tile.
// Chunk c = 4 j + wavefront holds entries 256 j + 64 wavefront + lane of the round (list order = chunk order); per chunk ONE packed
// prefix scan gives every lane its rank in all 16 block lists (four dwords of four 8-bit counters) and the chunk's 16 totals;
// the offsets of a wavefront's chunks come from passes over the table of totals in LDS (lane = 16 j + block).
__global__ __launch_bounds__(kCullThreads) void tile_cull_kernel(
    int gx, int tiles, const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list, uint64_t dcap,
    const float4 *__restrict__ conic_o, const float4 *__restrict__ xyh, const uint4 *__restrict__ span,
    const uint32_t *__restrict__ tile_order, uint2 *__restrict__ blk_list, uint32_t *__restrict__ blk_total, bool exact_cull) {
    // tile_order == null (D3GA_CULL_ORDERED=0, an A/B): workgroup b runs on XCD b % 8 (observed; speed only) and XCD x takes the tiles
    // [x T/8, (x+1) T/8) in row-major order -- a horizontal band of the image -- so that the span record of a Gaussian (it sits in ~3
    // neighbouring tiles) is fetched into ONE L2.  Measured: 57-63 us against 37-42 us for the work-ordered deal of the compositing
    // launches (the bands are unevenly loaded); the default is the work order.
    const int per = (tiles + 7) / 8;
    if (tile_order ? (int)blockIdx.x >= tiles : (int)blockIdx.x >= 8 * per) return;      // (the grid is rounded up to a multiple of 8)
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((unsigned)tile >= (unsigned)tiles) return;
    const uint32_t begin = (uint32_t)min((uint64_t)tile_start[tile], dcap);
    const uint32_t end = (uint32_t)min((uint64_t)tile_start[tile + 1], dcap);
    if (begin >= end) {                                     // uniform: empty tile
        if (threadIdx.x < 16) blk_total[16 * (size_t)tile + threadIdx.x] = 0u;
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef D3GA_DIAG_TIMELINE
    const unsigned long long dg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tcx = tile % gx, tcy = tile / gx;
    const float tx0 = (float)(tcx * kTile), ty0 = (float)(tcy * kTile);
    const uint32_t cap = end - begin;
    uint2 *const base = blk_list + 16 * (size_t)begin;
    __shared__ uint32_t s_cnt[2][kCullChunks][16];          // [parity][chunk = kCullWaves j + wavefront][block]: hits of this round
    __shared__ uint32_t s_run[2][16];                       // [parity][block]: entries written in earlier rounds
    if (threadIdx.x < 16) s_run[0][threadIdx.x] = 0u;

    int par = 0;
    for (uint32_t sbase = begin; sbase < end; sbase += kCullSub * kCullThreads, par ^= 1) {
        const int J = (int)min((uint32_t)kCullSub, (end - sbase + (uint32_t)kCullThreads - 1u) / (uint32_t)kCullThreads);   // uniform
        uint32_t id[kCullSub], mask[kCullSub];              // (the ranks are formed twice -- totals before the barrier, ranks behind it --
        uint4 sp[kCullSub];                                 //  rather than kept: 32 registers at eight entries per thread)
#pragma unroll
        for (int j = 0; j < kCullSub; ++j) {
            const uint32_t i = sbase + (uint32_t)(j * kCullThreads) + threadIdx.x;
            id[j] = (j < J && i < end) ? point_list[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < kCullSub; ++j)
            if (j < J) {
                if (D3GA_CULL_ABL & 2) sp[j] = make_uint4((id[j] * 2654435761u) & 0x00070007u, 0x21u | ((id[j] & 3u) << 4), 0u, 0u);
                else sp[j] = span[id[j]];                   // (id 0 is always readable)
            }
#pragma unroll
        for (int j = 0; j < kCullSub; ++j) {
            uint32_t mine = 0u;
            mask[j] = 0u;
            if (j < J) {
                const uint32_t i = sbase + (uint32_t)(j * kCullThreads) + threadIdx.x;
                const bool have = i < end;
                const bool big = have && span_is_big(sp[j]);
                mask[j] = (have && !big) ? span_mask16(sp[j], 4 * tcx, 4 * tcy) : 0u;
                if (__builtin_amdgcn_ballot_w64(big) != 0ull) {     // rare: a splat too large for a span record -- the geometric test
                    if (big) mask[j] = block_mask16_slow(id[j], xyh, conic_o, tx0, ty0, exact_cull);
                }
                uint32_t tot[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t pk = (((mask[j] >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u;      // bit k of the nibble -> byte k
                    const uint32_t incl = wave_incl_scan_u32(pk);
                    tot[q] = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
                const uint32_t t = lane < 4 ? tot[0] : (lane < 8 ? tot[1] : (lane < 12 ? tot[2] : tot[3]));
                mine = (t >> (8 * (lane & 3))) & 0xffu;
            }
            if (lane < 16) s_cnt[par][kCullWaves * j + wave][lane] = mine;
        }
        __syncthreads();
        // lane 16 jj + b of pass h: where the hits of chunk kCullWaves (4 h + jj) + wave land in block b's list
        uint32_t offs[(kCullSub + 3) / 4];
        {
            const int jj = lane >> 4, b = lane & 15;
            const uint32_t run = s_run[par][b];
            uint32_t all = run;
#pragma unroll
            for (int h = 0; h < (kCullSub + 3) / 4; ++h) {
                const int c = kCullWaves * (4 * h + jj) + wave;
                uint32_t s0 = run;
#pragma unroll
                for (int cc = 0; cc < kCullChunks; ++cc) {
                    const uint32_t v = s_cnt[par][cc][b];
                    s0 += cc < c ? v : 0u;
                    if (h == 0) all += v;
                }
                offs[h] = s0;
            }
            if (wave == 0 && jj == 0) s_run[par ^ 1][b] = all;
        }
#pragma unroll
        for (int j = 0; j < kCullSub; ++j) {
            if (j < J) {
                const uint32_t i = sbase + (uint32_t)(j * kCullThreads) + threadIdx.x;
                const uint2 rec = make_uint2(i - begin + 1u, id[j]);
                uint32_t P[4];                              // this lane's rank in the chunk's 16 lists, 8 bits each
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t pk = (((mask[j] >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u;
                    P[q] = wave_incl_scan_u32(pk) - pk;
                }
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const uint32_t ob = (uint32_t)__builtin_amdgcn_readlane((int)offs[j >> 2], 16 * (j & 3) + b);
                    if (((mask[j] >> b) & 1u) && !((D3GA_CULL_ABL & 1) && cap != 0xffffffffu)) base[(size_t)b * cap + ob + ((P[b >> 2] >> (8 * (b & 3))) & 0xffu)] = rec;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) blk_total[16 * (size_t)tile + threadIdx.x] = (D3GA_CULL_ABL & 1) ? 0u : s_run[par][threadIdx.x];
#ifdef D3GA_DIAG_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        unsigned long long *r = g_diag_lists_cull + 4 * (size_t)blockIdx.x;
        r[0] = dg_t0; r[1] = __builtin_amdgcn_s_memrealtime(); r[2] = (unsigned long long)(end - begin);
        r[3] = (unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xffff) | ((unsigned long long)(__builtin_amdgcn_s_getreg(6164) & 15) << 32);
    }
#endif
}

// DUAL / DEPTH / L1V: as composite_fwd_q_kernel (a second set of colours blended with the same alphas; the inverse-depth image;
// the L1 loss value against a target image: one partial per wavefront in partials[4 * tile + wavefront index])
template <bool DUAL, bool DEPTH, bool L1V>
__global__ __launch_bounds__(64, D3GA_LISTS_WAVES) void composite_fwd_lists_kernel(
    int W, int H, int gx, int gy, const uint32_t *__restrict__ tile_start, uint64_t dcap, const float2 *__restrict__ xy,
    const float4 *__restrict__ conic_o, const float4 *__restrict__ rgb_invd, const float *__restrict__ bg,
    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, float *__restrict__ out_color,
    float *__restrict__ out_invdepth, const uint32_t *__restrict__ tile_order, const float *__restrict__ colors2,
    const float *__restrict__ bg2, float *__restrict__ out_color2, const uint2 *__restrict__ blk_list,
    const uint32_t *__restrict__ blk_total, uint32_t *__restrict__ blk_used, L1Value l1v) {
    // workgroup b -> (tile rank k, wavefront w of the tile): the tile's four wavefronts are workgroups b, b+8, b+16, b+24 of one
    // XCD (they share an L2), ranks dealt round-robin over the XCDs, heaviest tile first (quad_of_block_ordered)
    const int b = blockIdx.x, tiles = gx * gy;
    const int k = (b & 7) + 8 * (b >> 5), w = (b >> 3) & 3;
    if (k >= tiles) return;
    const int tile = tile_order ? (int)tile_order[k] : k;
    const int tx0 = (tile % gx) * kTile, ty0 = (tile / gx) * kTile;
    const int lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15;
    const uint32_t begin = (uint32_t)min((uint64_t)tile_start[tile], dcap);
    const uint32_t end = (uint32_t)min((uint64_t)tile_start[tile + 1], dcap);
    const bool has = begin < end;                           // uniform
    const uint32_t cap = end - begin;
#ifdef D3GA_DIAG_TIMELINE
    const unsigned long long dg_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dg_groups = 0, dg_stage = 0;
#endif

    // the tile's 16 blocks by descending list length; this wavefront takes ranks 4 g .. 4 g + 3 (blocks of similar length: a
    // wavefront runs as long as its longest row), g rotating with the tile's rank
    __shared__ uint8_t s_perm[16];
    int blk = 4 * w + row;
    if (has) {
        const uint32_t myc = lane < 16 ? blk_total[16 * (size_t)tile + lane] : 0u;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)myc, j);
            rank += (c > myc || (c == myc && j < lane)) ? 1 : 0;
        }
        if (lane < 16) s_perm[rank] = (uint8_t)lane;
        __builtin_amdgcn_wave_barrier();
        blk = s_perm[4 * ((w + k) & 3) + row];
    }
    const int bq = blk >> 2, br = blk & 3;
    const int bx = 2 * (bq & 1) + (br & 1), by = 2 * (bq >> 1) + (br >> 1);
    const int px = tx0 + 4 * bx + (l16 & 3), py = ty0 + 4 * by + (l16 >> 2);
    const bool inside = px < W && py < H;
    const unsigned long long inm = __ballot(inside);
    if (inm == 0ull) {                                      // a wavefront without pixels (tile on the image border)
        if (L1V && lane == 0) l1v.partials[4 * (size_t)tile + w] = 0.f;
        if (has && l16 == 0) blk_used[16 * (size_t)tile + blk] = 0u;
        return;
    }
    const bool row_in = ((inm >> (16 * row)) & 0xffffull) != 0ull;
    const float fx = (float)px;
    float tg0 = 0.f, tg1 = 0.f, tg2 = 0.f;
    if (L1V && inside) {                                    // the target's pixel is requested NOW and used after the blend
        const float *tg = l1v.target_cell ? *l1v.target_cell : l1v.target;
        const size_t pid = (size_t)py * W + px, hw = (size_t)H * W;
        tg0 = tg[pid]; tg1 = tg[hw + pid]; tg2 = tg[2 * hw + pid];
    }
    const uint2 *const list = blk_list + 16 * (size_t)begin + (size_t)blk * cap;
    const uint32_t n = (has && row_in) ? blk_total[16 * (size_t)tile + blk] : 0u;

    // one staged entry of a row = five 16-byte words: {folded conic a, opacity, centre x, red} and per pixel line of the block
    // {green, blue, tb = q.b dy, tc = q.c dy^2}: a pixel reads word 0 and the word of ITS line
    // LDS layout: the 4 rows' words of ONE entry index are contiguous -- s_a[i][row] (64 B per index), s_l[i][4 row + line] (256 B
    // per index + 16 B of padding: 17 words) -- so that the 4 (16) distinct addresses of a broadcast read fall into distinct banks
    // and the 64 staging lanes' writes spread evenly over all of them.  (First layout [row][entry][5 words]: row stride 1280 B =
    // a multiple of the 128-byte bank cycle -> every read a 4-way conflict, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.24.)
    __shared__ float4 s_a[16][4];
    __shared__ float4 s_l[16][17];
    __shared__ float4 s_ext[(DUAL || DEPTH) ? 16 : 1][4];   // {1 / depth, second colour}
    const float4 *const rowbase = &s_a[0][row];             // entry i: + 4 i words
    const float4 *const linebase = &s_l[0][4 * row + (l16 >> 2)];      // entry i: + 17 i words
    const float y0 = (float)(ty0 + 4 * by);

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, E0 = 0.f, E1 = 0.f, E2 = 0.f;
    int lastk = -1;                                         // index in the block's list of the last entry this pixel blended
    bool done = !inside;

    // ---- two levels in flight: the records of group g + 1 (three gathers per entry) and the list entries of group g + 2, so that
    // neither the list load nor the gathers behind it are waited for at the top of a group (first version: both issued together
    // at the top of group g for g + 1 -- the gathers wait for the list entry, 35 % of a wavefront's lifetime: tools/diag_lists.py) ----
    uint2 pg = make_uint2(0u, 0u);                          // list entry whose records are in flight (pos 0: none)
    uint2 pgn = make_uint2(0u, 0u);                         // list entry of the group after that (in flight)
    float2 nxy = make_float2(0.f, 0.f);
    float4 nco = make_float4(0.f, 0.f, 0.f, 0.f), nrgb = nco;
    float n2r = 0.f, n2g = 0.f, n2b = 0.f;
    auto load_entry = [&](uint32_t g, bool row_live) {      // -> pgn
        const uint32_t kk = 16u * g + (uint32_t)l16;
        pgn = make_uint2(0u, 0u);
        if (row_live && kk < n) pgn = list[kk];
    };
    auto gather = [&]() {                                   // records of pgn -> the n* registers; pg <- pgn
        pg = pgn;
        nxy = make_float2(0.f, 0.f); nco = make_float4(0.f, 0.f, 0.f, 0.f); nrgb = nco;
        n2r = n2g = n2b = 0.f;
        if (pg.x != 0u) {
            nxy = xy[2 * (size_t)pg.y]; nco = conic_o[pg.y]; nrgb = rgb_invd[pg.y];
            if constexpr (DUAL) { n2r = colors2[3 * (size_t)pg.y]; n2g = colors2[3 * (size_t)pg.y + 1]; n2b = colors2[3 * (size_t)pg.y + 2]; }
        }
    };
    const uint32_t ngroups = (wave_max_u32(n) + 15u) >> 4;
#if D3GA_LISTS_PRIO == 1
    if (ngroups >= 14u) __builtin_amdgcn_s_setprio(3);
    else if (ngroups >= 11u) __builtin_amdgcn_s_setprio(2);
    else if (ngroups >= 8u) __builtin_amdgcn_s_setprio(1);
#elif D3GA_LISTS_PRIO == 2
    if (ngroups >= 12u) __builtin_amdgcn_s_setprio(3);
#endif
    load_entry(0u, true);
    gather();
    load_entry(1u, true);
    for (uint32_t g = 0; g < ngroups; ++g) {
        // ---- stage the group whose records have arrived; put the next one in flight ----
        const float2 cxy = nxy;
        const float4 cco = nco, crgb = nrgb;
        const float c2r = n2r, c2g = n2g, c2b = n2b;
        const bool have = pg.x != 0u;
#ifdef D3GA_DIAG_TIMELINE
        dg_groups += 1;
        const unsigned long long dg_ts = __builtin_amdgcn_s_memrealtime();
#endif
        const unsigned long long dm = __builtin_amdgcn_ballot_w64(done);
        gather();                                           // group g + 1: its list entries arrived a group ago
        load_entry(g + 2u, ((dm >> (16 * row)) & 0xffffull) != 0xffffull);
        __builtin_amdgcn_wave_barrier();                    // the previous group's slab reads are done (program order)
        {
            const ConicQ cq = conic_q(cco.x, cco.y, cco.z);
            s_a[l16][row] = make_float4(cq.a, have ? cco.w : 0.f, cxy.x, crgb.x);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const float dy = cxy.y - (y0 + (float)l);    // = e.y - (float)py of the pixels on line l (integers: exact)
                s_l[l16][4 * row + l] = make_float4(crgb.y, crgb.z, cq.b * dy, (cq.c * dy) * dy);
            }
            if constexpr (DUAL || DEPTH) s_ext[l16][row] = make_float4(crgb.w, c2r, c2g, c2b);
        }
        __builtin_amdgcn_wave_barrier();
#ifdef D3GA_DIAG_TIMELINE
        dg_stage += __builtin_amdgcn_s_memrealtime() - dg_ts;       // waiting for the group's records + staging them
#endif
        // ---- blend the 16 entries front to back ----
        const int kbase = (int)(16u * g);
        int li = -1;
        auto step = [&](int i) {                            // (i is a constant at every call: immediate LDS offsets)
            const float4 ea = rowbase[4 * i];
            const float4 el = linebase[17 * i];
            float al, G;
            bool ok;
            splat_eval_q(ea.z - fx, el.z, el.w, ea.x, ea.y, al, G, ok);
            const bool act = ok && !done;
            const float test_T = T * (1.0f - al);
            const bool keep = !(test_T < kTmin);
            const bool bl = act && keep;
            const float wgt = bl ? al * T : 0.f;
            C0 += ea.w * wgt; C1 += el.x * wgt; C2 += el.y * wgt;
            if constexpr (DUAL || DEPTH) {
                const float4 ex = s_ext[i][row];
                if constexpr (DEPTH) Dp += ex.x * wgt;
                if constexpr (DUAL) { E0 += ex.y * wgt; E1 += ex.z * wgt; E2 += ex.w * wgt; }
            }
            T = bl ? test_T : T;
            li = bl ? i : li;                                 // (an inline constant: no add per entry; folded into lastk per group)
            done = done || (act != bl);
        };
        auto saturated = [&]() { return __builtin_amdgcn_ballot_w64(done) == ~0ull; };      // every pixel of the wavefront
        step(0); step(1); step(2); step(3);
        if (!saturated()) {
            step(4); step(5); step(6); step(7);
            if (!saturated()) {
                step(8); step(9); step(10); step(11);
                if (!saturated()) { step(12); step(13); step(14); step(15); }
            }
        }
        const bool all_done = saturated();
        lastk = li >= 0 ? kbase + li : lastk;
        if (all_done) break;
        // rows whose list is exhausted have nothing left; the wavefront goes on while some unfinished pixel has entries ahead
        if (__builtin_amdgcn_ballot_w64(!done && 16u * (g + 1u) < n) == 0ull) break;
    }

    uint32_t last = 0u;
    if (lastk >= 0) last = list[lastk].x;                   // 1-based position in the FULL tile list
    if constexpr (L1V) {
        float d = inside ? fabsf(C0 + T * bg[0] - tg0) + fabsf(C1 + T * bg[1] - tg1) + fabsf(C2 + T * bg[2] - tg2) : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off);
        if (lane == 0) l1v.partials[4 * (size_t)tile + w] = d * l1v.inv_n;
    }
    if (inside) {
        const size_t pid = (size_t)py * W + px;
        const size_t hw = (size_t)H * W;
        final_T[pid] = T;
        n_contrib[pid] = last;
        out_color[pid] = C0 + T * bg[0];
        out_color[hw + pid] = C1 + T * bg[1];
        out_color[2 * hw + pid] = C2 + T * bg[2];
        if constexpr (DEPTH) out_invdepth[pid] = Dp;
        if constexpr (DUAL) {
            out_color2[pid] = E0 + T * bg2[0];
            out_color2[hw + pid] = E1 + T * bg2[1];
            out_color2[2 * hw + pid] = E2 + T * bg2[2];
        }
    }
    if (has) {                                              // what the backward walks: the prefix up to the last entry some pixel blended
        const uint32_t used = row_max_u32((uint32_t)(lastk + 1));
        if (l16 == 0) blk_used[16 * (size_t)tile + blk] = used;
    }
#ifdef D3GA_DIAG_TIMELINE
    const uint32_t dg_nmax = wave_max_u32(n);
    if (lane == 0 && has && blockIdx.x < 32768) {
        unsigned long long *r = g_diag_lists_blend + 4 * (size_t)blockIdx.x;
        r[0] = dg_t0; r[1] = __builtin_amdgcn_s_memrealtime();
        r[2] = dg_groups | ((unsigned long long)min(dg_nmax, 0xffffu) << 16) | ((unsigned long long)(end - begin) << 32);
        r[3] = (unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xffff) | ((unsigned long long)(__builtin_amdgcn_s_getreg(6164) & 15) << 32) | ((dg_stage & 0xfffffull) << 40);
    }
#endif
}

int launch_composite_fwd_lists(const d3ga_raster_params *prm, int gx, int gy, const BinBuf &bin, const GeomBuf &g, const ImgBuf &im,
                               int64_t d_capacity, const float *bg, float *out_color, float *out_invdepth, const float *colors2,
                               const float *bg2, float *out_color2, bool ordered, bool exact, const L1Value &l1v, bool lists_ready,
                               hipStream_t s) {
    const int tiles = gx * gy;
    const uint32_t *order = ordered ? (const uint32_t *)bin.tile_order : (const uint32_t *)nullptr;
    static const bool cull_ordered = [] { const char *e = getenv("D3GA_CULL_ORDERED"); return e ? atoi(e) != 0 : true; }();      // A/B: 0 = the band mapping
    static const int cull_grid = [] { const char *e = getenv("D3GA_CULL_GRID"); return e ? atoi(e) : 0; }();      // experiment: launch only the first n ranks
    if (!lists_ready) {                                     // (ready: d3ga_raster_bin_sort_lists emitted them from the per-tile sort)
        hipLaunchKernelGGL(tile_cull_kernel, dim3(cull_grid > 0 ? cull_grid : 8 * ((tiles + 7) / 8)), dim3(kCullThreads), 0, s, gx, tiles, (const uint32_t *)bin.tile_start,
                           (const uint32_t *)bin.point_list, (uint64_t)d_capacity, (const float4 *)g.conic_o, (const float4 *)g.xyh,
                           (const uint4 *)g.span, cull_ordered ? order : (const uint32_t *)nullptr, im.blk_list, im.blk_total, exact);
        D3GA_TRY(check_launch(s, prm->debug));
    }
    const dim3 grid(quad_grid_ordered(tiles));
#define D3GA_LAUNCH_LISTS(DUALV, DEPTHV, L1VV)                                                                                   \
    hipLaunchKernelGGL((composite_fwd_lists_kernel<DUALV, DEPTHV, L1VV>), grid, dim3(64),                                           \
                       lds_pad_bytes((const void *)composite_fwd_lists_kernel<DUALV, DEPTHV, L1VV>, "D3GA_FWD_LDS_TOTAL"), s, prm->W, prm->H, gx, gy, \
                       (const uint32_t *)bin.tile_start, (uint64_t)d_capacity, reinterpret_cast<const float2 *>(g.xyh),              \
                       (const float4 *)g.conic_o, (const float4 *)g.rgb_invd, bg, im.final_T, im.n_contrib, out_color, out_invdepth, \
                       order, colors2, bg2, out_color2, (const uint2 *)im.blk_list, (const uint32_t *)im.blk_total, im.blk_count, l1v)
    if (colors2 && l1v.partials) return D3GA_E_CONFIG;
    if (colors2) { if (out_invdepth) D3GA_LAUNCH_LISTS(true, true, false); else D3GA_LAUNCH_LISTS(true, false, false); }
    else if (l1v.partials) { if (out_invdepth) D3GA_LAUNCH_LISTS(false, true, true); else D3GA_LAUNCH_LISTS(false, false, true); }
    else { if (out_invdepth) D3GA_LAUNCH_LISTS(false, true, false); else D3GA_LAUNCH_LISTS(false, false, false); }
#undef D3GA_LAUNCH_LISTS
    return check_launch(s, prm->debug);
}

}  // namespace d3ga

#ifdef D3GA_DIAG
extern "C" int d3ga_diag_lists_read(unsigned long long *blend, int nb, unsigned long long *cull, int nc, int reset) {
    if (blend && nb > 0 && hipMemcpyFromSymbol(blend, HIP_SYMBOL(d3ga::g_diag_lists_blend), sizeof(unsigned long long) * 4 * (size_t)nb) != hipSuccess) return 1;
    if (cull && nc > 0 && hipMemcpyFromSymbol(cull, HIP_SYMBOL(d3ga::g_diag_lists_cull), sizeof(unsigned long long) * 4 * (size_t)nc) != hipSuccess) return 1;
    if (reset) {
        void *w = nullptr;
        if (hipGetSymbolAddress(&w, HIP_SYMBOL(d3ga::g_diag_lists_blend)) != hipSuccess || hipMemset(w, 0, sizeof(unsigned long long) * 4 * 32768) != hipSuccess) return 1;
        if (hipGetSymbolAddress(&w, HIP_SYMBOL(d3ga::g_diag_lists_cull)) != hipSuccess || hipMemset(w, 0, sizeof(unsigned long long) * 4 * 16384) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// raster_composite_scan.hip -- compositing backward: composite_bwd_tile_kernel.
//
// Entry per lane, workgroup per tile.  The forward (raster_composite.hip) leaves, per 4x4 pixel block, the culled list of
// the entries that can touch it (ImgBuf::blk_list: {1-based position in the tile list, Gaussian index}).  Here
//   * one 256-thread workgroup owns a 16x16 tile; each of its four wavefronts takes four blocks (the tile's 16 blocks sorted
//     by list length, ranks 4g .. 4g+3 per wavefront: kDefaultTileAssign) and each 16-lane DPP row walks ITS block's list
//     back to front, 16 consecutive entries in its 16 lanes ("group"), stepping through the block's 16 pixels;
//   * what couples the entries at one pixel is the transmittance T_i = T_final / prod_{j >= i}(1 - alpha_j) and the colour
//     blended behind entry i, S_i = sum_{j > i} (c_j . dL/dpixel) alpha_j T_j -- two PREFIX SCANS over the row (row_shr:1,2,4,8,
//     four pixels interleaved by hand), the pixel's running (T, S) carried from group to group in a 32-byte LDS record;
//     dL/dalpha_i = T_i (c_i . g) - (S_i + T_final (bg . g)) / (1 - alpha_i): algebraically upstream's accum_rec recurrence;
//   * every lane accumulates the nine moments of ITS entry over the 16 pixels in registers (no cross-lane reduction); the
//     geometric ones as per-line constant-offset sums, centred once per group;
//   * the records of a tile's 16 blocks meet in a position-keyed MERGE CACHE in LDS (S slots of 12 dwords, slot =
//     (position - 1) mod S; the tag word is the slot's lock, taken with ONE returning integer LDS atomic per attempt --
//     float LDS atomics are 46x slower on this part) before one 64-byte accumulator line per (tile, Gaussian) leaves the CU
//     as float atomics; displaced records (live window wider than S) leave early;
//   * a wavefront that is done leaves; the last one of the tile to arrive (an LDS counter) publishes the cache;
//   * alpha is evaluated by the same splat_eval_q() as the forward and the validity test is the forward's (power <= 0,
//     alpha >= 1/255, list position <= the pixel's n_contrib): the set of (pixel, entry) pairs is exactly the forward's.
// Optionally the L1 image-loss gradient is formed per pixel here (d3ga_raster_backward_l1) and a second image's gradient
// is folded in (DUAL: render_pair).  Numbers, history and the negative results: DESIGN.md sec. 4.
#include "composite_common.h"

// D3GA_SCAN_ABL: timing ablations of the kernel below (ablation builds only -- tools/_build/, results are WRONG, the Python
// layer refuses them unless D3GA_ALLOW_ABLATION=1; tools/gpu_ablate.sh):
//   1 nothing leaves the CU and no inserts | 8 every insert but no pixel steps | 12 rows not paced | 13 inserts but nothing leaves the CU
#ifndef D3GA_SCAN_ABL
#define D3GA_SCAN_ABL 0
#endif
// tile kernel only: 13 cache inserts but nothing leaves the CU; D3GA_TILE_WAVES: wavefronts per SIMD the register budget targets
#ifndef D3GA_TILE_WAVES
#define D3GA_TILE_WAVES 4
#endif
#ifndef D3GA_TILE_PRIO
#define D3GA_TILE_PRIO 0
#endif

namespace d3ga {

// Inclusive scans over the 16 lanes of a DPP row (row_shr:1, 2, 4, 8), FOUR independent values at once: the four chains
// are interleaved by hand so that the two wait states a DPP read needs after the VALU write of its source are filled with
// useful instructions instead of s_nop (the compiler serialises the chains and pads every step).
//   mul: v_mul_f32_dpp d, d(shifted), d with bound_ctrl:0 -- a lane without a source lane is disabled and keeps d.
//   add: bound_ctrl:1 -- a lane without a source lane adds 0.
#define D3GA_SCAN4(OP, BC, N) \
    OP " %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf" BC "\n" \
    OP " %1, %1, %1 row_shr:" #N " row_mask:0xf bank_mask:0xf" BC "\n" \
    OP " %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf" BC "\n" \
    OP " %3, %3, %3 row_shr:" #N " row_mask:0xf bank_mask:0xf" BC "\n"
// R rows per block (R = 2, 4: a block's list runs over 32 / 64 lanes): the row scans are continued across the rows with
// row_bcast:15 (lane 15 of a row -> every lane of the next row; row_mask 0xa: rows 1 and 3 take it) and, for R = 4,
// row_bcast:31 (lane 31 -> rows 2 and 3) -- the wave-wide scan idiom of this ISA family, two more steps per chain.
#define D3GA_SCANX(OP, CTRL, MASK) \
    OP " %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n" \
    OP " %1, %1, %1 " CTRL " row_mask:" MASK " bank_mask:0xf\n" \
    OP " %2, %2, %2 " CTRL " row_mask:" MASK " bank_mask:0xf\n" \
    OP " %3, %3, %3 " CTRL " row_mask:" MASK " bank_mask:0xf\n"
template <int R = 1>
__device__ __forceinline__ void row_scan_mul4(float &a, float &b, float &c, float &d) {
    asm("s_nop 1\n" D3GA_SCAN4("v_mul_f32_dpp", "", 1) D3GA_SCAN4("v_mul_f32_dpp", "", 2) D3GA_SCAN4("v_mul_f32_dpp", "", 4)
        D3GA_SCAN4("v_mul_f32_dpp", "", 8)
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (R >= 2) asm("s_nop 1\n" D3GA_SCANX("v_mul_f32_dpp", "row_bcast:15", "0xa") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (R >= 4) asm("s_nop 1\n" D3GA_SCANX("v_mul_f32_dpp", "row_bcast:31", "0xc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
template <int R = 1>
__device__ __forceinline__ void row_scan_add4(float &a, float &b, float &c, float &d) {
    asm("s_nop 1\n" D3GA_SCAN4("v_add_f32_dpp", " bound_ctrl:1", 1) D3GA_SCAN4("v_add_f32_dpp", " bound_ctrl:1", 2)
        D3GA_SCAN4("v_add_f32_dpp", " bound_ctrl:1", 4) D3GA_SCAN4("v_add_f32_dpp", " bound_ctrl:1", 8)
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (R >= 2) asm("s_nop 1\n" D3GA_SCANX("v_add_f32_dpp", "row_bcast:15", "0xa") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (R >= 4) asm("s_nop 1\n" D3GA_SCANX("v_add_f32_dpp", "row_bcast:31", "0xc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

struct ScanEntry {          // one list entry as the lane that owns it holds it
    uint32_t pos, gid;      // 1-based position in the tile list (0: no entry), Gaussian index
    float2 xy;
    float4 co;              // conic a, b, c | opacity
    float4 rgb;             // colour | 1/depth (unused here)
    float c2r, c2g, c2b;    // DUAL: the second image's colour
};

template <bool DUAL>
__device__ __forceinline__ ScanEntry scan_gather(uint2 pg, const float2 *__restrict__ xy, const float4 *__restrict__ conic_o,
                                                 const float4 *__restrict__ rgb_invd, const float *__restrict__ colors2 /* DUAL: already moved back by 3 x view x P (the lists hold (view, Gaussian) indices, colors2 is per Gaussian) */) {
    ScanEntry e;
    e.pos = pg.x; e.gid = pg.y;
    e.xy = make_float2(0.f, 0.f);
    e.co = make_float4(0.f, 0.f, 0.f, 0.f);
    e.rgb = make_float4(0.f, 0.f, 0.f, 0.f);
    e.c2r = e.c2g = e.c2b = 0.f;
    if (pg.x != 0u) {
        e.xy = xy[2 * (size_t)pg.y]; e.co = conic_o[pg.y]; e.rgb = rgb_invd[pg.y];      // xy: the xyh records viewed as float2 (stride 2)
        if constexpr (DUAL) {
            e.c2r = colors2[3 * (size_t)pg.y]; e.c2g = colors2[3 * (size_t)pg.y + 1]; e.c2b = colors2[3 * (size_t)pg.y + 2];
        }
    }
    return e;
}

#ifdef D3GA_DIAG
__device__ unsigned long long g_diag_scan[16];    // diagnostic build only (tools/diag_scan.py): loop statistics of the kernel below
__device__ unsigned long long g_diag_waves[32768 * 4];   // per active wave: s_memtime at start / end, groups, HW_ID | XCC_ID << 32
#endif
// forces the compiler's s_waitcnt for these registers HERE (an empty asm that reads them)
template <bool DUAL>
__device__ __forceinline__ void scan_consume(ScanEntry &e, uint2 &pg) {
    asm volatile("" : "+v"(e.xy.x), "+v"(e.xy.y), "+v"(e.co.x), "+v"(e.co.y), "+v"(e.co.z), "+v"(e.co.w));
    asm volatile("" : "+v"(e.rgb.x), "+v"(e.rgb.y), "+v"(e.rgb.z), "+v"(pg.x), "+v"(pg.y));
    if constexpr (DUAL) asm volatile("" : "+v"(e.c2r), "+v"(e.c2g), "+v"(e.c2b));
}

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// Everything the per-tile work of one wavefront reads (the kernels' pointer arguments, by value).
struct BwdArgs {
    int W, H, gx, gyv, P;    // gyv: tile rows per view (view-batched launches: composite_common.h, Quad); P: Gaussians per view
    const float2 *xy; const float4 *conic_o; const float4 *rgb_invd; const float *bg; const float *final_T; const uint32_t *n_contrib;
    const float *dL_dpix; float *acc; const float *colors2; const float *bg2; const float *dL_dpix2; const uint2 *blk_list;
    const uint32_t *blk_count; L1Source l1; const float *dL_dinvd;
};
struct BwdDiag { unsigned long long install = 0, hit = 0, evict = 0, valid = 0, entries = 0, rowgroups = 0, trips = 0, groups = 0; };

// One wavefront's share of one tile: its SEG = 4 / R blocks (`blk`: the block of THIS lane's segment, in the forward's numbering),
// walked back to front in groups of 16 R entries, every record merged into the tile's cache `s_cache` (see the file header).
// s_pix / s_dump: this wavefront's pixel records and staging area.
template <bool DUAL, int S, bool INVD, int R>
__device__ __forceinline__ void bwd_tile_wave(const BwdArgs &A, const int tile, const uint32_t begin, const uint32_t end, const int blk,
                                              uint32_t *const s_cache, float *const s_pix, float *const s_dump, const int lane,
                                              BwdDiag &dg) {
    constexpr int PIXF = DUAL ? 12 : 8;
    constexpr int ROWF = 16 * PIXF + 4;
    constexpr int kSlot = 12;                        // dwords per cache slot
    constexpr uint32_t kLocked = 0xffffffffu;
    constexpr int LW = 16 * R;
    const int row = lane / LW, l16 = lane & 15, lseg = lane & (LW - 1);      // row: which of the wavefront's blocks; lseg: lane within the block's group
    constexpr int kVals = INVD ? 10 : 9, kPerInst = 64 / kVals;           // publish: values per record, records per instruction
    const int fq = lane / kVals, fk = lane - kVals * fq;   // lane -> (record within a group of 7 (6), value)
    const int fk_off = fk < 2 ? fk : fk + 1;               // acc layout 0,1 | 3,4,5 | 6 | 7,8,9 | 10 (dL/d(1/depth), INVD)
    constexpr int kAccStride = D3GA_ACC_STRIDE;
    const int bx = 2 * ((blk >> 2) & 1) + (blk & 1), by = 2 * (blk >> 3) + ((blk >> 1) & 1);      // the forward numbers a tile's blocks 4 * quadrant + (block within the quadrant)
    const int tyb = tile / A.gx, view = tyb / A.gyv;                                          // tile row of the batch -> (view, row of the view)
    const int bx0 = (tile - tyb * A.gx) * kTile + 4 * bx, by0 = (tyb - view * A.gyv) * kTile + 4 * by;      // block origin in pixels (local to the view)
    const int px = bx0 + (l16 & 3), py = by0 + (l16 >> 2);
    const bool inside = px < A.W && py < A.H;
    const size_t hw = (size_t)A.H * A.W;
    const size_t pid1 = hw * view + (size_t)py * A.W + px;       // one-plane images (final_T, n_contrib, inverse depth) of the view
    const size_t pid = pid1 + 2 * hw * view;                     // three-plane images
    const float T_final = inside ? A.final_T[pid1] : 0.f;
    const uint32_t last = inside ? A.n_contrib[pid1] : 0u;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (inside && A.dL_dpix) { g0 = A.dL_dpix[pid]; g1 = A.dL_dpix[hw + pid]; g2 = A.dL_dpix[2 * hw + pid]; }
    if (inside && A.l1.image) {
        // fused L1 image loss (d3ga_raster_backward_l1): dL/dpixel += dL/dloss / n * sign(image - target), formed here
        // instead of being written to and read back from a (3,A.H,A.W) gradient image
        const float *tgt = A.l1.target_cell ? *A.l1.target_cell : A.l1.target;
        const float sc = A.l1.g_loss[0] * A.l1.inv_n;
        auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
        g0 += sc * sgn(A.l1.image[pid] - tgt[pid]);
        g1 += sc * sgn(A.l1.image[hw + pid] - tgt[hw + pid]);
        g2 += sc * sgn(A.l1.image[2 * hw + pid] - tgt[2 * hw + pid]);
    }
    float bg_dot = A.bg[0] * g0 + A.bg[1] * g1 + A.bg[2] * g2;
    float h0 = 0.f, h1 = 0.f, h2 = 0.f;
    if constexpr (DUAL) {
        if (inside) { h0 = A.dL_dpix2[pid]; h1 = A.dL_dpix2[hw + pid]; h2 = A.dL_dpix2[2 * hw + pid]; }
        bg_dot += A.bg2[0] * h0 + A.bg2[1] * h1 + A.bg2[2] * h2;
    }
    const uint32_t maxlast = wave_max_u32(last);
    if (maxlast == 0) return;

    float *const pixrow = s_pix + row * ROWF;
    float *const wr_base = lseg == LW - 1 ? pixrow : s_dump + 2 * lane;
    {
        float *rec = pixrow + l16 * PIXF;
        *reinterpret_cast<float4 *>(rec) = make_float4(T_final, 0.f, g0, g1);
        float gd = 0.f;
        if constexpr (INVD) gd = inside ? A.dL_dinvd[pid1] : 0.f;
        *reinterpret_cast<float4 *>(rec + 4) = make_float4(g2, T_final * bg_dot, __uint_as_float(last), gd);
        if constexpr (DUAL) *reinterpret_cast<float4 *>(rec + 8) = make_float4(h0, h1, h2, 0.f);
    }
    const uint32_t blk_cap = end - begin;
    // (the forward writes A.blk_count only for quadrants that start inside the image)
    const bool quad_in = bx0 - 4 * (bx & 1) < A.W && by0 - 4 * (by & 1) < A.H;
    const uint32_t cnt = quad_in ? A.blk_count[16 * (size_t)tile + blk] : 0u;
    const uint2 *const list = A.blk_list + 16 * (size_t)begin + (size_t)blk * blk_cap;
    const int ngroups = (int)((wave_max_u32(cnt) + (uint32_t)(LW - 1)) / (uint32_t)LW);
    const float ddelx_dx = 0.5f * A.W, ddely_dy = 0.5f * A.H;
    const float bxr = (float)bx0, byr = (float)by0;
    const int per = D3GA_SCAN_ABL == 12 ? LW : (ngroups > 0 ? ((int)cnt + ngroups - 1) / ngroups : 0);      // rows paced to finish together (see above)
    auto list_entry = [&](int g) -> uint2 {
        const int idx = (int)cnt - 1 - per * g - lseg;
        uint2 v = list[max(idx, 0)];
        v.x = (idx >= 0 && lseg < per) ? v.x : 0u;
        return v;
    };
    const float *const colors2v = DUAL ? A.colors2 - 3 * (size_t)view * (size_t)A.P : nullptr;
    ScanEntry e = scan_gather<DUAL>(list_entry(0), A.xy, A.conic_o, A.rgb_invd, colors2v);
    uint2 pg1 = list_entry(1);
    __builtin_amdgcn_wave_barrier();

    for (int g = 0; g < ngroups; ++g) {
        ScanEntry nxt = scan_gather<DUAL>(pg1, A.xy, A.conic_o, A.rgb_invd, colors2v);
        uint2 pg2 = list_entry(g + 2);
        const bool act = e.pos != 0u;
        const float exr = e.xy.x - bxr, eyr = e.xy.y - byr;
        const ConicQ cq = conic_q(e.co.x, e.co.y, e.co.z);
        // Geometric moments with the weight gop = G dL/dalpha (the opacity factor is applied once per entry) and the pixel
        // offsets k = 0..3 of a block line as compile-time constants: per line  A = sum gop, B = sum k gop, C = sum k^2 gop
        // (7 instructions for 4 pixels), accumulated as sums of A, B, C, dy A, dy B, dy^2 A; the centred moments follow at the
        // end of the group from dx = exr - k:  sum gop dx = exr SA - SB,  sum gop dx^2 = exr^2 SA - 2 exr SB + SC, ...
        // (3.5 instructions per pixel step instead of 9: w, wx, wy and six accumulations).
        float SA = 0.f, SB = 0.f, SC = 0.f, SyA = 0.f, SyB = 0.f, SyyA = 0.f, M6 = 0.f, M7 = 0.f, M8 = 0.f, M9 = 0.f;
#if D3GA_TILE_PRIO
        {   // (the priority is an immediate)
            const int left = ngroups - g;
            if (left > 8) __builtin_amdgcn_s_setprio(3);
            else if (left > 4) __builtin_amdgcn_s_setprio(2);
            else if (left > 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
#pragma unroll 1
        for (int ky = 0; ky < (D3GA_SCAN_ABL == 8 ? 0 : 4); ++ky) {
            const float *const pixq = pixrow + ky * 4 * PIXF;
            float4 pa[4], pb[4], pc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pa[k] = *reinterpret_cast<const float4 *>(pixq + k * PIXF);
                pb[k] = *reinterpret_cast<const float4 *>(pixq + k * PIXF + 4);
                if constexpr (DUAL) pc[k] = *reinterpret_cast<const float4 *>(pixq + k * PIXF + 8);
                // the whole record is loaded HERE: left alone the compiler sinks the load of T_final (A.bg . g) into a
                // divergent region behind `valid` -- an LDS round trip in the middle of every block line
                asm volatile("" : "+v"(pb[k].x), "+v"(pb[k].y), "+v"(pb[k].z));
                if constexpr (INVD) asm volatile("" : "+v"(pb[k].w));
            }
            const float dy = eyr - (float)ky;
            const float tb = cq.b * dy, tc = (cq.c * dy) * dy;
            float al[4], G[4], r[4], u[4], cgv[4], dx[4];
            bool valid[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dx[k] = exr - (float)k;
                bool ok;
                splat_eval_q(dx[k], tb, tc, cq.a, e.co.w, al[k], G[k], ok);
                valid[k] = ok & (e.pos <= __float_as_uint(pb[k].z));     // (a lane without an entry has opacity 0: never ok)
#ifdef D3GA_DIAG_COUNTERS
                dg.valid += valid[k] ? 1 : 0;          // lane efficiency: valid (entry, pixel) pairs / issued lane slots
#endif
                al[k] = valid[k] ? al[k] : 0.f;
                r[k] = __builtin_amdgcn_rcpf(1.0f - al[k]);
                cgv[k] = e.rgb.x * pa[k].z + e.rgb.y * pa[k].w + e.rgb.z * pb[k].x;
                if constexpr (DUAL) cgv[k] += e.c2r * pc[k].x + e.c2g * pc[k].y + e.c2b * pc[k].z;
                if constexpr (INVD) cgv[k] = fmaf(e.rgb.w, pb[k].w, cgv[k]);
            }
            float p0 = r[0], p1 = r[1], p2 = r[2], p3 = r[3];
            row_scan_mul4<R>(p0, p1, p2, p3);
            const float Ti[4] = {pa[0].x * p0, pa[1].x * p1, pa[2].x * p2, pa[3].x * p3};
            float dch[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { dch[k] = al[k] * Ti[k]; u[k] = cgv[k] * dch[k]; }
            float s0 = u[0], s1 = u[1], s2 = u[2], s3 = u[3];
            row_scan_add4<R>(s0, s1, s2, s3);
            const float Sin[4] = {s0 + pa[0].y, s1 + pa[1].y, s2 + pa[2].y, s3 + pa[3].y};
            float *const wq = wr_base + ky * 4 * PIXF;
            float gop[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dLda = Ti[k] * cgv[k] - (Sin[k] - u[k] + pb[k].y) * r[k];
                gop[k] = valid[k] ? G[k] * dLda : 0.f;
                M6 += dch[k] * pa[k].z; M7 += dch[k] * pa[k].w; M8 += dch[k] * pb[k].x;
                if constexpr (INVD) M9 = fmaf(dch[k], pb[k].w, M9);
                *reinterpret_cast<float2 *>(wq + k * PIXF) = make_float2(Ti[k], Sin[k]);
            }
            const float A = (gop[0] + gop[1]) + (gop[2] + gop[3]);
            const float B = fmaf(3.0f, gop[3], fmaf(2.0f, gop[2], gop[1]));
            const float C = fmaf(9.0f, gop[3], fmaf(4.0f, gop[2], gop[1]));
            SA += A; SB += B; SC += C;
            SyA = fmaf(dy, A, SyA); SyB = fmaf(dy, B, SyB); SyyA = fmaf(dy * dy, A, SyyA);
        }
        const float M5 = SA, ow = e.co.w;
        const float M0 = ow * (exr * SA - SB), M1 = ow * SyA;
        const float M2 = ow * (exr * (exr * SA - 2.0f * SB) + SC), M3 = ow * (exr * SyA - SyB), M4 = ow * SyyA;
        const float v0 = -(e.co.x * M0 + e.co.y * M1) * ddelx_dx;
        const float v1 = -(e.co.z * M1 + e.co.y * M0) * ddely_dy;
        const float4 va = make_float4(v0, v1, -0.5f * M2, -0.5f * M3), vb = make_float4(-0.5f * M4, M5, M6, M7);
        scan_consume<DUAL>(nxt, pg2);
        __builtin_amdgcn_wave_barrier();
        // ---- merge into the tile's cache (see the header) ----
        // One LDS round trip per attempt: the exchange that takes the slot and the (speculative) loads of its record
        // are issued together -- LDS instructions of a wavefront execute in order, so the loads see the record as the
        // lock holder owns it; a loser discards them.  Stores + the releasing tag store likewise need no wait.
        const uint32_t saddr = (uint32_t)(uintptr_t)(lds_u32 *)(s_cache + ((e.pos - 1u) & (uint32_t)(S - 1)) * kSlot);
        // an entry that touched no pixel has nothing but (exact) zeros: it stays out of the cache
        const uint32_t anybits = (__float_as_uint(SA) | __float_as_uint(SB) | __float_as_uint(SC)) | (__float_as_uint(SyA) | __float_as_uint(SyB) | __float_as_uint(SyyA)) |
                                 (__float_as_uint(M6) | __float_as_uint(M7) | __float_as_uint(M8) | __float_as_uint(M9));
        bool pending = D3GA_SCAN_ABL == 1 ? (anybits == 0x12345u) : (D3GA_SCAN_ABL == 8 ? act : (anybits << 1) != 0u);
#ifdef D3GA_DIAG_TIMELINE
        dg.groups += 1;
#endif
#ifdef D3GA_DIAG_COUNTERS
        dg.entries += act ? 1 : 0;
        dg.rowgroups += (act && lseg == 0) ? 1 : 0;
#endif
        while (__builtin_amdgcn_ballot_w64(pending) != 0ull) {
#ifdef D3GA_DIAG_TIMELINE
            dg.trips += 1;
#endif
            // straight-line body (selects, no nested divergent regions and no state carried around the loop: the first
            // version's phi copies and mask bookkeeping cost ~150 VALU instructions per attempt, PMC-measured)
            uint32_t old;
            f4 ca, cb;
            u2 cc;
            asm("" : "=v"(old), "=v"(ca), "=v"(cb), "=v"(cc));       // defined, arbitrary: lanes that are not pending never look
            float c9 = 0.f;
            if constexpr (INVD) {
                asm("" : "=v"(c9));
                if (pending)
                    asm volatile("ds_wrxchg_rtn_b32 %0, %5, %6 offset:40\n"
                                 "ds_read_b128 %1, %5\n"
                                 "ds_read_b128 %2, %5 offset:16\n"
                                 "ds_read_b64 %3, %5 offset:32\n"
                                 "ds_read_b32 %4, %5 offset:44\n"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "+v"(old), "+v"(ca), "+v"(cb), "+v"(cc), "+v"(c9)
                                 : "v"(saddr), "v"(kLocked)
                                 : "memory");
            } else
            if (pending)
                asm volatile("ds_wrxchg_rtn_b32 %0, %4, %5 offset:40\n"
                             "ds_read_b128 %1, %4\n"
                             "ds_read_b128 %2, %4 offset:16\n"
                             "ds_read_b64 %3, %4 offset:32\n"
                             "s_waitcnt lgkmcnt(0)"
                             : "+v"(old), "+v"(ca), "+v"(cb), "+v"(cc)
                             : "v"(saddr), "v"(kLocked)
                             : "memory");
            const bool won = pending && old != kLocked;
            const bool hit = won && old == e.pos;
            const bool evi = won && !hit && old != 0u;
#ifdef D3GA_DIAG_COUNTERS
            if (hit) dg.hit += 1; else if (evi) dg.evict += 1; else if (won) dg.install += 1;
#endif
            const f4 ta = {va.x + (hit ? ca.x : 0.f), va.y + (hit ? ca.y : 0.f), va.z + (hit ? ca.z : 0.f), va.w + (hit ? ca.w : 0.f)};
            const f4 tb4 = {vb.x + (hit ? cb.x : 0.f), vb.y + (hit ? cb.y : 0.f), vb.z + (hit ? cb.z : 0.f), vb.w + (hit ? cb.w : 0.f)};
            const u2 tc = {__float_as_uint(M8 + (hit ? __uint_as_float(cc.x) : 0.f)), e.gid};
            if constexpr (INVD) {
                const float t9 = M9 + (hit ? c9 : 0.f);
                if (won) asm volatile("ds_write_b32 %0, %1 offset:44" : : "v"(saddr), "v"(t9) : "memory");      // (before the tag store below releases the slot)
            }
            if (won)
                asm volatile("ds_write_b128 %0, %1\n"
                             "ds_write_b128 %0, %2 offset:16\n"
                             "ds_write_b64 %0, %3 offset:32\n"
                             "ds_write_b32 %0, %4 offset:40"
                             :
                             : "v"(saddr), "v"(ta), "v"(tb4), "v"(tc), "v"(e.pos)
                             : "memory");
            pending = pending && !won;
            const unsigned long long em = __builtin_amdgcn_ballot_w64(evi);
            if (em != 0ull) {
                // wave-uniform and rare (the window of live positions exceeded S): the displaced records leave through the
                // dump area, up to 24 at a time, nine consecutive lanes per record like every publish of this kernel
                const int rank = lanes_below(em), total = (int)__popcll(em);
                for (int c0 = 0; c0 < total; c0 += 20) {
                    if (evi && rank >= c0 && rank < c0 + 20) {
                        float *st = s_dump + (rank - c0) * 12;
                        st[0] = ca.x; st[1] = ca.y; st[2] = ca.z; st[3] = ca.w;
                        st[4] = cb.x; st[5] = cb.y; st[6] = cb.z; st[7] = cb.w;
                        st[8] = __uint_as_float(cc.x); st[9] = c9; st[10] = __uint_as_float(cc.y);      // value 9: dL/d(1/depth) (INVD), word 10: Gaussian id
                    }
                    const int n = min(total - c0, 20);
                    __builtin_amdgcn_wave_barrier();
                    for (int base = 0; base < n; base += kPerInst) {
                        const int ent = base + fq;
                        if (fq < kPerInst && ent < n) {
                            const float val = s_dump[ent * 12 + fk];
                            const uint32_t og = __float_as_uint(s_dump[ent * 12 + 10]);
                            if (D3GA_SCAN_ABL != 13 && val != 0.f) atomicAdd(A.acc + kAccStride * (size_t)og + fk_off, val);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        e = nxt;
        pg1 = pg2;
    }
}

// DUAL: a second image rendered with the same alphas (render_pair); S: slots of the tile's merge cache (see the file header).
// INVD: the inverse-depth image of branch dr_aa takes part as a fourth channel (its "colour" is the entry's 1 / depth): the
// incoming dL/dinvdepth joins c . g, and the entry's dL/d(1/depth) = sum alpha T dL/dinvdepth is a tenth accumulated value
// (the pad word of the cache slot, float 10 of the accumulator record).
// R: rows per block (D3GA_BWD_ROWS).  1: a wavefront's four rows walk four blocks, 16 entries per group and block, four
// wavefronts per tile.  2 / 4: 8 / 16 wavefronts per tile, a block's list runs over 2 / 4 rows (32 / 64 entries per group), a
// wavefront takes 2 / 1 blocks -- the same pixel steps in total, but the tile's longest list is walked in half / a quarter of
// the groups: the critical path of a heavy tile (18 groups of one wavefront at C3) shrinks with it.
template <bool DUAL, int S, bool INVD = false, int R = 1>
__global__ __launch_bounds__(256 * R, (DUAL ? (D3GA_TILE_WAVES < 3 ? D3GA_TILE_WAVES : 3) : D3GA_TILE_WAVES)) void composite_bwd_tile_kernel(   // (DUAL: 12-float pixel records -- the LDS of three workgroups per CU)
   
    int W, int H, int gx, int gy, int gyv, const uint32_t *__restrict__ tile_start, uint64_t dcap, const float2 *__restrict__ xy,
    const float4 *__restrict__ conic_o, const float4 *__restrict__ rgb_invd, const float *__restrict__ bg,
    const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix,
    float *__restrict__ acc, const uint32_t *__restrict__ tile_order, const float *__restrict__ colors2,
    const float *__restrict__ bg2, const float *__restrict__ dL_dpix2, const uint2 *__restrict__ blk_list,
    const uint32_t *__restrict__ blk_count, int assign, L1Source l1, const float *__restrict__ dL_dinvd, int split_cap, const uint32_t *__restrict__ split_cnt, int P) {
    static_assert((S & (S - 1)) == 0, "power of two");
    constexpr int NW = 4 * R;                        // wavefronts per tile (three, the third walking two sets of blocks: measured, slower -- DESIGN.md sec. 4)
    constexpr int SEG = 4 / R, LW = 16 * R;          // blocks per wavefront, lanes per block
    constexpr int PIXF = DUAL ? 12 : 8;
    constexpr int ROWF = 16 * PIXF + 4;
    constexpr int kSlot = 12;                        // dwords per cache slot
    const bool early_exit_off = (assign & 8) != 0;         // A/B: bit 3 of D3GA_TILE_ASSIGN keeps every wavefront until the tile is done
    assign &= 7;
    const int tiles = gx * gy;
    // split_h (D3GA_BWD_SPLIT, R = 1 only, needs the work order): each of the split_h heaviest tiles gets TWO workgroups -- the
    // even / odd ranks of its 16 blocks by length -- whose wavefronts take two blocks each over two rows (the R = 2 walk: 32
    // entries per group): the tile's longest list is walked in half the groups, at the price of a merge cache per half
    // The grid has room for split_cap of them; how many there are this frame is the order kernel's count (split_cnt; null: split_cap).
    int rank = (int)blockIdx.x, half = -1;
    if (R == 1 && split_cap > 0) {
        const int split_h = split_cnt ? min((int)*split_cnt, split_cap) : split_cap;
        if (rank < 2 * split_h) { half = rank & 1; rank >>= 1; } else rank -= split_h;
    }
    const int tile = tile_order ? (rank < tiles ? (int)tile_order[rank] : -1) : (rank < tiles ? rank : -1);
    if (tile < 0) return;                                  // uniform over the workgroup
    const uint32_t begin = (uint32_t)min((uint64_t)tile_start[tile], dcap);
    const uint32_t end = (uint32_t)min((uint64_t)tile_start[tile + 1], dcap);
    if (begin >= end) return;                              // uniform: empty tile

    __shared__ __attribute__((aligned(16))) float s_pix_all[NW][SEG * ROWF];
    __shared__ __attribute__((aligned(16))) float s_dump_all[NW][64 * 2 + 16 * PIXF];
    __shared__ __attribute__((aligned(16))) uint32_t s_cache[S * kSlot];
    // slot = (list position - 1) mod S: a list of n < S entries uses the first n slots only (init and publish stop there: a
    // third of the publish at C4, where lists average 340 entries)
    const int nslots = min(S, (int)(end - begin));
    for (int i = threadIdx.x; i < nslots; i += 64 * NW) s_cache[i * kSlot + 10] = 0u;   // tags: every slot empty
    __shared__ uint32_t s_arrived;                          // wavefronts of this tile that are done (the last one publishes the cache)
    if (threadIdx.x == 0) s_arrived = 0u;
    __shared__ uint8_t s_perm[16];                          // assign 2: the tile's 16 blocks by descending list length
    if (R > 1 || half >= 0) assign = 2;                     // (the quadrant / interleaved assignments exist for R = 1 only)
    if (assign == 2 && threadIdx.x < 16) {
        const int b = threadIdx.x, tx0 = (tile % gx) * kTile, ty0 = ((tile / gx) % gyv) * kTile;
        auto count_of = [&](int j) -> uint32_t {            // (the forward writes the counts of quadrants that start inside the image)
            const int q = j >> 2;
            return (tx0 + ((q & 1) << 3) < W && ty0 + ((q >> 1) << 3) < H) ? blk_count[16 * (size_t)tile + j] : 0u;
        };
        const uint32_t mine = count_of(b);
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t c = count_of(j);
            rank += (c > mine || (c == mine && j < b)) ? 1 : 0;
        }
        s_perm[rank] = (uint8_t)b;
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // block -> (wavefront, row).  assign 0: a wavefront takes the 2x2 blocks of one quadrant (the forward's grouping);
    // assign 1: it takes the blocks (wave & 1) + 2 i, (wave >> 1) + 2 j -- 8 pixels apart, so its rows rarely hold the same
    // Gaussian in the same group (copies that meet in ONE instruction serialise on the slot; copies in different wavefronts
    // simply hit the cache).  The forward numbers a tile's blocks 4 * quadrant + (block within the quadrant).
    // assign 2 (default): the four blocks of SIMILAR list length -- ranks 4 g .. 4 g + 3 of the tile's blocks sorted by length,
    // g = (wave + workgroup) mod 4.  A wavefront runs as many groups as its longest row needs, so rows of unequal length pad
    // (quadrants: 57.5 k wave-groups at C3, interleaved 59.3 k, against 49 k row-groups / 4); and rotating g with the workgroup
    // index gives every SIMD of a CU (wave w of a workgroup runs on SIMD w) one wavefront of each weight class.
    const int row = lane / LW;                             // which of the wavefront's blocks
    constexpr int kVals = INVD ? 10 : 9, kPerInst = 64 / kVals;           // publish: values per record, records per instruction
    const int fq = lane / kVals, fk = lane - kVals * fq;   // lane -> (record within a group of 7 (6), value)
    const int fk_off = fk < 2 ? fk : fk + 1;               // acc layout 0,1 | 3,4,5 | 6 | 7,8,9 | 10 (dL/d(1/depth), INVD)
    constexpr int kAccStride = D3GA_ACC_STRIDE;
    float *const s_pix = s_pix_all[wave];
    float *const s_dump = s_dump_all[wave];
    BwdDiag dg;
#ifdef D3GA_DIAG_TIMELINE
    const unsigned long long diag_t0 = __builtin_readcyclecounter(), diag_w0 = __builtin_amdgcn_s_memrealtime();
#endif

    {
        int blk;
        if (assign == 2) blk = s_perm[SEG * ((wave + (int)blockIdx.x) & (NW - 1)) + row];
        else {
            const int bx = assign ? 2 * (row & 1) + (wave & 1) : 2 * (wave & 1) + (row & 1);
            const int by = assign ? 2 * (row >> 1) + (wave >> 1) : 2 * (wave >> 1) + (row >> 1);
            blk = 4 * ((bx >> 1) + 2 * (by >> 1)) + ((bx & 1) + 2 * (by & 1));
        }
        const BwdArgs A = {W, H, gx, gyv, P, xy, conic_o, rgb_invd, bg, final_T, n_contrib, dL_dpix, acc, colors2, bg2, dL_dpix2, blk_list, blk_count, l1, dL_dinvd};
        if (R == 1 && half >= 0) {
            blk = s_perm[2 * (2 * ((wave + (int)blockIdx.x) & 3) + (lane >> 5)) + half];
            bwd_tile_wave<DUAL, S, INVD, (R == 1 ? 2 : R)>(A, tile, begin, end, blk, s_cache, s_pix, s_dump, lane, dg);
        } else
        bwd_tile_wave<DUAL, S, INVD, R>(A, tile, begin, end, blk, s_cache, s_pix, s_dump, lane, dg);
    }

#ifdef D3GA_DIAG_COUNTERS
    atomicAdd(&g_diag_scan[3], dg.install); atomicAdd(&g_diag_scan[4], dg.hit); atomicAdd(&g_diag_scan[5], dg.evict);
    atomicAdd(&g_diag_scan[8], dg.valid); atomicAdd(&g_diag_scan[9], dg.entries); atomicAdd(&g_diag_scan[10], dg.rowgroups);
#endif
#ifdef D3GA_DIAG_TIMELINE
    if (lane == 0 && dg.groups && blockIdx.x < 32768 / NW) {      // no atomics here: returning same-address atomics serialise at the memory side and would BE the timeline
        const unsigned long long diag_w1 = __builtin_amdgcn_s_memrealtime();
        const size_t slot = NW * (size_t)blockIdx.x + wave;
        g_diag_waves[4 * slot] = diag_w0 | ((__builtin_readcyclecounter() - diag_t0) << 40);
        g_diag_waves[4 * slot + 1] = diag_w1;
        g_diag_waves[4 * slot + 2] = dg.groups | (dg.trips << 16) | ((unsigned long long)(end - begin) << 32);
        g_diag_waves[4 * slot + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)(__builtin_amdgcn_s_getreg(6164) & 15) << 32);
    }
#endif
    // Publish.  A wavefront that is done LEAVES (its registers and wave slot go to the next workgroup: a tile's four
    // wavefronts take 5..18 groups, and with a closing barrier all four slots stayed occupied until the slowest was done --
    // the second dispatch round of the launch then started at half of the kernel's span, tools/diag_scan.py); the LAST one
    // to arrive (one returning integer LDS atomic per wavefront; LDS operations of a wavefront execute in order, so every
    // insert of a wavefront is complete when its arrival is counted) sends the whole cache to HBM, nine consecutive lanes
    // per record, seven records per instruction.
    if (D3GA_SCAN_ABL == 13 || D3GA_SCAN_ABL == 1) return;
    uint32_t arrived = 0u;
    // acq_rel at workgroup scope: this wavefront's cache stores are ordered BEFORE its arrival, and the publisher's reads of the
    // cache AFTER it has seen the others' arrivals -- by the memory model, not only by in-order LDS issue
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // every lane's cache stores (not only lane 0's) before the arrival
    if (lane == 0) arrived = __hip_atomic_fetch_add(&s_arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
    if (arrived != (uint32_t)(NW - 1) && !early_exit_off) return;
    if (early_exit_off) { __syncthreads(); if (wave != 0) return; }
    for (int base = 0; base < nslots; base += kPerInst) {
        const int ent = base + min(fq, kPerInst - 1);
        const bool mine = fq < kPerInst && ent < nslots;
        const uint32_t *const sl = s_cache + min(ent, S - 1) * kSlot;
        const uint32_t tag = sl[10], gid = sl[9];
        const float val = __uint_as_float(sl[fk < 9 ? fk : 11]);
        if (mine && tag != 0u && val != 0.f) atomicAdd(acc + kAccStride * (size_t)gid + fk_off, val);
    }
}

#ifdef D3GA_DIAG
extern "C" __attribute__((visibility("default"))) int d3ga_diag_scan_read(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_diag_scan), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag_scan), z, sizeof(z)) != hipSuccess) return 1;
        void *w = nullptr;
        if (hipGetSymbolAddress(&w, HIP_SYMBOL(g_diag_waves)) != hipSuccess || hipMemset(w, 0, sizeof(unsigned long long) * 4 * 32768) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int d3ga_diag_scan_waves(unsigned long long *out, int n) {      // n <= 32768 records of 4 words
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_waves), sizeof(unsigned long long) * 4 * (size_t)n) != hipSuccess;
}
#endif

int launch_composite_bwd_scan(const d3ga_raster_params *prm, int gx, int gy, const BinBuf &bin, const GeomBuf &g,
                              const ImgBuf &im, int64_t d_capacity, const float *bg, const float *dL_dpix, float *acc,
                              bool ordered, const float *colors2, const float *bg2, const float *dL_dpix2, const L1Source &l1,
                              hipStream_t s, const float *dL_dinvd) {
    // workgroup per tile, heaviest tiles first (tile_order of the bin stage); S = slots of the tile's merge cache
    const uint32_t *order = ordered ? (const uint32_t *)bin.tile_order : (const uint32_t *)nullptr;
    // split of the heaviest tiles (see the kernel): D3GA_BWD_SPLIT < 0 (default): the order kernel's count (a tenth of the non-empty
    // tiles, D3GA_CNT_HEAVY), read by every workgroup and capped by the grid's room for it; >= 0: that many (0: none)
    const int split_knob = composite_bwd_split();
    const int split_cap = !ordered ? 0 : (split_knob < 0 ? (gx * gy + 9) / 10 : min(split_knob, gx * gy));
    const uint32_t *split_cnt = (ordered && split_knob < 0) ? (const uint32_t *)(bin.counters + D3GA_CNT_HEAVY) : (const uint32_t *)nullptr;
    const dim3 tgrid(gx * gy + split_cap);
    const int S = composite_merge_slots();
#define D3GA_LAUNCH_TILE_R(DUALV, SV, INVDV, RV)                                                                              \
    hipLaunchKernelGGL((composite_bwd_tile_kernel<DUALV, SV, INVDV, RV>), tgrid, dim3(256 * RV),                                \
                       0, s, prm->W, prm->H, gx, gy, gy / n_views_of(prm), bin.tile_start, \
                       (uint64_t)d_capacity, reinterpret_cast<const float2 *>(g.xyh), g.conic_o, g.rgb_invd, bg, im.final_T, im.n_contrib, dL_dpix, acc, order,   \
                       colors2, bg2, dL_dpix2, (const uint2 *)im.blk_list, (const uint32_t *)im.blk_count, composite_tile_assign(), l1, dL_dinvd, split_cap, split_cnt, prm->P)
#define D3GA_LAUNCH_TILE(DUALV, SV, INVDV)                                                                                    \
    D3GA_LAUNCH_TILE_R(DUALV, SV, INVDV, 1)
    if (dL_dinvd) {                                          // inverse-depth gradient (branch dr_aa): single-image launches only
        if (colors2) return D3GA_E_CONFIG;
        if (S >= 512) D3GA_LAUNCH_TILE(false, 512, true); else D3GA_LAUNCH_TILE(false, 256, true);
    } else if (colors2) { if (S >= 512) D3GA_LAUNCH_TILE(true, 512, false); else D3GA_LAUNCH_TILE(true, 256, false); }
    else if (S >= 512) D3GA_LAUNCH_TILE(false, 512, false);
    else D3GA_LAUNCH_TILE(false, 256, false);
#undef D3GA_LAUNCH_TILE
#undef D3GA_LAUNCH_TILE_R
    return check_launch(s, prm->debug);
}

// test hook for the two row scans (d3ga_selftest_row_scan): lane l of every 16-lane row ends up with the inclusive sum /
// product of lanes 0..l of its row; the four chains of one call carry x, 2x, 3x, 4x (sums) and 1+x/8, 1+x/4, ... (products)
__global__ void row_scan_selftest_kernel(const float *__restrict__ in, float *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = in[i];
    float a0 = x, a1 = 2.f * x, a2 = 3.f * x, a3 = 4.f * x;
    row_scan_add4(a0, a1, a2, a3);
    float m0 = 1.f + 0.125f * x, m1 = 1.f + 0.25f * x, m2 = 1.f + 0.375f * x, m3 = 1.f + 0.5f * x;
    row_scan_mul4(m0, m1, m2, m3);
    float *o = out + 8 * (size_t)i;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = m0; o[5] = m1; o[6] = m2; o[7] = m3;
}

}  // namespace d3ga

extern "C" int d3ga_selftest_row_scan(int n, const float *in, float *out, d3ga_stream_t stream) {
    if (n <= 0 || (n % 256) != 0) return D3GA_E_SIZE;
    if (!in || !out) return D3GA_E_NULL;
    hipLaunchKernelGGL(d3ga::row_scan_selftest_kernel, dim3(n / 256), dim3(256), 0, (hipStream_t)stream, in, out, n);
    return d3ga::check_launch((hipStream_t)stream, 0);
}

// raster_composite_scan.hip -- compositing backward, entry-per-lane ("scan") formulation.
//
// The round-1 backward gave every LANE a pixel and every ITERATION a list entry, like the forward: the nine partial
// derivatives of that entry then had to be summed across the 16 lanes of the row (32 DPP instructions per iteration) and
// meet in an LDS accumulator, and only ~8 of the 16 lanes of a row are inside the entry's footprint.
// Here the roles are swapped.  A 16-lane DPP row still owns one 4x4 pixel block and walks that block's culled list (which
// the FORWARD wrote: ImgBuf::blk_list), but 16 consecutive list entries sit in the 16 lanes and the row steps through the
// block's 16 pixels:
//   * what couples the entries at one pixel is the transmittance T_i = T_final / prod_{j >= i}(1 - alpha_j) and the colour
//     blended behind entry i, S_i = sum_{j > i} (c_j . dL/dpixel) alpha_j T_j  -- two PREFIX SCANS over the lanes in
//     back-to-front order (4 row_shr DPP steps each), with the pixel's running (T, S) carried from group to group through
//     a 32-byte LDS record per pixel;
//   * dL/dalpha_i = T_i (c_i . g) - (S_i + T_final (bg . g)) / (1 - alpha_i)   -- algebraically upstream's recurrence
//     (accum_rec = last_alpha last_color + (1 - last_alpha) accum_rec), evaluated without the serial chain;
//   * every lane accumulates the nine moments of ITS entry over the 16 pixels in registers: no cross-lane reduction, no
//     LDS atomics.  After the 16 steps the per-entry constants are applied once (conic, -1/2, NDC scale) and the group is
//     published through an LDS transpose: nine consecutive lanes write one entry (36 contiguous bytes, two memory-side
//     atomic requests), seven entries per instruction.
//   * what bounds the kernel is the number of 64-byte accumulator lines it sends to the memory-side atomic units (DESIGN.md
//     sec. 4: with the arithmetic removed it takes as long), so the four rows are paced to reach the same depth together and
//     the copies of a Gaussian that meet in one flush are merged in LDS first (28 % of the entries at C3).
// Instructions per (entry, 4x4 block) pair: ~16 x 58 / 16 + ~7 = 65, against ~100 per (entry, block) visit before -- and a
// visit used to occupy a whole wavefront iteration in which on average 2.6 of the 4 rows had an entry at all.
// alpha is evaluated by the same splat_eval_q() as the forward and the validity test is the forward's (power <= 0,
// alpha >= 1/255, list position <= the pixel's n_contrib), so the set of (pixel, entry) pairs is exactly the forward's.
#include "composite_common.h"

// D3GA_SCAN_ABL: timing ablations of the kernel below (diagnostic builds only, results are WRONG; tools/gpu_ablate.sh):
//   1 no atomics | 6 one of the four pixel lines and no atomics | 7 plain stores instead of atomics |
//   8 every atomic but no pixel steps | 11 no merge of the rows' duplicates | 12 rows not paced
#ifndef D3GA_SCAN_ABL
#define D3GA_SCAN_ABL 0
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
__device__ __forceinline__ void row_scan_mul4(float &a, float &b, float &c, float &d) {
    asm("s_nop 1\n" D3GA_SCAN4("v_mul_f32_dpp", "", 1) D3GA_SCAN4("v_mul_f32_dpp", "", 2) D3GA_SCAN4("v_mul_f32_dpp", "", 4)
        D3GA_SCAN4("v_mul_f32_dpp", "", 8)
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void row_scan_add4(float &a, float &b, float &c, float &d) {
    asm("s_nop 1\n" D3GA_SCAN4("v_add_f32_dpp", " bound_ctrl:1", 1) D3GA_SCAN4("v_add_f32_dpp", " bound_ctrl:1", 2)
        D3GA_SCAN4("v_add_f32_dpp", " bound_ctrl:1", 4) D3GA_SCAN4("v_add_f32_dpp", " bound_ctrl:1", 8)
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
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
                                                 const float4 *__restrict__ rgb_invd, const float *__restrict__ colors2) {
    ScanEntry e;
    e.pos = pg.x; e.gid = pg.y;
    e.xy = make_float2(0.f, 0.f);
    e.co = make_float4(0.f, 0.f, 0.f, 0.f);
    e.rgb = make_float4(0.f, 0.f, 0.f, 0.f);
    e.c2r = e.c2g = e.c2b = 0.f;
    if (pg.x != 0u) {
        e.xy = xy[pg.y]; e.co = conic_o[pg.y]; e.rgb = rgb_invd[pg.y];
        if constexpr (DUAL) {
            e.c2r = colors2[3 * (size_t)pg.y]; e.c2g = colors2[3 * (size_t)pg.y + 1]; e.c2b = colors2[3 * (size_t)pg.y + 2];
        }
    }
    return e;
}

#ifdef D3GA_DIAG
__device__ unsigned long long g_diag_scan[8];     // diagnostic build only (tools/diag_scan.py): loop statistics of the kernel below
__device__ unsigned long long g_diag_waves[32768 * 4];   // per active wave: s_memtime at start / end, groups, HW_ID | XCC_ID << 32
#endif
// forces the compiler's s_waitcnt for these registers HERE (an empty asm that reads them)
template <bool DUAL>
__device__ __forceinline__ void scan_consume(ScanEntry &e, uint2 &pg) {
    asm volatile("" : "+v"(e.xy.x), "+v"(e.xy.y), "+v"(e.co.x), "+v"(e.co.y), "+v"(e.co.z), "+v"(e.co.w));
    asm volatile("" : "+v"(e.rgb.x), "+v"(e.rgb.y), "+v"(e.rgb.z), "+v"(pg.x), "+v"(pg.y));
    if constexpr (DUAL) asm volatile("" : "+v"(e.c2r), "+v"(e.c2g), "+v"(e.c2b));
}

constexpr int kStageStride = 12;     // floats per entry in the flush staging area: nine values, the id, the list position, a pad
constexpr uint32_t kNoGaussian = 0xffffffffu;   // id of a staged entry that was merged into a lower row's

template <bool DUAL>
__global__ __launch_bounds__(64, 4) void composite_bwd_scan_kernel(
    int W, int H, int gx, int gy, const uint32_t *__restrict__ tile_start, uint64_t dcap, const float2 *__restrict__ xy,
    const float4 *__restrict__ conic_o, const float4 *__restrict__ rgb_invd, const float *__restrict__ bg,
    const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix,
    float *__restrict__ acc, const uint32_t *__restrict__ tile_order, const float *__restrict__ colors2,
    const float *__restrict__ bg2, const float *__restrict__ dL_dpix2, const uint2 *__restrict__ blk_list,
    const uint32_t *__restrict__ blk_count) {
    constexpr int PIXF = DUAL ? 12 : 8;              // floats per pixel record
    constexpr int ROWF = 16 * PIXF + 4;              // floats per row of records (+16 B: the four rows start in different banks)
    const Quad q = tile_order ? quad_of_block_ordered(gx, gx * gy, tile_order) : quad_of_block(gx, gy);
    if (!q.valid || q.qx0 >= W || q.qy0 >= H) return;     // wave-uniform (the forward wrote blk_count for exactly these)
    const int lane = threadIdx.x & 63;
    const RowGeom rg = row_geom(q, lane);
    const bool inside = rg.px < W && rg.py < H;
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
    float h0 = 0.f, h1 = 0.f, h2 = 0.f;
    if constexpr (DUAL) {
        if (inside) { h0 = dL_dpix2[pid]; h1 = dL_dpix2[hw + pid]; h2 = dL_dpix2[2 * hw + pid]; }
        bg_dot += bg2[0] * h0 + bg2[1] * h1 + bg2[2] * h2;
    }
    const uint32_t maxlast = wave_max_u32(last);
    if (maxlast == 0) return;

    __shared__ __attribute__((aligned(16))) float s_pix[4 * ROWF];
    __shared__ __attribute__((aligned(16))) float s_stage[64 * kStageStride];
    __shared__ __attribute__((aligned(16))) float s_dump[64 * 2 + 16 * PIXF];   // where lanes 0..14 of a row "write" the carries
    const int l16 = lane & 15;
    float *const pixrow = s_pix + rg.row * ROWF;             // this row's 16 pixel records
    float *const wr_base = l16 == 15 ? pixrow : s_dump + 2 * lane;
    {   // pixel l16 of the block: running transmittance, running colour behind, dL/dpixel, T_final (bg . g), n_contrib
        float *rec = pixrow + l16 * PIXF;
        *reinterpret_cast<float4 *>(rec) = make_float4(T_final, 0.f, g0, g1);
        *reinterpret_cast<float4 *>(rec + 4) = make_float4(g2, T_final * bg_dot, __uint_as_float(last), 0.f);
        if constexpr (DUAL) *reinterpret_cast<float4 *>(rec + 8) = make_float4(h0, h1, h2, 0.f);
    }
    const uint32_t blk_cap = end - begin;
    const uint32_t cnt = blk_count[16 * (size_t)q.tile + 4 * q.quad + rg.row];
    const uint2 *const list = blk_list + 16 * (size_t)begin + (size_t)(4 * q.quad + rg.row) * blk_cap;
    const int ngroups = (int)((wave_max_u32(cnt) + 15u) >> 4);
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    const float bxr = (float)(q.qx0 + ((rg.row & 1) << 2)), byr = (float)(q.qy0 + ((rg.row >> 1) << 2));   // block origin
    const int fq = lane / 9, fk = lane - 9 * fq;           // flush: lane -> (entry within a group of 7, value)
    const int fk_off = fk < 2 ? fk : fk + 1;               // acc layout 0,1 | 3,4,5 | 6 | 7,8,9
    constexpr int kAccStride = D3GA_ACC_STRIDE;

    // Back to front, all four rows PACED TO FINISH TOGETHER: the wavefront runs as many groups as its longest list needs
    // anyway, so a shorter list hands out per = ceil(cnt / ngroups) <= 16 entries per group instead of 16 until it runs dry.
    // The four lists are subsets of one depth-ordered tile list; at equal fractions of their length they are at (nearly) the
    // same depth, so the copies of a Gaussian in different rows meet in the same flush, where they are merged (below).
    // Group g holds list entries cnt-1-per*g ... (per of them), lane l < per the entry cnt-1-per*g-l.
    // (unconditional load from a clamped index + select: a load under a branch makes the compiler copy the result into the
    // merge register right behind the load, i.e. wait for it on the spot)
    const int per = D3GA_SCAN_ABL == 12 ? 16 : (ngroups > 0 ? ((int)cnt + ngroups - 1) / ngroups : 0);
    auto list_entry = [&](int g) -> uint2 {
        const int idx = (int)cnt - 1 - per * g - l16;
        uint2 v = list[max(idx, 0)];
        v.x = (idx >= 0 && l16 < per) ? v.x : 0u;
        return v;
    };
#ifdef D3GA_DIAG_COUNTERS
    // per-wave record only (one returning atomic for the slot): shared counters would serialise the start of 5401 waves
    const unsigned long long diag_t0 = __builtin_readcyclecounter(), diag_w0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long diag_slot = 0, diag_rowgroups = 0, diag_entries = 0, diag_live = 0, diag_dups = 0;
    {
        const uint32_t c0 = __builtin_amdgcn_readlane(cnt, 0), c1 = __builtin_amdgcn_readlane(cnt, 16);
        const uint32_t c2 = __builtin_amdgcn_readlane(cnt, 32), c3 = __builtin_amdgcn_readlane(cnt, 48);
        diag_rowgroups = ((c0 + 15) >> 4) + ((c1 + 15) >> 4) + ((c2 + 15) >> 4) + ((c3 + 15) >> 4);
        diag_entries = c0 + c1 + c2 + c3;
        if (lane == 0) diag_slot = atomicAdd(&g_diag_scan[0], 1ull);
        diag_slot = __builtin_amdgcn_readfirstlane((unsigned)diag_slot);
    }
#endif
    // Software pipeline over the groups.  vmcnt is ONE in-order counter for loads and (fire-and-forget) atomics on gfx9, and the
    // number of atomic instructions a flush issues is data dependent, so a wait for a load that was issued before a flush
    // but is consumed after it degenerates to vmcnt(0): it waits for the flush's atomics to be acknowledged by the L2 (measured:
    // 35 % of all wave cycles in s_waitcnt).  Hence: the loads of group g+1 (list entry -> record) and the list entry of
    // group g+2 are issued at the TOP of iteration g and consumed (scan_consume) at the END of its 16 steps, BEFORE the flush.
    ScanEntry e = scan_gather<DUAL>(list_entry(0), xy, conic_o, rgb_invd, colors2);
    uint2 pg1 = list_entry(1);
    __builtin_amdgcn_wave_barrier();

    for (int g = 0; g < ngroups; ++g) {
        ScanEntry nxt = scan_gather<DUAL>(pg1, xy, conic_o, rgb_invd, colors2);    // group g+1: in flight during the 16 steps
        uint2 pg2 = list_entry(g + 2);
        const bool act = e.pos != 0u;
        const float exr = e.xy.x - bxr, eyr = e.xy.y - byr;                   // centre relative to the block origin
        const ConicQ cq = conic_q(e.co.x, e.co.y, e.co.z);
        float M0 = 0.f, M1 = 0.f, M2 = 0.f, M3 = 0.f, M4 = 0.f, M5 = 0.f, M6 = 0.f, M7 = 0.f, M8 = 0.f;
#pragma unroll 1
        for (int ky = 0; ky < (D3GA_SCAN_ABL == 6 ? 1 : (D3GA_SCAN_ABL == 8 ? 0 : 4)); ++ky) {
            // the four pixels of block line ky, side by side (independent until the carries are written back)
            const float *const pixq = pixrow + ky * 4 * PIXF;
            float4 pa[4], pb[4], pc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pa[k] = *reinterpret_cast<const float4 *>(pixq + k * PIXF);        // T, S, g0, g1
                pb[k] = *reinterpret_cast<const float4 *>(pixq + k * PIXF + 4);    // g2, T_final (bg . g), n_contrib
                if constexpr (DUAL) pc[k] = *reinterpret_cast<const float4 *>(pixq + k * PIXF + 8);
            }
            const float dy = eyr - (float)ky;
            const float tb = cq.b * dy, tc = (cq.c * dy) * dy;              // shared by the four pixels of the line
            float al[4], G[4], r[4], u[4], cgv[4], dx[4];
            bool valid[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dx[k] = exr - (float)k;
                bool ok;
                splat_eval_q(dx[k], tb, tc, cq.a, e.co.w, al[k], G[k], ok);
                valid[k] = ok & act & (e.pos <= __float_as_uint(pb[k].z));
                al[k] = valid[k] ? al[k] : 0.f;
                r[k] = __builtin_amdgcn_rcpf(1.0f - al[k]);
                cgv[k] = e.rgb.x * pa[k].z + e.rgb.y * pa[k].w + e.rgb.z * pb[k].x;     // c . dL/dpixel
                if constexpr (DUAL) cgv[k] += e.c2r * pc[k].x + e.c2g * pc[k].y + e.c2b * pc[k].z;
            }
            float p0 = r[0], p1 = r[1], p2 = r[2], p3 = r[3];
            row_scan_mul4(p0, p1, p2, p3);
            const float Ti[4] = {pa[0].x * p0, pa[1].x * p1, pa[2].x * p2, pa[3].x * p3};   // transmittance in front of the entry
            float dch[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { dch[k] = al[k] * Ti[k]; u[k] = cgv[k] * dch[k]; }
            float s0 = u[0], s1 = u[1], s2 = u[2], s3 = u[3];
            row_scan_add4(s0, s1, s2, s3);
            const float Sin[4] = {s0 + pa[0].y, s1 + pa[1].y, s2 + pa[2].y, s3 + pa[3].y};   // colour behind, this entry included
            float *const wq = wr_base + ky * 4 * PIXF;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dLda = Ti[k] * cgv[k] - (Sin[k] - u[k] + pb[k].y) * r[k];
                const float gop = valid[k] ? G[k] * dLda : 0.f;
                const float w = e.co.w * gop;                                 // the 0.99 clamp passes the gradient through
                const float wx = w * dx[k], wy = w * dy;
                M0 += wx; M1 += wy; M2 += wx * dx[k]; M3 += wx * dy; M4 += wy * dy;
                M5 += gop;
                M6 += dch[k] * pa[k].z; M7 += dch[k] * pa[k].w; M8 += dch[k] * pb[k].x;
                // carry to the next group: lane 15 of the row holds the totals; the other lanes write to a dump area
                *reinterpret_cast<float2 *>(wq + k * PIXF) = make_float2(Ti[k], Sin[k]);
            }
        }
        if (D3GA_SCAN_ABL == 8 && act) { M0 = M1 = M2 = M3 = M4 = M5 = M6 = M7 = M8 = 1.0f; }   // ablation: every atomic, no arithmetic
        // publish: per-entry constants, then nine consecutive lanes per entry
        const float v0 = -(e.co.x * M0 + e.co.y * M1) * ddelx_dx;
        const float v1 = -(e.co.z * M1 + e.co.y * M0) * ddely_dy;
        const float4 va = make_float4(v0, v1, -0.5f * M2, -0.5f * M3), vb = make_float4(-0.5f * M4, M5, M6, M7);
        {
            float *st = s_stage + lane * kStageStride;
            *reinterpret_cast<float4 *>(st) = va;
            *reinterpret_cast<float4 *>(st + 4) = vb;
            *reinterpret_cast<float4 *>(st + 8) = make_float4(M8, __uint_as_float(e.gid), __uint_as_float(e.pos), 0.f);
        }
        scan_consume<DUAL>(nxt, pg2);                      // the loads issued at the top have landed (see above)
        __builtin_amdgcn_wave_barrier();
#ifdef D3GA_DIAG_COUNTERS
        {   // how many of this flush's entries are the same Gaussian as an entry of a LOWER row (a merge would save their line)
            const uint32_t mygid = e.gid;
            bool dup = false;
            for (int o = 0; o < 16 * rg.row; ++o) dup = dup || (__float_as_uint(s_stage[o * kStageStride + 9]) == mygid && s_stage[o * kStageStride + 5] != 0.f);
            const bool live = act && M5 != 0.f;
            diag_live += __popcll(__ballot(live));
            diag_dups += __popcll(__ballot(live && dup));
        }
#endif
        if (D3GA_SCAN_ABL != 11) {
            // Merge the rows' copies of one Gaussian before they leave the CU.  What bounds this kernel is the number of 64-byte
            // accumulator lines it sends to the memory-side atomic units; the four rows walk subsets of ONE
            // depth-ordered tile list roughly in step, so 18 % of a flush's entries (C3) are a Gaussian that a lower row
            // publishes in the same flush.  Rows 1..3 look their position up in each lower row (binary search over the row's 16
            // staged positions: descending, 0 = no entry), add their nine values to the first match and retire their own entry.
            // All rows search at once (a search only reads positions); then one SOURCE row per phase adds its entries to their
            // targets (its lanes hold distinct Gaussians, so they hit distinct entries and a plain LDS read-add-write is safe).
            // The lower rows are searched in order, so a Gaussian ends up in the lowest row that has it.
            int found = -1;
#pragma unroll
            for (int tr = 0; tr < 3; ++tr) {
                if (rg.row > tr && act) {
                    const float *const trow = s_stage + (16 * tr) * kStageStride;
                    int j = 0;
                    uint32_t aj = __float_as_uint(trow[10]);
#pragma unroll
                    for (int sft = 8; sft >= 1; sft >>= 1) {
                        const uint32_t t = __float_as_uint(trow[(j + sft) * kStageStride + 10]);
                        if (t >= e.pos) { j += sft; aj = t; }
                    }
                    if (found < 0 && aj == e.pos) found = 16 * tr + j;
                }
            }
#pragma unroll
            for (int sr = 1; sr < 4; ++sr) {
                if (rg.row == sr && found >= 0) {
                    float *tg = s_stage + found * kStageStride;
                    float4 ta = *reinterpret_cast<float4 *>(tg), tb = *reinterpret_cast<float4 *>(tg + 4);
                    const float t8 = tg[8];
                    ta.x += va.x; ta.y += va.y; ta.z += va.z; ta.w += va.w;
                    tb.x += vb.x; tb.y += vb.y; tb.z += vb.z; tb.w += vb.w;
                    *reinterpret_cast<float4 *>(tg) = ta;
                    *reinterpret_cast<float4 *>(tg + 4) = tb;
                    tg[8] = t8 + M8;
                    s_stage[lane * kStageStride + 9] = __uint_as_float(kNoGaussian);       // the flush skips this entry
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        float fval[10];
        uint32_t fgid[10];
#pragma unroll
        for (int it = 0; it < 10; ++it) {                  // all twenty LDS reads in flight together
            const int ent = it * 7 + (it == 9 ? min(fq, 0) : min(fq, 6));      // 10 x 7 covers 64 with the last round holding one entry
            fval[it] = s_stage[ent * kStageStride + fk];
            fgid[it] = __float_as_uint(s_stage[ent * kStageStride + 9]);
        }
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            const bool mine = it == 9 ? fq == 0 : fq < 7;
            if (D3GA_SCAN_ABL != 1 && D3GA_SCAN_ABL != 6 && mine && fval[it] != 0.f && fgid[it] != kNoGaussian) {
                if (D3GA_SCAN_ABL == 7) acc[kAccStride * (size_t)fgid[it] + fk_off] = fval[it];          // plain store instead of the atomic
                else atomicAdd(acc + kAccStride * (size_t)fgid[it] + fk_off, fval[it]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        e = nxt;
        pg1 = pg2;
    }
#ifdef D3GA_DIAG_COUNTERS
    if (lane == 0) { atomicAdd(&g_diag_scan[1], diag_live); atomicAdd(&g_diag_scan[2], diag_dups); }
    if (lane == 0 && diag_slot < 32768) {
        g_diag_waves[4 * diag_slot] = diag_w0 | ((__builtin_readcyclecounter() - diag_t0) << 40);   // 100 MHz wall | s_memtime duration
        g_diag_waves[4 * diag_slot + 1] = __builtin_amdgcn_s_memrealtime();
        g_diag_waves[4 * diag_slot + 2] = (unsigned long long)ngroups | (diag_rowgroups << 16) | (diag_entries << 32);
        g_diag_waves[4 * diag_slot + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)(__builtin_amdgcn_s_getreg(6164) & 15) << 32);
    }
#endif
}

#ifdef D3GA_DIAG
extern "C" int d3ga_diag_scan_read(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_diag_scan), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag_scan), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int d3ga_diag_scan_waves(unsigned long long *out, int n) {      // n <= 32768 records of 4 words
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_waves), sizeof(unsigned long long) * 4 * (size_t)n) != hipSuccess;
}
#endif

int launch_composite_bwd_scan(const d3ga_raster_params *prm, int gx, int gy, const BinBuf &bin, const GeomBuf &g,
                              const ImgBuf &im, int64_t d_capacity, const float *bg, const float *dL_dpix, float *acc,
                              bool ordered, const float *colors2, const float *bg2, const float *dL_dpix2, hipStream_t s) {
    const dim3 grid(ordered ? quad_grid_ordered(gx * gy) : quad_grid(gx, gy));
    const uint32_t *order = ordered ? (const uint32_t *)bin.tile_order : (const uint32_t *)nullptr;
    if (colors2)
        hipLaunchKernelGGL(composite_bwd_scan_kernel<true>, grid, dim3(64), 0, s, prm->W, prm->H, gx, gy, bin.tile_start,
                           (uint64_t)d_capacity, g.xy, g.conic_o, g.rgb_invd, bg, im.final_T, im.n_contrib, dL_dpix, acc, order,
                           colors2, bg2, dL_dpix2, (const uint2 *)im.blk_list, (const uint32_t *)im.blk_count);
    else
        hipLaunchKernelGGL(composite_bwd_scan_kernel<false>, grid, dim3(64), 0, s, prm->W, prm->H, gx, gy, bin.tile_start,
                           (uint64_t)d_capacity, g.xy, g.conic_o, g.rgb_invd, bg, im.final_T, im.n_contrib, dL_dpix, acc, order,
                           (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const uint2 *)im.blk_list,
                           (const uint32_t *)im.blk_count);
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

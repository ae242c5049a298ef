// raster_bin.hip -- binning and depth ordering for gfx950.
//
// Instead of one global 64-bit radix sort over all (tile, depth) duplicates (6 passes x 2 x 12 B per duplicate
// of HBM traffic), the duplicates are first binned by tile with a counting sort (histogram in the preprocess
// kernel -> exclusive scan -> atomic-cursor scatter: 8 B written per duplicate), and each tile's list is then
// sorted by (depth, Gaussian index) entirely inside the LDS of one workgroup with a bitonic network
// (8 B read + 4 B written per duplicate).  The 64-bit key (depth bits << 32 | index) makes the order total, so
// the result does not depend on the order in which the atomics of the scatter pass landed.
//
//   tile_scan_order_kernel  two workgroups: [0] exclusive prefix of the tile histogram, D, overflow flag;
//                         [1] tile indices by descending list length (work-ordered dispatch)
//   tile_scatter_kernel   one thread per Gaussian: emit its key into every tile of its rectangle; slots are reserved
//                         per (block, tile) through an LDS window (d3ga_internal.h: TileWindow)
//   tile_sort_lds_kernel       one workgroup per tile: LDS bitonic sort, 8 keys per thread (lists up to 2048 entries)
//   tile_sort_lds_list_kernel  longer lists (up to 8192): 64 KB LDS, persistent grid over a device-side work list
//                              (+ in the same launch: even longer lists, same network on global memory)
#include <atomic>

#include "composite_common.h"

namespace d3ga {

constexpr int kScanBlock = 1024;

__device__ __forceinline__ void tile_scan_body(int tiles, const uint32_t *__restrict__ count,
                                               uint32_t *__restrict__ start, uint32_t *__restrict__ cursor,
                                               uint32_t *__restrict__ counters, uint64_t dcap,
                                               uint32_t *__restrict__ big_tiles, uint32_t *__restrict__ huge_tiles,
                                               uint32_t *__restrict__ mid_tiles, uint32_t n_small, uint32_t n_mid,
                                               uint32_t n_large) {
    // 8 CONSECUTIVE tiles per thread, moved with two 16-byte loads / stores per array (lane-contiguous: 2 KB per wavefront
    // and instruction), a wavefront scan + 16 LDS totals: ONE barrier per chunk of 8192 tiles.  (Rounds 2-3 staged the
    // counts and the results through LDS, three barriers per chunk: the scan workgroup was the 8-10 us of this launch.)
    constexpr int kPer = 8, kChunk = kPer * kScanBlock;
    __shared__ uint32_t s_wave[2][kScanBlock / 64];
    __shared__ uint32_t s_max;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_max = 0;
    uint32_t carry = 0, mx = 0;
    int par = 0;
    for (int c0 = 0; c0 < tiles; c0 += kChunk, par ^= 1) {
        const int t0 = c0 + kPer * tid;
        uint32_t c[kPer];
        if (t0 + kPer <= tiles) {
            const uint4 lo = reinterpret_cast<const uint4 *>(count + t0)[0], hi = reinterpret_cast<const uint4 *>(count + t0)[1];
            c[0] = lo.x; c[1] = lo.y; c[2] = lo.z; c[3] = lo.w; c[4] = hi.x; c[5] = hi.y; c[6] = hi.z; c[7] = hi.w;
        } else {
#pragma unroll
            for (int k = 0; k < kPer; ++k) c[k] = t0 + k < tiles ? count[t0 + k] : 0u;
        }
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) { sum += c[k]; mx = max(mx, c[k]); }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) s_wave[par][wave] = incl;
        __syncthreads();                                   // (two buffers: the next chunk's totals do not race this chunk's reads)
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kScanBlock / 64; ++w) {
            const uint32_t v = s_wave[par][w];
            wbase += w < wave ? v : 0u;
            total += v;
        }
        uint32_t run = carry + wbase + incl - sum;
        uint32_t e[kPer];
        int cls[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            e[k] = run;
            run += c[k];
            cls[k] = c[k] > n_large ? 2 : (c[k] > n_mid ? 1 : (c[k] > n_small ? 0 : -1));
        }
        // work lists for the long-list sort kernels (usually empty: those launches then cost one tiny grid).  Slots are reserved
        // per WAVEFRONT AND CLASS for all of a thread's 8 tiles at once: a wavefront scan of the per-lane counts and ONE returning
        // atomic.  (One same-address global atomic per tile made this kernel 79 us at 4K with 2M Gaussians, where thousands of
        // tiles are in the mid class; one per wavefront, class and k -- up to 24 dependent round trips per chunk -- 34 us.)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            uint32_t nq = 0;
#pragma unroll
            for (int k = 0; k < kPer; ++k) nq += cls[k] == q ? 1u : 0u;
            if (__ballot(nq != 0u) == 0ull) continue;                    // wave-uniform
            uint32_t inq = nq;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(inq, off);
                if (lane >= off) inq += v;
            }
            const uint32_t tot = (uint32_t)__shfl((int)inq, 63);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&counters[q == 0 ? D3GA_CNT_MID : (q == 1 ? D3GA_CNT_BIG : D3GA_CNT_HUGE)], tot);
            uint32_t pos = (uint32_t)__shfl((int)base, 0) + inq - nq;
            uint32_t *list = q == 0 ? mid_tiles : (q == 1 ? big_tiles : huge_tiles);
#pragma unroll
            for (int k = 0; k < kPer; ++k)
                if (cls[k] == q) list[pos++] = (uint32_t)(t0 + k);
        }
        if (t0 + kPer <= tiles) {
            const uint4 lo = make_uint4(e[0], e[1], e[2], e[3]), hi = make_uint4(e[4], e[5], e[6], e[7]);
            reinterpret_cast<uint4 *>(start + t0)[0] = lo; reinterpret_cast<uint4 *>(start + t0)[1] = hi;
            reinterpret_cast<uint4 *>(cursor + t0)[0] = lo; reinterpret_cast<uint4 *>(cursor + t0)[1] = hi;
        } else {
#pragma unroll
            for (int k = 0; k < kPer; ++k)
                if (t0 + k < tiles) { start[t0 + k] = e[k]; cursor[t0 + k] = e[k]; }
        }
        carry += total;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    if (lane == 0) atomicMax(&s_max, mx);
    __syncthreads();
    if (tid == 0) {
        start[tiles] = carry;
        counters[D3GA_CNT_D] = carry;
        counters[D3GA_CNT_OVERFLOW] = (uint64_t)carry > dcap ? 1u : 0u;
        counters[D3GA_CNT_MAXTILE] = s_max;
    }
}

// Work-ordered dispatch of the compositing kernels: tile indices by DESCENDING list length (bucket sort, 256 buckets
// scaled to the longest list; order inside a bucket is arbitrary).  All active compositing wavefronts are resident at once
// and the dispatcher deals workgroups round-robin, so dealing them in descending order of work gives every SIMD one
// wavefront from each work quantile instead of a random handful.
__device__ __forceinline__ void tile_order_body(int tiles, const uint32_t *__restrict__ count,
                                                uint32_t *__restrict__ order, uint32_t *__restrict__ counters) {
    // Round 3: ONE pass over the counts.  The bucket of a tile is its list length in units of 16 entries (the group size of
    // the compositing backward), clipped at 254 -- a fixed map, so the maximum need not be known first (round 2 took the
    // maximum, then built the histogram, then scattered: three dependent rounds of global loads on one workgroup, 10 us);
    // every thread keeps its (up to) 8 counts in registers between the histogram and the scatter.  Lists of 4064+ entries
    // share the first bucket: they are the heaviest tiles either way.
    __shared__ uint32_t s_hist[256];                      // buckets 0..254: non-empty tiles, longest lists first
    __shared__ uint32_t s_zero;                           // empty tiles (most of the image for an avatar): wavefront-
    const int tid = threadIdx.x, lane = tid & 63;         // aggregated, thousands of same-address LDS atomics serialise
    constexpr int kPer = 8;
    if (tid < 256) s_hist[tid] = 0;
    if (tid == 0) s_zero = 0;
    __syncthreads();
    auto bucket = [](uint32_t c) { return 254u - min(c >> 4, 254u); };
    for (int c0 = 0; c0 < tiles; c0 += kPer * kScanBlock) {               // (one trip up to 8192 tiles)
        uint32_t c[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int t = c0 + k * kScanBlock + tid;
            c[k] = t < tiles ? count[t] : 0xffffffffu;                    // 0xffffffff: no tile
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const bool zero = c[k] == 0u;
            const unsigned long long zm = __ballot(zero);
            if (lane == 0 && zm) atomicAdd(&s_zero, (uint32_t)__popcll(zm));
            if (c[k] != 0u && c[k] != 0xffffffffu) atomicAdd(&s_hist[bucket(c[k])], 1u);
        }
        if (c0 + kPer * kScanBlock < tiles) continue;                     // more than 8192 tiles: finish the histogram first
        // ---- the common case: everything in registers; exclusive prefix of the buckets, then the scatter ----
        __syncthreads();
        if (tid < 64) {
            uint32_t v[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = (4 * tid + k < 255) ? s_hist[4 * tid + k] : 0u; sum += v[k]; }
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t u = __shfl_up(incl, off);
                if (tid >= off) incl += u;
            }
            uint32_t run = incl - sum;
#pragma unroll
            for (int k = 0; k < 4; ++k) { s_hist[4 * tid + k] = run; run += v[k]; }   // [255] = number of non-empty tiles
        }
        __syncthreads();
        if (tid == 0) {
            s_zero = s_hist[255];                         // empty tiles follow the non-empty ones
            // the compositing backward gives the heaviest tenth of the non-empty tiles two workgroups each (raster_composite_scan.hip:
            // -5 % at C3 / C4, where the launch is 1.3 rounds of resident workgroups and ends when its longest lists do) -- unless
            // the launch is many rounds anyway (more than 4 x 1024 resident workgroups: measured +3 % at C5, whose tail is light tiles)
            counters[D3GA_CNT_HEAVY] = s_hist[255] <= 4096u ? (s_hist[255] + 9u) / 10u : 0u;
        }
        __syncthreads();
        for (int c1 = 0; c1 < tiles; c1 += kPer * kScanBlock) {
            if (c0 != 0) {                                // more than one chunk (> 8192 tiles): the registers hold one chunk only
#pragma unroll
                for (int k = 0; k < kPer; ++k) {
                    const int t = c1 + k * kScanBlock + tid;
                    c[k] = t < tiles ? count[t] : 0xffffffffu;
                }
            }
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int t = c1 + k * kScanBlock + tid;
                const bool zero = c[k] == 0u;
                const unsigned long long zm = __ballot(zero);
                uint32_t zbase = 0;
                if (lane == 0 && zm) zbase = atomicAdd(&s_zero, (uint32_t)__popcll(zm));
                zbase = (uint32_t)__shfl((int)zbase, 0);
                if (zero) order[zbase + (uint32_t)__popcll(zm & ((1ull << lane) - 1ull))] = (uint32_t)t;
                if (c[k] != 0u && c[k] != 0xffffffffu) order[atomicAdd(&s_hist[bucket(c[k])], 1u)] = (uint32_t)t;
            }
        }
        // (c0 was the last chunk: the loop ends)
    }
}

// one launch, two workgroups: block 0 scans the histogram, block 1 orders the tiles (they only share the read-only counts)
__global__ __launch_bounds__(kScanBlock) void tile_scan_order_kernel(int tiles, const uint32_t *__restrict__ count,
                                                                     uint32_t *__restrict__ start,
                                                                     uint32_t *__restrict__ cursor,
                                                                     uint32_t *__restrict__ counters, uint64_t dcap,
                                                                     uint32_t *__restrict__ big_tiles,
                                                                     uint32_t *__restrict__ huge_tiles,
                                                                     uint32_t *__restrict__ mid_tiles, uint32_t n_small,
                                                                     uint32_t n_mid, uint32_t n_large,
                                                                     uint32_t *__restrict__ order) {
    if (blockIdx.x == 0) tile_scan_body(tiles, count, start, cursor, counters, dcap, big_tiles, huge_tiles, mid_tiles, n_small, n_mid, n_large);
    else tile_order_body(tiles, count, order, counters);
}

__global__ __launch_bounds__(kBlock) void tile_scatter_kernel(int P, int gx, const uint2 *__restrict__ rect,
                                                              const float *__restrict__ depth,
                                                              uint32_t *__restrict__ cursor, uint64_t *__restrict__ keys,
                                                              uint64_t dcap) {
    __shared__ int s_box[4];
    // pass A: per-tile count of this block; pass B: the first slot reserved for the block in the tile's list; pass C: the running
    // slot.  ONE array (round 4: a second one for the bases made it 32 KB -- four workgroups per CU, 1.9 rounds of the 1954
    // workgroups at C3; with 16 KB the launch is one round)
    __shared__ uint32_t s_cnt[kWinTiles];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    uint64_t key = 0;
    if (i < P) {
        const uint2 rc = rect[i];
        x0 = rc.x & 0xffffu; y0 = rc.x >> 16; x1 = rc.y & 0xffffu; y1 = rc.y >> 16;
        key = ((uint64_t)__float_as_uint(depth[i]) << 32) | (uint32_t)i;
    }
    const bool visible = x1 > x0 && y1 > y0;
    const TileWindow win = block_tile_window(s_box, visible, x0, y0, x1, y1);
    const int area = win.area();
    if (area == 0) return;                                   // uniform
    if (win.fits()) {
        for (int k = tid; k < area; k += kBlock) s_cnt[k] = 0;
        __syncthreads();
        if (visible)
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) atomicAdd(&s_cnt[(ty - win.y0) * win.w + (tx - win.x0)], 1u);
        __syncthreads();
        const float inv_w = 1.0f / (float)win.w;
        for (int k = tid; k < area; k += kBlock) {           // ONE global (returning) atomic per touched tile
            const uint32_t c = s_cnt[k];
            if (c) s_cnt[k] = atomicAdd(&cursor[win.tile_of(k, gx, inv_w)], c);
        }
        __syncthreads();
        if (visible)
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) {
                    const int l = (ty - win.y0) * win.w + (tx - win.x0);
                    const uint32_t pos = atomicAdd(&s_cnt[l], 1u);
                    if (pos < dcap) keys[pos] = key;
                }
    } else if (visible) {
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                const uint32_t pos = atomicAdd(&cursor[ty * gx + tx], 1u);
                if (pos < dcap) keys[pos] = key;
            }
    }
}

// Bitonic network in its "flip / disperse" form: every compare-exchange moves the smaller key to the lower
// address, so slots >= n behave as +infinity without being stored and n need not be a power of two.
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr k, int n, int tid, int nthreads) {
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    const int half = n2 >> 1;
    // (sizes and distances are powers of two: shifts, not integer divisions -- a 32-bit division costs ~40 instructions)
    for (int size = 2, lsz = 1; size <= n2; size <<= 1, ++lsz) {
        const int hs = size >> 1;
        for (int i = tid; i < half; i += nthreads) {          // flip
            const int blk = i >> (lsz - 1), off = i & (hs - 1);
            const int lo = (blk << lsz) + off, hi = (blk << lsz) + size - 1 - off;
            if (hi < n) {
                const uint64_t a = k[lo], b = k[hi];
                if (b < a) { k[lo] = b; k[hi] = a; }
            }
        }
        __syncthreads();
        for (int j = hs >> 1, lj = lsz - 2; j >= 1; j >>= 1, --lj) {   // disperse
            for (int i = tid; i < half; i += nthreads) {
                const int blk = i >> lj, off = i & (j - 1);
                const int lo = (blk << (lj + 1)) + off, hi = lo + j;
                if (hi < n) {
                    const uint64_t a = k[lo], b = k[hi];
                    if (b < a) { k[lo] = b; k[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Depth sort of ONE tile list in LDS by BUCKETING (counting sort on a power-of-two quantisation of the depth bits, then an exact
// fix-up inside every bucket).  The bitonic network above moves every 8-byte key through LDS 36 times (2048 keys: 1.5 MB of
// LDS traffic per tile -- the sort was LDS-bandwidth bound, 52 us of the 82 us bin+sort at C3); here a key is written to LDS
// once and read about twice:
//   1. keys -> registers (CAP / BLOCK = 8 per thread); block-wide min / max of the depth bits (view-space z > 0.2, so the
//      float bits order like unsigned integers);
//   2. bucket b = (bits - min) >> sh with sh = the smallest shift that maps the range into CAP buckets: monotone in the depth,
//      so buckets are depth-ordered; one RETURNING LDS atomic per key counts the bucket and hands out a slot in it;
//   3. exclusive scan of the CAP counters (8 per thread + wave scan + wave totals);
//   4. the key goes to s_key[base[b] + slot]: bucket-ordered, arbitrary order inside a bucket (atomic arrival order);
//   5. exact position = base[b] + #{keys of the bucket smaller than mine} (full 64-bit compare: depth bits, then index -- the
//      total order of DESIGN.md sec. 2).  A bucket holds ~0.6 keys on average (n = 1200, CAP = 2048); the loop is O(bucket^2)
//      only for coplanar splats of IDENTICAL quantised depth, bounded by CAP.
// The result is the sorted list, independent of the atomic arrival order.
template <int BLOCK, int CAP>
__device__ __forceinline__ void sort_one_tile_bucket(uint64_t *s_key, uint32_t *s_bin, uint32_t *s_red, int tile,
                                                     const uint32_t *__restrict__ start, const uint64_t *__restrict__ keys,
                                                     uint32_t *__restrict__ point_list, uint64_t dcap) {
    static_assert(CAP == 8 * BLOCK, "eight keys / eight counters per thread");
    constexpr int NW = BLOCK / 64;
    const uint64_t b64 = min((uint64_t)start[tile], dcap), e64 = min((uint64_t)start[tile + 1], dcap);
    const int n = min((int)(e64 - b64), CAP);             // (lists are clamped only if the capacity overflowed)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t r[8];
    uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + k * BLOCK;
        r[k] = i < n ? keys[b64 + i] : ~0ull;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + k * BLOCK;
        if (i < n) { const uint32_t d = (uint32_t)(r[k] >> 32); lo = min(lo, d); hi = max(hi, d); }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
    }
    if (lane == 0) { s_red[wave] = lo; s_red[NW + wave] = hi; }
#pragma unroll
    for (int k = 0; k < 8; ++k) s_bin[tid + k * BLOCK] = 0u;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) { lo = min(lo, s_red[w]); hi = max(hi, s_red[NW + w]); }
    const uint32_t range = hi - lo;                                         // < 2^32
    int sh = 32 - __clz((int)(range | 1u)) - (31 - __clz(CAP));             // range >> sh < CAP
    if (range & 0x80000000u) sh = 32 - (31 - __clz(CAP));                   // (__clz of a "negative" int)
    sh = max(sh, 0);
    uint32_t bucket[8], slot[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + k * BLOCK;
        bucket[k] = ((uint32_t)(r[k] >> 32) - lo) >> sh;
        slot[k] = 0u;
        if (i < n) slot[k] = atomicAdd(&s_bin[bucket[k]], 1u);
    }
    __syncthreads();
    {   // exclusive scan of the CAP counters, in place; s_bin[CAP] = n
        uint32_t c[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { c[k] = s_bin[8 * tid + k]; sum += c[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += u;
        }
        if (lane == 63) s_red[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
#pragma unroll
        for (int w = 0; w < NW; ++w) if (w < wave) run += s_red[w];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s_bin[8 * tid + k] = run; run += c[k]; }
        if (tid == BLOCK - 1) s_bin[CAP] = run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + k * BLOCK;
        if (i < n) s_key[s_bin[bucket[k]] + slot[k]] = r[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + k * BLOCK;
        if (i < n) {
            const uint32_t b0 = s_bin[bucket[k]], b1 = s_bin[bucket[k] + 1];
            uint32_t less = 0;
            for (uint32_t p = b0; p < b1; ++p) less += s_key[p] < r[k] ? 1u : 0u;
            point_list[b64 + b0 + less] = (uint32_t)r[k];
        }
    }
}

// Lists longer than the largest LDS class (> CAP keys): the same bucket sort in SEGMENTS.  One pass over the list (from HBM)
// builds the CAP-bucket histogram of the quantised depth and, by the scan, every bucket's final offset in the output; the
// bucket range is then cut into consecutive segments of at most CAP keys, and for each segment the list is streamed once more:
// keys of the segment's buckets are gathered into LDS, ordered exactly inside their buckets and written to their final
// positions.  ceil(n / CAP) + 1 passes over n 8-byte keys instead of a log^2(n)-stage bitonic network on global memory
// (round 1; 0.78 ms for one 11 926-entry tile list).  A single bucket with more than CAP keys (that many splats of identical
// quantised depth) makes the caller fall back to that network.  Returns false in that case (wave-uniform for the workgroup).
template <int BLOCK, int CAP>
__device__ __forceinline__ bool sort_huge_tile_bucket(uint64_t *s_key, uint32_t *s_bin, uint32_t *s_cur, uint32_t *s_red, int tile,
                                                      const uint32_t *__restrict__ start, const uint64_t *__restrict__ keys,
                                                      uint32_t *__restrict__ point_list, uint64_t dcap) {
    static_assert(CAP == 8 * BLOCK, "eight counters per thread");
    constexpr int NW = BLOCK / 64;
    const uint64_t b64 = min((uint64_t)start[tile], dcap), e64 = min((uint64_t)start[tile + 1], dcap);
    const int n = (int)(e64 - b64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int i = tid; i < n; i += BLOCK) { const uint32_t d = (uint32_t)(keys[b64 + i] >> 32); lo = min(lo, d); hi = max(hi, d); }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
    }
    __syncthreads();
    if (lane == 0) { s_red[wave] = lo; s_red[NW + wave] = hi; }
#pragma unroll
    for (int k = 0; k < 8; ++k) s_bin[tid + k * BLOCK] = 0u;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) { lo = min(lo, s_red[w]); hi = max(hi, s_red[NW + w]); }
    const uint32_t range = hi - lo;
    const int sh = max(32 - __clz((int)(range | 1u)) - (31 - __clz(CAP)), 0);
    for (int i = tid; i < n; i += BLOCK) atomicAdd(&s_bin[((uint32_t)(keys[b64 + i] >> 32) - lo) >> sh], 1u);
    __syncthreads();
    uint32_t cmax = 0;
    {   // exclusive scan of the CAP counters, in place; s_bin[CAP] = n; largest single bucket
        uint32_t c[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { c[k] = s_bin[8 * tid + k]; sum += c[k]; cmax = max(cmax, c[k]); }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += u;
            cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, off));
        }
        __syncthreads();
        if (lane == 63) { s_red[wave] = incl; s_red[NW + wave] = cmax; }
        __syncthreads();
        uint32_t run = incl - sum;
#pragma unroll
        for (int w = 0; w < NW; ++w) { if (w < wave) run += s_red[w]; cmax = max(cmax, s_red[NW + w]); }
#pragma unroll
        for (int k = 0; k < 8; ++k) { s_bin[8 * tid + k] = run; run += c[k]; }
        if (tid == BLOCK - 1) s_bin[CAP] = run;
    }
    __syncthreads();
    if (cmax > (uint32_t)CAP) return false;                  // one bucket alone does not fit: caller falls back
    for (int segA = 0; segA < CAP;) {
        // largest segB with base[segB] - base[segA] <= CAP (at least one bucket: every bucket fits)
        if (tid == 0) {
            const uint32_t limit = s_bin[segA] + (uint32_t)CAP;
            int a = segA + 1, b = CAP;                       // invariant: base[a] <= limit
            while (a < b) { const int m = (a + b + 1) >> 1; if (s_bin[m] <= limit) a = m; else b = m - 1; }
            s_red[0] = (uint32_t)a;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s_cur[tid + k * BLOCK] = 0u;
        __syncthreads();
        const int segB = (int)s_red[0];
        const uint32_t baseA = s_bin[segA], m = s_bin[segB] - baseA;
        if (m) {
            for (int i = tid; i < n; i += BLOCK) {
                const uint64_t key = keys[b64 + i];
                const uint32_t b = ((uint32_t)(key >> 32) - lo) >> sh;
                if (b >= (uint32_t)segA && b < (uint32_t)segB) s_key[s_bin[b] - baseA + atomicAdd(&s_cur[b], 1u)] = key;
            }
            __syncthreads();
            for (uint32_t i = tid; i < m; i += BLOCK) {
                const uint64_t key = s_key[i];
                const uint32_t b = ((uint32_t)(key >> 32) - lo) >> sh;
                const uint32_t b0 = s_bin[b] - baseA, b1 = s_bin[b + 1] - baseA;
                uint32_t less = 0;
                for (uint32_t p = b0; p < b1; ++p) less += s_key[p] < key ? 1u : 0u;
                point_list[b64 + baseA + b0 + less] = (uint32_t)key;
            }
        }
        __syncthreads();
        segA = segB;
    }
    return true;
}

// Resident workgroups walk the tiles in the order of tile_scan_order_kernel (descending list length: the empty tiles are
// last) and stop at the first empty one; tiles with longer lists are left to the list-driven kernels below.  (Rounds 2-3
// launched one workgroup per tile: at C3 6826 of the 8160 workgroups found an empty tile -- two dependent loads each while
// holding a slot and 24 KB of LDS -- and took as long as the 1334 that sorted: 12.8 -> .. us.)
template <int BLOCK, int CAP>
__global__ __launch_bounds__(BLOCK) void tile_sort_lds_kernel(const uint32_t *__restrict__ start,
                                                              const uint64_t *__restrict__ keys,
                                                              uint32_t *__restrict__ point_list, uint64_t dcap,
                                                              const uint32_t *__restrict__ order, int tiles) {
    __shared__ uint64_t s_key[CAP];
    __shared__ __attribute__((aligned(8))) uint32_t s_bin[CAP + 1];
    __shared__ uint32_t s_red[2 * (BLOCK / 64)];
    for (int rank = blockIdx.x; rank < tiles; rank += gridDim.x) {
        const int tile = (int)order[rank];
        const uint32_t n = start[tile + 1] - start[tile];
        if (n == 0) return;                                  // (uniform) every later rank is empty as well
        if (n <= (uint32_t)CAP) sort_one_tile_bucket<BLOCK, CAP>(s_key, s_bin, s_red, tile, start, keys, point_list, dcap);
        __syncthreads();                                     // the LDS arrays are reused
    }
}

// persistent grid over a work list written by the scan workgroup of tile_scan_order_kernel (tiles whose list does not fit the kernel above);
// dynamic LDS: CAP keys + CAP + 1 bucket counters (+ reduction scratch): 48 KB for 4096, 96 KB for 8192
template <int BLOCK, int CAP>
__global__ __launch_bounds__(BLOCK) void tile_sort_lds_list_kernel(const uint32_t *__restrict__ start,
                                                                   const uint64_t *__restrict__ keys,
                                                                   uint32_t *__restrict__ point_list, uint64_t dcap,
                                                                   const uint32_t *__restrict__ list,
                                                                   const uint32_t *__restrict__ list_count,
                                                                   uint64_t *__restrict__ keys_rw,
                                                                   const uint32_t *__restrict__ huge_list,
                                                                   const uint32_t *__restrict__ huge_count,
                                                                   const uint32_t *__restrict__ list2,
                                                                   const uint32_t *__restrict__ list2_count) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_key_dyn[];
    uint32_t *s_bin = reinterpret_cast<uint32_t *>(s_key_dyn + CAP);
    uint32_t *s_red = s_bin + CAP + 1;
    uint32_t *s_cur = s_red + 2 * (BLOCK / 64) + 2;              // huge lists only (the launch sizes the LDS accordingly)
    const uint32_t count = *list_count;
    for (uint32_t v = blockIdx.x; v < count; v += gridDim.x) {
        __syncthreads();
        sort_one_tile_bucket<BLOCK, CAP>(s_key_dyn, s_bin, s_red, (int)list[v], start, keys, point_list, dcap);
    }
    // a second work list of a smaller class in the same launch (round 5: frames with few lists of the 2049..4096 class send them
    // through the 8192-key kernel instead of a launch of its own -- at C3 that launch sorted ONE list and the other found none)
    if (list2) {
        const uint32_t count2 = *list2_count;
        for (uint32_t v = gridDim.x - 1u - blockIdx.x; v < count2; v += gridDim.x) {      // (from the other end of the grid)
            __syncthreads();
            sort_one_tile_bucket<BLOCK, CAP>(s_key_dyn, s_bin, s_red, (int)list2[v], start, keys, point_list, dcap);
        }
    }
    // lists that do not fit any LDS class (huge_list != null: same launch, saves a near-empty grid per frame): segmented
    // bucket sort; the bitonic network on global memory only if one bucket alone exceeds the LDS class
    if (huge_list) {
        const uint32_t hcount = *huge_count;
        const int tid = threadIdx.x;
        for (uint32_t k = blockIdx.x; k < hcount; k += gridDim.x) {
            const int tile = (int)huge_list[k];
            const uint64_t b64 = min((uint64_t)start[tile], dcap), e64 = min((uint64_t)start[tile + 1], dcap);
            const int n = (int)(e64 - b64);
            __syncthreads();
            if (!sort_huge_tile_bucket<BLOCK, CAP>(s_key_dyn, s_bin, s_cur, s_red, tile, start, keys, point_list, dcap)) {
                bitonic_sort(keys_rw + b64, n, tid, BLOCK);
                for (int i = tid; i < n; i += BLOCK) point_list[b64 + i] = (uint32_t)keys_rw[b64 + i];
            }
        }
    }
}

}  // namespace d3ga

using namespace d3ga;

constexpr int kSortSmall = 2048;   // 24 KiB LDS, 256 threads, one workgroup per tile
constexpr int kSortMid = 4096;     // 48 KiB LDS, 512 threads, list-driven
constexpr int kSortLarge = 8192;   // 96 KiB LDS, 1024 threads, list-driven
static inline size_t sort_lds_bytes(int cap, int block, bool huge = false) {       // keys | bases | reduction scratch [| cursors]
    return (size_t)cap * 8 + ((size_t)cap + 1 + 2 * (block / 64) + 2 + (huge ? cap : 0)) * 4 + 16;
}


extern "C" int d3ga_raster_bin_sort(const d3ga_raster_params *prm, void *geom, void *binning, int64_t d_capacity,
                                    d3ga_stream_t stream) {
    if (!prm || !geom || !binning) return D3GA_E_NULL;
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0 || d_capacity < 0) return D3GA_E_SIZE;
    hipStream_t s = (hipStream_t)stream;
    // a batch of views is one tall frame for this stage: views x P records, views x tiles lists (d3ga.h: n_views)
    const int views = n_views_of(prm);
    const int gx = tiles_x(prm->W);
    const int tiles = gx * tiles_y(prm->H) * views;
    const int64_t P = (int64_t)prm->P * views;
    BinBuf bin = carve_bin(binning, tiles, d_capacity);
    GeomBuf g = carve_geom(geom, P);
    hipLaunchKernelGGL(tile_scan_order_kernel, dim3(2), dim3(kScanBlock), 0, s, tiles, bin.tile_count, bin.tile_start,
                       bin.tile_cursor, bin.counters, (uint64_t)d_capacity, bin.big_tiles, bin.huge_tiles, bin.mid_tiles,
                       (uint32_t)kSortSmall, (uint32_t)kSortMid, (uint32_t)kSortLarge, bin.tile_order);
    D3GA_TRY(check_launch(s, prm->debug));
    if (P == 0 || d_capacity == 0) return D3GA_OK;
    hipLaunchKernelGGL(tile_scatter_kernel, dim3((unsigned)((P + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, (int)P, gx, g.rect,
                       g.depth, bin.tile_cursor, bin.keys, (uint64_t)d_capacity);
    D3GA_TRY(check_launch(s, prm->debug));
    {
        static std::atomic<int> resident[64] = {};           // per device: workgroups of the per-tile sort the chip holds at once (a cache of two queries)
        int dev = 0;
        D3GA_HIP(hipGetDevice(&dev));
        int want = (dev >= 0 && dev < 64) ? resident[dev].load(std::memory_order_relaxed) : 0;
        if (want == 0) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)tile_sort_lds_kernel<256, kSortSmall>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            want = per_cu * cus;
            if (dev >= 0 && dev < 64) resident[dev].store(want, std::memory_order_relaxed);
        }
        hipLaunchKernelGGL((tile_sort_lds_kernel<256, kSortSmall>), dim3(tiles < want ? tiles : want), dim3(256), 0, s, bin.tile_start,
                           bin.keys, bin.point_list, (uint64_t)d_capacity, (const uint32_t *)bin.tile_order, tiles);
    }
    D3GA_TRY(check_launch(s, prm->debug));
    // longer lists: persistent grids driven by the device-side work lists (empty for avatar-sized scenes), 256 workgroups each.
    // (Round 3 also tried sorting the 2049..4096 class inside the per-tile kernel, in segments: one launch less, but a single
    // 256-thread workgroup then takes 2.5x as long as any other -- 13 -> 21 us for that kernel at C3, no net gain; reverted.)
    // (... and running the two list kernels on a side stream beside the per-tile kernel -- fork / join events, parallel graph
    // branches under capture: the events cost more than the overlap saves, bin+sort 49 -> 64 us eager, step 0.409 -> 0.422 ms.)
    {   // > 64 KiB of dynamic LDS needs an explicit opt-in, once per device (function attributes are per device)
        static std::atomic<bool> attr_set[64] = {};
        int dev = 0;
        D3GA_HIP(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !attr_set[dev].load(std::memory_order_relaxed)) {
            D3GA_HIP(hipFuncSetAttribute((const void *)tile_sort_lds_list_kernel<1024, kSortLarge>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds_bytes(kSortLarge, 1024, true)));
            attr_set[dev].store(true, std::memory_order_relaxed);
        }
    }
    // (48 KB of LDS: three workgroups of the 2049..4096 class fit a CU -- at 4K with 2M Gaussians thousands of tiles are in it;
    //  96 KB: one workgroup of the larger class per CU)
    const int lgrid = tiles < 256 ? tiles : 256, mgrid = tiles < 768 ? tiles : 768;
    // ONE list launch when few lists of the 2049..4096 class are expected: they ride in the 8192-key kernel (one workgroup of 1024
    // threads per CU instead of three of 512 -- for a handful of such lists that is no loss, and a launch that finds an empty work
    // list still costs 4-5 us of a frame).  "Few" is decided from what the host knows without a read-back: the mean list length over
    // ALL tiles of the frame, d_capacity / tiles (the capacity tracks the duplicate count: its high-water mark + 25 %).  An avatar
    // frame at 1080p (C3: 2.1 M / 8160 = 255) has a handful of mid-class lists; a dense scene -- millions of Gaussians at 1080p, or
    // C5's 4K frame (13 M / 32 400 = 400, thousands of tiles in the class) -- keeps the launch of its own, three workgroups per CU
    // (the tile count alone sent a dense 1080p scene through the one-workgroup-per-CU kernel).  D3GA_KNOB_SORT_MERGE overrides.
    const int merge_knob = debug_knob(D3GA_KNOB_SORT_MERGE);
    const bool merged = merge_knob >= 0 ? merge_knob != 0 : d_capacity / (tiles > 0 ? tiles : 1) < 320;
#define D3GA_LAUNCH_LIST_SORT(BLOCKV, CAPV, GRID, HUGE, LIST, CNT, KEYS_RW, HLIST, HCNT, LIST2, CNT2)                              \
    hipLaunchKernelGGL((tile_sort_lds_list_kernel<BLOCKV, CAPV>), dim3(GRID), dim3(BLOCKV), sort_lds_bytes(CAPV, BLOCKV, HUGE), s, \
                       bin.tile_start, bin.keys, bin.point_list, (uint64_t)d_capacity, LIST, CNT, KEYS_RW, HLIST, HCNT, LIST2, CNT2)
    const uint32_t *const nil = nullptr;
    if (!merged) {
        D3GA_LAUNCH_LIST_SORT(512, kSortMid, mgrid, false, bin.mid_tiles, bin.counters + D3GA_CNT_MID, (uint64_t *)nullptr, nil, nil, nil, nil);
        D3GA_TRY(check_launch(s, prm->debug));
    }
    const uint32_t *l2 = merged ? (const uint32_t *)bin.mid_tiles : nil, *c2 = merged ? (const uint32_t *)(bin.counters + D3GA_CNT_MID) : nil;
    D3GA_LAUNCH_LIST_SORT(1024, kSortLarge, lgrid, true, bin.big_tiles, bin.counters + D3GA_CNT_BIG, bin.keys,
                          (const uint32_t *)bin.huge_tiles, (const uint32_t *)(bin.counters + D3GA_CNT_HUGE), l2, c2);
#undef D3GA_LAUNCH_LIST_SORT
    return check_launch(s, prm->debug);
}

// d3ga_internal.h -- scratch-buffer layout and launch helpers shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/d3ga.h"
#include "d3ga_math.h"

namespace d3ga {

constexpr int kBlock = 256;

// the library's debug knobs (include/d3ga.h: D3GA_KNOB_*, d3ga_debug_set): value in effect.  No environment is read anywhere.
int debug_knob(int key);

static inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

// Per-Gaussian state written by the preprocess kernel (structure of arrays, 256-byte aligned sections).
// View-batched renders (d3ga_raster_*_views): the arrays hold n_views * P records, view v's Gaussian i at v * P + i.
struct GeomBuf {
    float *depth;       // P      view-space z
    float4 *conic_o;    // P      conic (a,b,c) + opacity
    float4 *rgb_invd;   // P      colour + 1/depth
    uint2 *rect;        // P      tile rectangle packed: x = minx | miny<<16, y = maxx | maxy<<16
    uint8_t *clamped;   // P      bit c set: SH colour channel c was clamped at 0
    float *cov3D;       // P*6    3D covariance built from (scale, rotation); unused with a precomputed one (read from the caller's tensor)
    float4 *xyh;        // P      pixel-space centre | half extents of the alpha >= 1/255 ellipse's bounding box (splat_cull):
                        //        the 16-byte record the compositing forward's first stage tests a list entry with
    float *dcol;        // 9 planes of `dcol_stride` floats: d(SH colour)/d(unit view direction), plane 3 * (direction x, y, z) + channel -- left by
                        //        the forward's staged SH evaluation so that preprocess_bwd need not read the coefficients again (96 MB at C3) for dL/dmean
    int64_t dcol_stride;
};
static inline int64_t geom_bytes(int64_t P) {
    return align256(4 * P) + align256(16 * P) + align256(16 * P) + align256(8 * P) + align256(P) +
           align256(24 * P) + align256(16 * P) + align256(36 * P);
}
static inline GeomBuf carve_geom(void *base, int64_t P) {
    char *p = (char *)base;
    GeomBuf g;
    g.depth = (float *)p;     p += align256(4 * P);
    g.conic_o = (float4 *)p;  p += align256(16 * P);
    g.rgb_invd = (float4 *)p; p += align256(16 * P);
    g.rect = (uint2 *)p;      p += align256(8 * P);
    g.clamped = (uint8_t *)p; p += align256(P);
    g.cov3D = (float *)p;     p += align256(24 * P);
    g.xyh = (float4 *)p;      p += align256(16 * P);
    g.dcol = (float *)p;
    g.dcol_stride = P;
    return g;
}
// the records of view v of a batch (every array advanced by v * P records; the dcol planes keep the batch's stride)
__host__ __device__ static inline GeomBuf geom_view(const GeomBuf &g, int64_t P, int64_t v) {
    GeomBuf o = g;
    const int64_t d = v * P;
    o.depth += d; o.conic_o += d; o.rgb_invd += d; o.rect += d; o.clamped += d; o.cov3D += 6 * d; o.xyh += d; o.dcol += d;
    return o;
}

// Binning state.
struct BinBuf {
    uint32_t *counters;     // 8  (D3GA_CNT_*)
    uint32_t *tile_count;   // tiles
    uint32_t *tile_start;   // tiles + 1   exclusive prefix of tile_count
    uint32_t *tile_cursor;  // tiles
    uint64_t *keys;         // d_capacity  (depth bits << 32 | gaussian index), grouped by tile
    uint32_t *point_list;   // d_capacity  gaussian indices, per tile ascending (depth, index)
    uint32_t *big_tiles;    // tiles       tiles with 4097..8192 entries (counters[4] of them)
    uint32_t *huge_tiles;   // tiles       tiles with more than 8192 entries: no LDS sort (counters[5])
    uint32_t *mid_tiles;    // tiles       tiles with 2049..4096 entries (counters[6])
    uint32_t *tile_order;   // tiles       tile indices by descending list length (work-ordered dispatch of compositing)
};
static inline int64_t bin_bytes(int64_t tiles, int64_t dcap) {
    return 256 + align256(4 * tiles) + align256(4 * (tiles + 1)) + align256(4 * tiles) + align256(8 * dcap) +
           align256(4 * dcap) + 4 * align256(4 * tiles);
}
static inline BinBuf carve_bin(void *base, int64_t tiles, int64_t dcap) {
    char *p = (char *)base;
    BinBuf b;
    b.counters = (uint32_t *)p;    p += 256;
    b.tile_count = (uint32_t *)p;  p += align256(4 * tiles);
    b.tile_start = (uint32_t *)p;  p += align256(4 * (tiles + 1));
    b.tile_cursor = (uint32_t *)p; p += align256(4 * tiles);
    b.keys = (uint64_t *)p;        p += align256(8 * dcap);
    b.point_list = (uint32_t *)p;  p += align256(4 * dcap);
    b.big_tiles = (uint32_t *)p;   p += align256(4 * tiles);
    b.huge_tiles = (uint32_t *)p;  p += align256(4 * tiles);
    b.mid_tiles = (uint32_t *)p;   p += align256(4 * tiles);
    b.tile_order = (uint32_t *)p;
    return b;
}

struct ImgBuf {
    float *final_T;        // H*W  (view-batched: n_views images back to back)
    uint32_t *n_contrib;   // H*W   1-based tile-list position of the last contributing Gaussian
    // Per-4x4-block culled lists, written by the compositing forward for its backward (raster_composite_scan.hip):
    // a 16x16 tile has 16 blocks (quadrant q = 0..3, row r = 0..3 of that quadrant's wavefront -> block 4q + r); the
    // list of block b of a tile whose depth-sorted list is [begin, end) occupies blk_list[16*begin + b*(end-begin) ...],
    // entries {1-based position in the tile list, Gaussian index}, front to back; blk_count[16*tile + b] = the length of the
    // prefix that holds every entry some pixel of the block blended -- what the backward walks.
    uint32_t *blk_count;   // 16 * tiles
    uint2 *blk_list;       // 16 * d_capacity
};
// W x H: one view; tiles: of the whole batch; views: images in the batch
static inline int64_t img_bytes(int64_t W, int64_t H, int64_t tiles, int64_t dcap, int64_t views = 1) {
    return 2 * align256(4 * W * H * views) + align256(64 * tiles) + align256(128 * dcap);
}
static inline ImgBuf carve_img(void *base, int64_t W, int64_t H, int64_t tiles, int64_t views = 1) {
    ImgBuf i;
    char *p = (char *)base;
    i.final_T = (float *)p;        p += align256(4 * W * H * views);
    i.n_contrib = (uint32_t *)p;   p += align256(4 * W * H * views);
    i.blk_count = (uint32_t *)p;   p += align256(64 * tiles);
    i.blk_list = (uint2 *)p;
    return i;
}

// ---------------------------------------------------------------------------------------------------------
// Block-cooperative tile window.  The Gaussians of one 256-thread block are spatially coherent (they are stored in
// tet order), so the union of their tile rectangles is small.  Counting / slot reservation is done with LDS atomics
// inside that bounding window and only ONE global atomic per (block, touched tile) leaves the CU, instead of one per
// (Gaussian, tile) -- which serialises badly because neighbouring lanes hit the same few tile counters.
// ---------------------------------------------------------------------------------------------------------
constexpr int kWinTiles = 4096;   // LDS window capacity in tiles (16 KiB of u32); larger unions fall back to global atomics

struct TileWindow {
    int x0, y0, w, h;             // window origin and size in tiles; w*h == 0 -> nothing visible in the block
    __device__ __forceinline__ int area() const { return w * h; }
    __device__ __forceinline__ bool fits() const { return w * h <= kWinTiles; }
    // window slot k -> global tile index (exact for k < 4096: the fractional part of (k + 0.5) / w is >= 0.5 / w away from
    // an integer, the float error is < 5e-4 / w; a 32-bit integer division costs ~40 instructions)
    __device__ __forceinline__ int tile_of(int k, int gx, float inv_w) const {
        const int r = (int)(((float)k + 0.5f) * inv_w);
        return (y0 + r) * gx + x0 + (k - r * w);
    }
};

#ifdef __HIPCC__
// s_box: 4 ints of LDS.  Every thread of the block must call this (contains barriers).
__device__ __forceinline__ TileWindow block_tile_window(int *s_box, bool visible, int rx0, int ry0, int rx1, int ry1) {
    if (threadIdx.x == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = 0; s_box[3] = 0; }
    __syncthreads();
    // wavefront-level min/max first (butterfly shuffles), then ONE lane per wavefront touches the four LDS words:
    // 256 same-address LDS atomics per word serialise
    int bx0 = visible ? rx0 : 0x7fffffff, by0 = visible ? ry0 : 0x7fffffff, bx1 = visible ? rx1 : 0, by1 = visible ? ry1 : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, off)); by0 = min(by0, __shfl_xor(by0, off));
        bx1 = max(bx1, __shfl_xor(bx1, off)); by1 = max(by1, __shfl_xor(by1, off));
    }
    if ((threadIdx.x & 63) == 0 && bx1 > 0) {
        atomicMin(&s_box[0], bx0); atomicMin(&s_box[1], by0);
        atomicMax(&s_box[2], bx1); atomicMax(&s_box[3], by1);
    }
    __syncthreads();
    TileWindow win;
    win.x0 = s_box[0]; win.y0 = s_box[1];
    win.w = s_box[2] > s_box[0] ? s_box[2] - s_box[0] : 0;
    win.h = s_box[3] > s_box[1] ? s_box[3] - s_box[1] : 0;
    if (win.w == 0 || win.h == 0) { win.w = 0; win.h = 0; }
    return win;
}
#endif

static inline int tiles_x(int W) { return (W + kTile - 1) / kTile; }
static inline int tiles_y(int H) { return (H + kTile - 1) / kTile; }
// cameras in the batch (d3ga_raster_params::n_views: 0 and 1 both mean one)
static inline int n_views_of(const d3ga_raster_params *prm) { return prm->n_views > 1 ? prm->n_views : 1; }

// launch check: returns hipError (>0) or 0; in debug mode also synchronises
static inline int check_launch(hipStream_t s, int debug) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (debug) {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

#define D3GA_TRY(expr)            \
    do {                          \
        int _st = (expr);         \
        if (_st != 0) return _st; \
    } while (0)
#define D3GA_HIP(expr)                          \
    do {                                        \
        hipError_t _e = (expr);                 \
        if (_e != hipSuccess) return (int)_e;   \
    } while (0)

// Zero-fill as a KERNEL node.  hipMemsetAsync becomes a memset node under stream capture, and on this stack (ROCm 7.2,
// gfx950) a replayed graph did not reliably order such a node against the kernels around it: with the tile histogram
// zeroed by a memset node, replays of preprocess -> bin_sort faulted as soon as anything disturbed the caches between two
// replays (tools/diag_graph3.py), i.e. the histogram was being zeroed while the next kernel already counted into it.
// Kernel -> kernel edges are honoured, so every clear in this library is a (tiny) kernel.
#ifdef __HIPCC__
static __global__ void zero_words_kernel(uint32_t *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static __global__ void zero_bytes_kernel(unsigned char *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0;
}
static inline hipError_t zero_async(void *ptr, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((((uintptr_t)ptr | bytes) & 3) == 0) {
        const size_t n = bytes / 4;
        const size_t blocks = (n + 4 * kBlock - 1) / (4 * kBlock);
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks))), dim3(kBlock), 0, s,
                           reinterpret_cast<uint32_t *>(ptr), n);
    } else {
        const size_t blocks = (bytes + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(kBlock), 0, s,
                           reinterpret_cast<unsigned char *>(ptr), bytes);
    }
    return hipGetLastError();
}
#endif

}  // namespace d3ga

// raster_pre.hip -- per-Gaussian stages of the tile rasterizer for gfx950.
//
//   preprocess_kernel      project, EWA 2D covariance, conic, radius, tile rectangle, SH colour, tile histogram
//   preprocess_bwd_kernel  conic/mean2D/colour/opacity gradients -> mean3D, cov3D | (scale, rotation), SH
//
// Both are HBM-streaming: one thread per Gaussian, every per-Gaussian array read/written once (algorithmic bytes per
// Gaussian: fwd 88 + 12 M).  The wide (P,M,3) SH arrays go through per-wavefront LDS slabs so that
// global accesses are 16 B per lane and lane-contiguous; the camera matrices are wave-uniform and come through the
// scalar cache.  No MFMA.
#include "composite_common.h"
#include "raster_pre_body.h"

namespace d3ga {

// ---------------------------------------------------------------------------------------------------------
// SH staging.  shs / dL_dsh are (P, M, 3): 12*M bytes per Gaussian, so a thread-per-Gaussian access walks memory with
// a 192-byte stride (M = 16) and every load instruction of a wavefront touches 64 different cache lines -- measured
// 3.5x over-fetch.  Instead each wavefront moves the 64 rows it owns as ONE contiguous block with 16-byte-per-lane,
// lane-contiguous accesses (1 KiB per instruction) through an LDS slab, and every lane then works on its own row in
// LDS.  Rows are padded to 52 floats = 13 x 16 B (odd number of 16-byte slots -> conflict-free b128 row access).
// Used when 3*M is a multiple of 4 floats (M = 4, 8, 12, 16; D3GA always has M = 16); otherwise rows are accessed
// in global memory directly.
// ---------------------------------------------------------------------------------------------------------
constexpr int kShRow = 52;
constexpr int kShSlab = 64 * kShRow;            // floats per wavefront
constexpr size_t kShLdsBytes = (size_t)(kBlock / 64) * kShSlab * sizeof(float);   // 53,248 B per block

__device__ __forceinline__ bool sh_staged(int M) { return M > 0 && (3 * M) % 4 == 0 && 3 * M <= 48; }
#ifndef D3GA_PRE_DCOL        // 1: the forward leaves GeomBuf::dcol for the backward (round 5); 0: the backward reads the coefficients again (A/B build)
#define D3GA_PRE_DCOL 1
#endif
#ifndef D3GA_DCOL_PLANAR     // 1 (default): GeomBuf::dcol as nine planes of P floats -- every store / load instruction is one contiguous run of a
                             // wavefront (preprocess 57 -> 54.5 us against 36-byte records, same-box A/B); 0: records (A/B build)
#define D3GA_DCOL_PLANAR 1
#endif
#if D3GA_DCOL_PLANAR
#define D3GA_DCOL_AT(base, i, k, P) ((base)[(size_t)(k) * (size_t)(P) + (size_t)(i)])       /* P: the planes' stride (GeomBuf::dcol_stride) */
#else
#define D3GA_DCOL_AT(base, i, k, P) ((base)[9 * (size_t)(i) + (k)])
#endif

// global (rows x 3M floats, contiguous) -> LDS slab; `rows` valid rows of this wavefront (<= 64)
__device__ __forceinline__ void sh_slab_load(float *slab, const float *__restrict__ src, int rows, int M3, int lane) {
    const int nvec = rows * M3 / 4;
    for (int v = lane; v < nvec; v += 64) {
        const float4 x = reinterpret_cast<const float4 *>(src)[v];
        const int e = 4 * v, r = e / M3, c = e - r * M3;
        *reinterpret_cast<float4 *>(slab + r * kShRow + c) = x;
    }
}
__device__ __forceinline__ void sh_slab_store(const float *slab, float *__restrict__ dst, int rows, int M3, int lane) {
    const int nvec = rows * M3 / 4;
    for (int v = lane; v < nvec; v += 64) {
        const int e = 4 * v, r = e / M3, c = e - r * M3;
        reinterpret_cast<float4 *>(dst)[v] = *reinterpret_cast<const float4 *>(slab + r * kShRow + c);
    }
}

// The common case -- full rows of 48 floats (M = 16) and a full set of ROWS rows -- with every load of the set in flight at
// once: the loop above is one load -> wait -> LDS store per iteration (an integer division by the run-time row length
// sits between them), i.e. ROWS*12/64 memory round trips in series per wavefront, and that chain, not bandwidth, set the
// pace of both per-Gaussian kernels (preprocess 53 -> .. us, preprocess_bwd 58 -> .. us at C3).
template <int ROWS>
struct ShRegs { float4 v[ROWS * 12 / 64]; };
template <int ROWS>
__device__ __forceinline__ ShRegs<ROWS> sh_rows48_load(const float *__restrict__ src, int lane) {
    ShRegs<ROWS> r;
#pragma unroll
    for (int k = 0; k < ROWS * 12 / 64; ++k) r.v[k] = reinterpret_cast<const float4 *>(src)[lane + 64 * k];
    return r;
}
template <int ROWS>
__device__ __forceinline__ void sh_rows48_to_slab(float *slab, const ShRegs<ROWS> &r, int lane) {
#pragma unroll
    for (int k = 0; k < ROWS * 12 / 64; ++k) {
        const int e = 4 * (lane + 64 * k), row = e / 48, c = e - row * 48;
        *reinterpret_cast<float4 *>(slab + row * kShRow + c) = r.v[k];
    }
}
template <int ROWS>
__device__ __forceinline__ void sh_rows48_store(const float *slab, float *__restrict__ dst, int lane) {
#pragma unroll
    for (int k = 0; k < ROWS * 12 / 64; ++k) {
        const int e = 4 * (lane + 64 * k), row = e / 48, c = e - row * 48;
        reinterpret_cast<float4 *>(dst)[lane + 64 * k] = *reinterpret_cast<const float4 *>(slab + row * kShRow + c);
    }
}

// forward staging (see preprocess_kernel): kShPassRows of a wavefront's 64 rows at a time
#ifndef D3GA_SH_PASS_ROWS
#define D3GA_SH_PASS_ROWS 32
#endif
constexpr int kShPassRows = D3GA_SH_PASS_ROWS;                  // 32: two passes (default), 16: four (measured, slower: below)
constexpr int kShPasses = 64 / kShPassRows;
constexpr int kShHalfSlab = kShPassRows * kShRow;
constexpr size_t kShHalfLdsBytes = (size_t)(kBlock / 64) * kShHalfSlab * sizeof(float);      // 26,624 B per block (32 rows)

// sum_k Y_k(dir_i) * coeff_k of this thread's Gaussian, with the (P,M,3) block read through wavefront-private LDS.
// SH colour in PASSES over a part of the wavefront's rows (full 192-byte rows, so every byte is fetched once): the slab of
// a wavefront is kShPassRows x 52 floats -- 13 KiB for all 64 rows (12 resident wavefronts per CU), 6.5 KiB for 32 (20: the
// default).  Round 4 measured 16 rows per pass (3.3 KiB: the tile window's 16 KiB is then the block's LDS, 28 wavefronts per
// CU, the launch's 1954 workgroups in 1.09 instead of 1.53 rounds): with all four quarters in flight the registers hold the
// kernel at 20 wavefronts anyway (88 VGPRs); with two in flight (68 VGPRs, 28 wavefronts) preprocess takes 61-65 us against
// 55.6 (eager stage events, same box) -- the second memory round trip per wavefront and a quarter of the lanes per
// accumulate pass cost more than the occupancy returns.  Only wavefront-private LDS is touched: no workgroup barrier,
// program order + wave_barrier suffice.  Every thread of the block must call it.
template <bool WANT_J>
__device__ __forceinline__ void staged_sh_colour(const d3ga_raster_params &prm, const float *__restrict__ means3D,
                                                 const float *__restrict__ shs, const float *__restrict__ campos,
                                                 float *s_sh, float acc[3], ShColJ &cj) {
    constexpr bool want_j = WANT_J;
    // WANT_J: also J = d(colour)/d(unit direction) of this Gaussian (sh_accumulate_jacobian), from the same staged row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * kBlock + tid;
    const int M3 = 3 * prm.M;
    float *slab = s_sh + wave * kShHalfSlab;
    const int row0 = blockIdx.x * kBlock + wave * 64;           // first Gaussian of this wavefront
    const int rows = min(64, prm.P - row0);
    const int nb = (prm.sh_degree + 1) * (prm.sh_degree + 1);
    float B[16];
    float dx = 0.f, dy = 0.f, dz = 1.f;
    if (want_j && i < prm.P) sh_view_dir(means3D, i, campos, dx, dy, dz);
    if (M3 == 48 && rows == 64) {                              // wave-uniform: a full wavefront of full rows
        const float *src = shs + (size_t)48 * row0;
        auto pass = [&](const ShRegs<kShPassRows> &h, int p) {
            __builtin_amdgcn_wave_barrier();
            sh_rows48_to_slab<kShPassRows>(slab, h, lane);
            __builtin_amdgcn_wave_barrier();
            if (lane / kShPassRows == p) {
                if (want_j) cj = sh_accumulate_jacobian(B, dx, dy, dz, slab + (lane % kShPassRows) * kShRow, nb, cj);
                else sh_accumulate(B, slab + (lane % kShPassRows) * kShRow, 0, 16, nb, acc);
            }
        };
        if constexpr (kShPasses == 2) {
            // both halves' loads are issued up front (the second half waits in registers while the first is evaluated)
            const ShRegs<kShPassRows> h0 = sh_rows48_load<kShPassRows>(src, lane);
            const ShRegs<kShPassRows> h1 = sh_rows48_load<kShPassRows>(src + 48 * kShPassRows, lane);
            sh_view_basis(prm, means3D, i, campos, B);
            pass(h0, 0);
            pass(h1, 1);
        } else {
            static_assert(kShPasses == 2 || kShPasses == 4, "two or four passes");
            // two quarters' loads in flight at a time: quarter p + 2 is requested when quarter p has been evaluated
            const ShRegs<kShPassRows> h0 = sh_rows48_load<kShPassRows>(src, lane);
            const ShRegs<kShPassRows> h1 = sh_rows48_load<kShPassRows>(src + 48 * kShPassRows, lane);
            sh_view_basis(prm, means3D, i, campos, B);
            pass(h0, 0);
            const ShRegs<kShPassRows> h2 = sh_rows48_load<kShPassRows>(src + 48 * kShPassRows * 2, lane);
            pass(h1, 1);
            const ShRegs<kShPassRows> h3 = sh_rows48_load<kShPassRows>(src + 48 * kShPassRows * 3, lane);
            pass(h2, 2);
            pass(h3, 3);
        }
        return;
    }
    if (i < prm.P) sh_view_basis(prm, means3D, i, campos, B);
    for (int h = 0; h < kShPasses; ++h) {
        const int r = min(kShPassRows, rows - kShPassRows * h);
        __builtin_amdgcn_wave_barrier();
        if (r > 0) sh_slab_load(slab, shs + (size_t)M3 * (row0 + kShPassRows * h), r, M3, lane);
        __builtin_amdgcn_wave_barrier();
        if (i < prm.P && lane / kShPassRows == h) {
            if (want_j) cj = sh_accumulate_jacobian(B, dx, dy, dz, slab + (lane % kShPassRows) * kShRow, nb, cj);
            else sh_accumulate(B, slab + (lane % kShPassRows) * kShRow, 0, 16, nb, acc);
        }
    }
}

// what preprocess leaves per Gaussian (GeomBuf records of ITS view, radius); returns the tile rectangle in the batch's grid
struct TileRect { bool visible; int r0, r1, r2, r3; };
__device__ __forceinline__ TileRect write_geom_records(const GeomBuf &geom, int i, const PreOut &o, bool keep_cov, int tile_row0,
                                                       int32_t *__restrict__ radii, bool want_j, const ShColJ &cj) {
    const Splat &sp = o.sp;
    if (keep_cov) {                         // (uniform) a precomputed covariance is read again from the caller's tensor
#pragma unroll
        for (int k = 0; k < 6; ++k) geom.cov3D[6 * (size_t)i + k] = o.c6[k];
    }
    radii[i] = sp.radius;
    geom.depth[i] = sp.depth;
    // culled Gaussians keep an EMPTY rectangle: the scatter pass and the backward test visibility through it
    geom.rect[i] = sp.visible ? make_uint2((uint32_t)sp.rect[0] | ((uint32_t)(sp.rect[1] + tile_row0) << 16),
                                           (uint32_t)sp.rect[2] | ((uint32_t)(sp.rect[3] + tile_row0) << 16))
                              : make_uint2(0u, 0u);
    geom.conic_o[i] = make_float4(sp.conic[0], sp.conic[1], sp.conic[2], o.opacity);
    {   // half extents of the splat's alpha >= 1/255 box, by the function the compositing stage culls with (bit-identical)
        const SplatCull sc = splat_cull(sp.conic[0], sp.conic[1], sp.conic[2], o.opacity);
        geom.xyh[i] = make_float4(sp.px, sp.py, sp.visible ? sc.hx : -1.0f, sc.hy);
    }
    geom.rgb_invd[i] = make_float4(o.rgb[0], o.rgb[1], o.rgb[2], sp.visible ? 1.0f / sp.depth : 0.f);
    geom.clamped[i] = o.clampmask;
    if (want_j) {
        const float jv[9] = {cj.j0, cj.j1, cj.j2, cj.j3, cj.j4, cj.j5, cj.j6, cj.j7, cj.j8};
#pragma unroll
        for (int k = 0; k < 9; ++k) D3GA_DCOL_AT(geom.dcol, i, k, geom.dcol_stride) = jv[k];
    }
    return TileRect{sp.visible, sp.rect[0], sp.rect[1] + tile_row0, sp.rect[2], sp.rect[3] + tile_row0};
}

// tile histogram (counting-sort pass 1) of this block's Gaussians through its LDS window.  Every thread of the block must call it
// (barriers inside); s_cnt: kWinTiles words that nobody else uses between the call's first and last barrier.
__device__ __forceinline__ void tile_histogram(int *s_box, uint32_t *s_cnt, const TileRect &t, int gx, uint32_t *__restrict__ tile_count,
                                               uint32_t *__restrict__ counters) {
    const int tid = threadIdx.x;
    const TileWindow win = block_tile_window(s_box, t.visible, t.r0, t.r1, t.r2, t.r3);   // barriers inside
    const int nvis = __syncthreads_count(t.visible);
    if (tid == 0 && nvis) atomicAdd(&counters[D3GA_CNT_VISIBLE], (uint32_t)nvis);
    const int area = win.area();
    if (area == 0) return;                                   // uniform
    if (win.fits()) {
        for (int k = tid; k < area; k += kBlock) s_cnt[k] = 0;
        __syncthreads();
        if (t.visible)
            for (int ty = t.r1; ty < t.r3; ++ty)
                for (int tx = t.r0; tx < t.r2; ++tx) atomicAdd(&s_cnt[(ty - win.y0) * win.w + (tx - win.x0)], 1u);
        __syncthreads();
        const float inv_w = 1.0f / (float)win.w;
        for (int k = tid; k < area; k += kBlock) {
            const uint32_t c = s_cnt[k];
            if (c) atomicAdd(&tile_count[win.tile_of(k, gx, inv_w)], c);
        }
    } else if (t.visible) {                                  // huge footprints: straight to global memory
        for (int ty = t.r1; ty < t.r3; ++ty)
            for (int tx = t.r0; tx < t.r2; ++tx) atomicAdd(&tile_count[ty * gx + tx], 1u);
    }
}

template <bool WANT_J>      // WANT_J: a backward will follow (forward_only == 0) and the SH coefficients are staged: leave d(colour)/d(direction) for it
__global__ __launch_bounds__(kBlock) void preprocess_kernel(
    d3ga_raster_params prm, const float *__restrict__ means3D, const float *__restrict__ shs,
    const float *__restrict__ colors_precomp, const float *__restrict__ opacities, const float *__restrict__ scales,
    const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp, const float *__restrict__ viewmatrix,
    const float *__restrict__ projmatrix, const float *__restrict__ campos, GeomBuf geom,
    uint32_t *__restrict__ tile_count, uint32_t *__restrict__ counters, int32_t *__restrict__ radii,
    int tile_row0 /* view-batched renders: this view's first tile row in the batch's grid (d3ga.h: n_views); else 0 */) {
    if (!(prm.tanfovx > 0.f)) { prm.tanfovx = campos[3]; prm.tanfovy = campos[4]; }     // camera slot: see d3ga.h
    // one dynamic LDS region, used first as the SH staging slabs and then (after a barrier) as the tile window
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_box[4];
    float *s_sh = reinterpret_cast<float *>(smem);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem);
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    const int M3 = 3 * prm.M;
    const bool staged = shs != nullptr && sh_staged(prm.M);
    float acc[3] = {0.f, 0.f, 0.f};
    // covariance row and opacity: in flight while the SH rows are staged
    float pc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, pop = 0.f;
    const bool pre = staged && cov3D_precomp != nullptr;
    if (pre && i < prm.P) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pc6[k] = cov3D_precomp[6 * (size_t)i + k];
        pop = opacities[i];
    }
    ShColJ cj = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr bool want_j = WANT_J;                                 // (the launcher: staged && !forward_only)
    if (staged) {
        staged_sh_colour<WANT_J>(prm, means3D, shs, campos, s_sh, acc, cj);
        if (want_j) { acc[0] = cj.a0; acc[1] = cj.a1; acc[2] = cj.a2; }
        __syncthreads();                                            // the region becomes the tile window below
    }
    TileRect tr = {false, 0, 0, 0, 0};
#ifdef D3GA_DIAG
    if (prm.debug & 0x200) { if (acc[0] == 12345.f) radii[0] = 1; return; }     // diag: SH staging only
#endif
    if (i < prm.P) {
        // two call sites so that each inlined copy sees ONE address space (registers vs global_load, never flat)
        PreLoaded pl;
        pl.has_sh = staged; pl.has_c6 = pre;
        pl.sh[0] = acc[0]; pl.sh[1] = acc[1]; pl.sh[2] = acc[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) pl.c6[k] = pc6[k];
        pl.op = pop;
        const PreOut o = staged ? preprocess_one(prm, i, means3D, nullptr, colors_precomp, opacities, scales, rotations,
                                                 cov3D_precomp, viewmatrix, projmatrix, campos, pl)
                                : preprocess_one(prm, i, means3D, shs ? shs + (size_t)M3 * i : nullptr, colors_precomp,
                                                 opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                                 campos);
        tr = write_geom_records(geom, i, o, !cov3D_precomp, tile_row0, radii, want_j, cj);
    }
#ifdef D3GA_DIAG
    if (prm.debug & 0x100) return;                                              // diag: no histogram
#endif
    tile_histogram(s_box, s_cnt, tr, (prm.W + kTile - 1) / kTile, tile_count, counters);      // (barriers inside: the slabs are dead now)
}

// The same for KV views of ONE set of Gaussians in one pass (view-batched renders with shared geometry, d3ga.h: n_views): the
// Gaussian's mean, covariance, opacity and -- the bulk of the kernel's traffic, 12 M bytes -- its SH row are read ONCE; per view the
// colour (and its direction Jacobian) is formed from the staged row with that view's direction, the projection runs, the records go
// to the view's section of the batch's buffers and the view's histogram pass follows.  Per view the arithmetic is that of
// preprocess_kernel (same inlined functions, same order): bit-identical records.  views: KV cameras starting at `view0`.
template <int KV>
struct ViewCams { const float *vm[KV], *pm[KV], *cp[KV]; };
template <bool WANT_J, int KV>
__device__ __forceinline__ void staged_sh_colour_views(const d3ga_raster_params &prm, const float *__restrict__ means3D, size_t pv,
                                                       const float *__restrict__ shs, const ViewCams<KV> &cams, float *s_sh,
                                                       ShColJ (&cj)[KV]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * kBlock + tid;
    const int M3 = 3 * prm.M;
    float *slab = s_sh + wave * kShHalfSlab;
    const int row0 = blockIdx.x * kBlock + wave * 64;
    const int rows = min(64, prm.P - row0);
    const int nb = (prm.sh_degree + 1) * (prm.sh_degree + 1);
    auto eval = [&](const float *row) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            float B[16];
            const float *mv = means3D + 3 * pv * v;                  // (a batch of frames: every view's own means)
            if (WANT_J) {
                float dx, dy, dz;
                sh_view_dir(mv, i, cams.cp[v], dx, dy, dz);
                sh_basis(prm.sh_degree, dx, dy, dz, B);
                cj[v] = sh_accumulate_jacobian(B, dx, dy, dz, row, nb, cj[v]);
            } else {
                sh_view_basis(prm, mv, i, cams.cp[v], B);
                float acc[3] = {cj[v].a0, cj[v].a1, cj[v].a2};
                sh_accumulate(B, row, 0, 16, nb, acc);
                cj[v].a0 = acc[0]; cj[v].a1 = acc[1]; cj[v].a2 = acc[2];
            }
        }
    };
    if (M3 == 48 && rows == 64) {                              // wave-uniform: a full wavefront of full rows
        static_assert(kShPasses == 2, "two passes");
        const float *src = shs + (size_t)48 * row0;
        const ShRegs<kShPassRows> h0 = sh_rows48_load<kShPassRows>(src, lane);
        const ShRegs<kShPassRows> h1 = sh_rows48_load<kShPassRows>(src + 48 * kShPassRows, lane);
        __builtin_amdgcn_wave_barrier();
        sh_rows48_to_slab<kShPassRows>(slab, h0, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane / kShPassRows == 0) eval(slab + (lane % kShPassRows) * kShRow);
        __builtin_amdgcn_wave_barrier();
        sh_rows48_to_slab<kShPassRows>(slab, h1, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane / kShPassRows == 1) eval(slab + (lane % kShPassRows) * kShRow);
        return;
    }
    for (int h = 0; h < kShPasses; ++h) {
        const int r = min(kShPassRows, rows - kShPassRows * h);
        __builtin_amdgcn_wave_barrier();
        if (r > 0) sh_slab_load(slab, shs + (size_t)M3 * (row0 + kShPassRows * h), r, M3, lane);
        __builtin_amdgcn_wave_barrier();
        if (i < prm.P && lane / kShPassRows == h) eval(slab + (lane % kShPassRows) * kShRow);
    }
}

template <bool WANT_J, int KV>
__global__ __launch_bounds__(kBlock) void preprocess_views_kernel(
    d3ga_raster_params prm, const float *__restrict__ means3D, const float *__restrict__ shs,
    const float *__restrict__ colors_precomp, const float *__restrict__ opacities, const float *__restrict__ scales,
    const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp, ViewCams<KV> cams, GeomBuf geom /* of the first view */,
    uint32_t *__restrict__ tile_count, uint32_t *__restrict__ counters, int32_t *__restrict__ radii /* of the first view */,
    int tile_row0, int gyv, size_t pv /* records between the views' geometry: 0 = k cameras of one set of Gaussians, P = a batch of frames */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_box[4];
    float *s_sh = reinterpret_cast<float *>(smem);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem);
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    const bool slot = !(prm.tanfovx > 0.f);                     // camera slots: the tangents ride behind every view's position
    // shared by the views: covariance row and opacity (in flight while the SH rows are staged)
    float pc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, pop = 0.f;
    const bool pre = cov3D_precomp != nullptr && pv == 0;
    if (pre && i < prm.P) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pc6[k] = cov3D_precomp[6 * (size_t)i + k];
        pop = opacities[i];
    }
    ShColJ cj[KV];
#pragma unroll
    for (int v = 0; v < KV; ++v) cj[v] = ShColJ{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    staged_sh_colour_views<WANT_J, KV>(prm, means3D, pv, shs, cams, s_sh, cj);
    const int gx = (prm.W + kTile - 1) / kTile;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
        __syncthreads();                                        // the LDS region changes hands: slabs -> window, window -> window
        TileRect tr = {false, 0, 0, 0, 0};
        if (i < prm.P) {
            d3ga_raster_params pvw = prm;
            if (slot) { pvw.tanfovx = cams.cp[v][3]; pvw.tanfovy = cams.cp[v][4]; }
            PreLoaded pl;
            pl.has_sh = true; pl.has_c6 = pre;
            pl.sh[0] = cj[v].a0; pl.sh[1] = cj[v].a1; pl.sh[2] = cj[v].a2;
#pragma unroll
            for (int k = 0; k < 6; ++k) pl.c6[k] = pc6[k];
            pl.op = pop;
            const size_t og = pv * v;
            const PreOut o = preprocess_one(pvw, i, means3D + 3 * og, nullptr, colors_precomp, opacities, scales ? scales + 3 * og : nullptr,
                                            rotations ? rotations + 4 * og : nullptr, cov3D_precomp ? cov3D_precomp + 6 * og : nullptr,
                                            cams.vm[v], cams.pm[v], cams.cp[v], pl);
            tr = write_geom_records(geom_view(geom, prm.P, v), i, o, !cov3D_precomp, tile_row0 + v * gyv, radii + (size_t)prm.P * v, WANT_J, cj[v]);
        }
        tile_histogram(s_box, s_cnt, tr, gx, tile_count, counters);
    }
}

// Second render of the SAME geometry with other colours (the reference's training step renders RGB and a silhouette
// pass from one garment_pkg, models/trainer.py:102-110): copies the geometry records of `src` to `dst` and evaluates
// only the colour (SH of this camera or colors_precomp) -- the projection, the tile histogram and the whole binning
// stage of the second pass disappear; `dst` then shares the first pass's binning buffer.
template <bool WANT_J>
__global__ __launch_bounds__(kBlock) void recolor_kernel(d3ga_raster_params prm, const float *__restrict__ means3D,
                                                         const float *__restrict__ shs,
                                                         const float *__restrict__ colors_precomp,
                                                         const float *__restrict__ campos, GeomBuf src, GeomBuf dst) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_sh = reinterpret_cast<float *>(smem);
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int M3 = 3 * prm.M;
    const bool staged = shs != nullptr && sh_staged(prm.M);
    float acc[3] = {0.f, 0.f, 0.f};
    ShColJ cj = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr bool want_j = WANT_J;
    if (staged) staged_sh_colour<WANT_J>(prm, means3D, shs, campos, s_sh, acc, cj);
    if (i >= prm.P) return;
    if (want_j) {
        acc[0] = cj.a0; acc[1] = cj.a1; acc[2] = cj.a2;
        const float jv[9] = {cj.j0, cj.j1, cj.j2, cj.j3, cj.j4, cj.j5, cj.j6, cj.j7, cj.j8};
#pragma unroll
        for (int k = 0; k < 9; ++k) D3GA_DCOL_AT(dst.dcol, i, k, dst.dcol_stride) = jv[k];
    }
    const uint2 rc = src.rect[i];
    const bool visible = ((rc.y & 0xffffu) > (rc.x & 0xffffu)) && ((rc.y >> 16) > (rc.x >> 16));
    const float depth = src.depth[i];
    dst.depth[i] = depth;
    dst.conic_o[i] = src.conic_o[i];
    dst.xyh[i] = src.xyh[i];
    dst.rect[i] = rc;
#pragma unroll
    for (int k = 0; k < 6; ++k) dst.cov3D[6 * (size_t)i + k] = src.cov3D[6 * (size_t)i + k];
    float rgb[3] = {0.f, 0.f, 0.f};
    uint8_t mask = 0;
    if (visible) {
        if (colors_precomp) {
            rgb[0] = colors_precomp[3 * (size_t)i]; rgb[1] = colors_precomp[3 * (size_t)i + 1];
            rgb[2] = colors_precomp[3 * (size_t)i + 2];
        } else {
            if (!staged) {
                float B[16];
                sh_view_basis(prm, means3D, i, campos, B);
                sh_accumulate(B, shs + (size_t)M3 * i, 0, 16, (prm.sh_degree + 1) * (prm.sh_degree + 1), acc);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = acc[c] + 0.5f;
                if (v < 0.f) mask |= (uint8_t)(1u << c);
                rgb[c] = fmaxf(v, 0.f);
            }
        }
    }
    dst.rgb_invd[i] = make_float4(rgb[0], rgb[1], rgb[2], visible ? 1.0f / depth : 0.f);
    dst.clamped[i] = mask;
}

__global__ __launch_bounds__(kBlock) void preprocess_bwd_kernel(
    d3ga_raster_params prm, const float *__restrict__ means3D, const float *__restrict__ shs,
    const float *__restrict__ scales, const float *__restrict__ rotations, const float *__restrict__ viewmatrix,
    const float *__restrict__ projmatrix, const float *__restrict__ campos, GeomBuf geom,
    const float *__restrict__ acc, float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D,
    float *__restrict__ dL_dopacity, float *__restrict__ dL_dsh, float *__restrict__ dL_dcolors,
    float *__restrict__ dL_dcov3D, float *__restrict__ dL_dscales, float *__restrict__ dL_drots,
    const float *__restrict__ cov3D_precomp, int accum /* views > 0 of a batch: add to the gradients of inputs the views share (raster_pre_body.h) */) {
    if (!(prm.tanfovx > 0.f)) { prm.tanfovx = campos[3]; prm.tanfovy = campos[4]; }     // camera slot: see d3ga.h
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_sh = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * kBlock + tid;
    const int M3 = 3 * prm.M;
    const bool staged = shs != nullptr && sh_staged(prm.M);      // dL_dsh may be null (factored SH gradient)
    const int row0 = blockIdx.x * kBlock + wave * 64;
    const int rows = min(64, prm.P - row0);
    float *slab = s_sh + wave * kShSlab;
    const bool full48 = M3 == 48 && rows == 64;                 // wave-uniform
    // This Gaussian's records first, unconditionally (an invisible one has an all-zero accumulator row and a valid
    // covariance row): they are in flight together with the SH rows instead of two dependent round trips behind them.
    uint2 rc = make_uint2(0u, 0u);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    float a[12], c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint8_t clampmask = 0;
    float act_opacity = 0.f;
    if (i < prm.P) {
        rc = geom.rect[i];
        const float4 *ap = reinterpret_cast<const float4 *>(acc + D3GA_ACC_STRIDE * (size_t)i);
        a0 = ap[0]; a1 = ap[1]; a2 = ap[2];
        if (prm.acc_self_clearing) {
            // the caller keeps this buffer for the next backward: leave the record as we found the untouched ones, all zero
            // (only records the compositing backward wrote are written back: 48 of their 64 bytes)
            const uint32_t any = (__float_as_uint(a0.x) | __float_as_uint(a0.y) | __float_as_uint(a0.z) | __float_as_uint(a0.w)) |
                                 (__float_as_uint(a1.x) | __float_as_uint(a1.y) | __float_as_uint(a1.z) | __float_as_uint(a1.w)) |
                                 (__float_as_uint(a2.x) | __float_as_uint(a2.y) | __float_as_uint(a2.z) | __float_as_uint(a2.w));
            if (any) {
                float4 *wp = const_cast<float4 *>(ap);
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                wp[0] = z; wp[1] = z; wp[2] = z;
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = (cov3D_precomp ? cov3D_precomp : geom.cov3D)[6 * (size_t)i + k];   // (the forward kept no copy of a precomputed one)
        clampmask = geom.clamped[i];
        act_opacity = geom.conic_o[i].w;
    }
    // Round 5: the forward left d(colour)/d(direction) of every Gaussian (geom.dcol, 36 bytes) when it staged the coefficients and
    // a backward was to follow: the coefficients themselves (192 bytes per Gaussian) are then not read here at all -- the slab
    // only collects the gradient rows for the coalesced store.
    const bool have_j = D3GA_PRE_DCOL && staged && !prm.forward_only;
    ShColJ jd = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (have_j && i < prm.P) {
        const float *d = geom.dcol;
        jd.j0 = D3GA_DCOL_AT(d, i, 0, geom.dcol_stride); jd.j1 = D3GA_DCOL_AT(d, i, 1, geom.dcol_stride); jd.j2 = D3GA_DCOL_AT(d, i, 2, geom.dcol_stride);
        jd.j3 = D3GA_DCOL_AT(d, i, 3, geom.dcol_stride); jd.j4 = D3GA_DCOL_AT(d, i, 4, geom.dcol_stride); jd.j5 = D3GA_DCOL_AT(d, i, 5, geom.dcol_stride);
        jd.j6 = D3GA_DCOL_AT(d, i, 6, geom.dcol_stride); jd.j7 = D3GA_DCOL_AT(d, i, 7, geom.dcol_stride); jd.j8 = D3GA_DCOL_AT(d, i, 8, geom.dcol_stride);
    }
    if (staged && !have_j) {
        if (full48) sh_rows48_to_slab<64>(slab, sh_rows48_load<64>(shs + (size_t)48 * row0, lane), lane);
        else if (rows > 0) sh_slab_load(slab, shs + (size_t)M3 * row0, rows, M3, lane);
        __builtin_amdgcn_wave_barrier();          // the slab is private to the wavefront: program order suffices
    }
    if (i < prm.P) {
        const bool visible = ((rc.y & 0xffffu) > (rc.x & 0xffffu)) && ((rc.y >> 16) > (rc.x >> 16));
        a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
        a[8] = a2.x; a[9] = a2.y; a[10] = a2.z; a[11] = a2.w;
        if (!visible) {
#pragma unroll
            for (int k = 0; k < 12; ++k) a[k] = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) c6[k] = 0.f;
            act_opacity = 0.f;
        }
        // two call sites so that each inlined copy sees ONE address space (LDS row vs global row, never flat);
        // staged: the gradient row overwrites the coefficient row in place
        if (staged)
            preprocess_bwd_one(prm, i, visible, means3D, slab + lane * kShRow, scales, rotations, viewmatrix, projmatrix,
                               campos, c6, clampmask, a, dL_dmeans3D, dL_dmeans2D, dL_dopacity,
                               dL_dsh ? slab + lane * kShRow : nullptr, dL_dcolors, dL_dcov3D, dL_dscales, dL_drots,
                               act_opacity, have_j, jd, accum);
        else
            preprocess_bwd_one(prm, i, visible, means3D, shs ? shs + (size_t)M3 * i : nullptr, scales, rotations,
                               viewmatrix, projmatrix, campos, c6, clampmask, a, dL_dmeans3D, dL_dmeans2D,
                               dL_dopacity, dL_dsh ? dL_dsh + (size_t)M3 * i : nullptr, dL_dcolors, dL_dcov3D,
                               dL_dscales, dL_drots,
                               act_opacity, false, ShColJ(), accum);
    }
    if (staged && dL_dsh) {
        __builtin_amdgcn_wave_barrier();          // the slab is private to the wavefront: program order suffices
        if (full48) sh_rows48_store<64>(slab, dL_dsh + (size_t)48 * row0, lane);
        else if (rows > 0) sh_slab_store(slab, dL_dsh + (size_t)M3 * row0, rows, M3, lane);
    }
}

// R6 for the views of a batch that share their geometry (d3ga.h: n_views without per_view_geometry), ONE pass: thread i walks the kv
// views of Gaussian i -- per view its accumulator record, rectangle, clamp mask, stored opacity and d(colour)/d(direction) --, sums
// the gradients of the shared inputs in registers and writes every output ONCE: no read-modify-write of dL/dmeans3D / dL/dcov /
// dL/dopacity per view, no (P,3) factors through memory, the (P,M,3) SH row formed here (sum_v Y(dir_v) (x) g_v, as
// sh_grad_from_views_kernel forms it) instead of by a pass of its own.  Same per-view arithmetic as preprocess_bwd_one (cov2d_bwd,
// project_bwd, the direction term from the forward's Jacobian), views added in ascending order like the per-view launches add them.
// SH colours need the forward's dcol planes (staged coefficients, forward_only == 0): the launcher falls back to per-view launches
// otherwise, and for the factored SH output of the camera-sharded exchange.  accum: a later group of a batch of more than kMaxGroup
// views -- every output is added to.
constexpr int kMaxGroup = 8;
struct ViewCamsN { const float *vm[kMaxGroup], *pm[kMaxGroup], *cp[kMaxGroup]; };
__global__ __launch_bounds__(kBlock) void preprocess_bwd_views_kernel(
    d3ga_raster_params prm, int kv, const float *__restrict__ means3D, bool sh_path, const float *__restrict__ scales,
    const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp, ViewCamsN cams, GeomBuf geom /* first view of the group */,
    const float *__restrict__ acc /* first view of the group */, float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D /* first view | null */,
    float *__restrict__ dL_dopacity, float *__restrict__ dL_dsh, float *__restrict__ dL_dcolors, float *__restrict__ dL_dcov3D,
    float *__restrict__ dL_dscales, float *__restrict__ dL_drots, bool accum,
    size_t pv /* records between the views' geometry AND geometry gradients: 0 = shared (summed), P = a batch of frames (written per view) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_sh = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * kBlock + tid;
    const int M3 = 3 * prm.M;
    const int row0 = blockIdx.x * kBlock + wave * 64;
    const int rows = min(64, prm.P - row0);
    float *slab = s_sh + wave * kShSlab;
    const bool slot = !(prm.tanfovx > 0.f);
    const int nb = (prm.sh_degree + 1) * (prm.sh_degree + 1);
    if (i < prm.P) {
        V3 mean = ld3(means3D, i);
        float c6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = (cov3D_precomp ? cov3D_precomp : geom.cov3D)[6 * (size_t)i + k];   // (from scale / rotation: view 0's record -- the same in every view of shared geometry)
        float gmean[3] = {0.f, 0.f, 0.f}, g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gop = 0.f, gcol[3] = {0.f, 0.f, 0.f};
        auto put = [&](float *p, float v, bool add) { *p = add ? *p + v : v; };
        auto geometry_out = [&](size_t og, const float (&gm)[3], const float (&g)[6], bool vis, bool add) {     // og: record offset of the geometry gradients
            for (int k = 0; k < 3; ++k) put(dL_dmeans3D + 3 * (og + i) + k, gm[k], add);
            if (dL_dcov3D)
                for (int k = 0; k < 6; ++k) put(dL_dcov3D + 6 * (og + i) + k, g[k], add);
            if (dL_dscales && dL_drots) {
                float gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
                if (vis) {
                    const float sc[3] = {scales[3 * (og + i)], scales[3 * (og + i) + 1], scales[3 * (og + i) + 2]};
                    const float q[4] = {rotations[4 * (og + i)], rotations[4 * (og + i) + 1], rotations[4 * (og + i) + 2], rotations[4 * (og + i) + 3]};
                    cov3d_from_scale_rot_bwd(sc, prm.scale_modifier, q, g, gs, gq);      // (linear in g: a sum over views goes through once)
                }
                for (int k = 0; k < 3; ++k) put(dL_dscales + 3 * (og + i) + k, gs[k], add);
                for (int k = 0; k < 4; ++k) put(dL_drots + 4 * (og + i) + k, gq[k], add);
            }
        };
        const float zero3[3] = {0.f, 0.f, 0.f}, zero6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float out[48];
#pragma unroll
        for (int k = 0; k < 48; ++k) out[k] = 0.f;
        for (int v = 0; v < kv; ++v) {
            const size_t j = (size_t)prm.P * v + i;                          // this view's record
            const uint2 rc = geom.rect[j];
            const float4 *ap = reinterpret_cast<const float4 *>(acc + D3GA_ACC_STRIDE * j);
            const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
            if (prm.acc_self_clearing) {
                const uint32_t any = (__float_as_uint(a0.x) | __float_as_uint(a0.y) | __float_as_uint(a0.z) | __float_as_uint(a0.w)) |
                                     (__float_as_uint(a1.x) | __float_as_uint(a1.y) | __float_as_uint(a1.z) | __float_as_uint(a1.w)) |
                                     (__float_as_uint(a2.x) | __float_as_uint(a2.y) | __float_as_uint(a2.z) | __float_as_uint(a2.w));
                if (any) {
                    float4 *wp = const_cast<float4 *>(ap);
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    wp[0] = z; wp[1] = z; wp[2] = z;
                }
            }
            const bool visible = ((rc.y & 0xffffu) > (rc.x & 0xffffu)) && ((rc.y >> 16) > (rc.x >> 16));
            if (dL_dmeans2D) {                                                // screen-space: per view
                float *m2 = dL_dmeans2D + 3 * j;
                m2[0] = visible ? a0.x : 0.f; m2[1] = visible ? a0.y : 0.f; m2[2] = 0.f;
            }
            if (!visible) {
                if (pv) geometry_out(pv * v, zero3, zero6, false, false);      // a batch of frames: every view's geometry gradients are written
                continue;
            }
            if (pv) {                                                        // this view's own geometry
                mean = ld3(means3D + 3 * pv * v, i);
#pragma unroll
                for (int k = 0; k < 6; ++k) c6[k] = (cov3D_precomp ? cov3D_precomp + 6 * pv * v : geom.cov3D + 6 * (size_t)prm.P * v)[6 * (size_t)i + k];
            }
            const uint8_t clampmask = geom.clamped[j];
            const float act_opacity = geom.conic_o[j].w;
            const float a[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
            const float *vm = cams.vm[v], *pm = cams.pm[v], *cp = cams.cp[v];
            const float tfx = slot ? cp[3] : prm.tanfovx, tfy = slot ? cp[4] : prm.tanfovy;
            float gm_v[3] = {0.f, 0.f, 0.f}, g6_v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, aa = 1.0f;
            cov2d_bwd(mean, c6, vm, prm.W, prm.H, tfx, tfy, a[3], a[4], a[5], g6_v, gm_v, prm.antialiasing != 0, a[6], act_opacity, &aa);
            project_bwd(mean, pm, a[0], a[1], gm_v);
            const float z = vm[2] * mean.x + vm[6] * mean.y + vm[10] * mean.z + vm[14];
            const float gz = -a[10] / (z * z);
            gm_v[0] += vm[2] * gz; gm_v[1] += vm[6] * gz; gm_v[2] += vm[10] * gz;
            if (sh_path) {
                const float gr[3] = {(clampmask & 1) ? 0.f : a[7], (clampmask & 2) ? 0.f : a[8], (clampmask & 4) ? 0.f : a[9]};
                const V3 d0 = mean - v3(cp[0], cp[1], cp[2]);
                const float inv = 1.0f / sqrtf(dot(d0, d0));
                float B[16];
                sh_basis(prm.sh_degree, d0.x * inv, d0.y * inv, d0.z * inv, B);
                const float *d = geom.dcol;
                const float j0 = D3GA_DCOL_AT(d, j, 0, geom.dcol_stride), j1 = D3GA_DCOL_AT(d, j, 1, geom.dcol_stride), j2 = D3GA_DCOL_AT(d, j, 2, geom.dcol_stride);
                const float j3 = D3GA_DCOL_AT(d, j, 3, geom.dcol_stride), j4 = D3GA_DCOL_AT(d, j, 4, geom.dcol_stride), j5 = D3GA_DCOL_AT(d, j, 5, geom.dcol_stride);
                const float j6 = D3GA_DCOL_AT(d, j, 6, geom.dcol_stride), j7 = D3GA_DCOL_AT(d, j, 7, geom.dcol_stride), j8 = D3GA_DCOL_AT(d, j, 8, geom.dcol_stride);
                const V3 gd = v3(j0 * gr[0] + j1 * gr[1] + j2 * gr[2], j3 * gr[0] + j4 * gr[1] + j5 * gr[2], j6 * gr[0] + j7 * gr[1] + j8 * gr[2]);
                const V3 gm = normalize_bwd(d0, gd);
                gm_v[0] += gm.x; gm_v[1] += gm.y; gm_v[2] += gm.z;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < nb) { out[3 * k] += B[k] * gr[0]; out[3 * k + 1] += B[k] * gr[1]; out[3 * k + 2] += B[k] * gr[2]; }
            } else {
                gcol[0] += a[7]; gcol[1] += a[8]; gcol[2] += a[9];
            }
            {
                const float op = act_opacity / aa, g_op = a[6] * aa;
                gop += prm.opacity_activation == D3GA_OPACITY_SIGMOID ? g_op * op * (1.0f - op) : g_op;
            }
            if (pv) geometry_out(pv * v, gm_v, g6_v, true, false);
            else {
                gmean[0] += gm_v[0]; gmean[1] += gm_v[1]; gmean[2] += gm_v[2];
#pragma unroll
                for (int k = 0; k < 6; ++k) g6[k] += g6_v[k];
            }
        }
        if (!pv) geometry_out(0, gmean, g6, true, accum);
        if (dL_dopacity) put(dL_dopacity + i, gop, accum);
        if (!sh_path && dL_dcolors) { put(dL_dcolors + 3 * (size_t)i, gcol[0], accum); put(dL_dcolors + 3 * (size_t)i + 1, gcol[1], accum); put(dL_dcolors + 3 * (size_t)i + 2, gcol[2], accum); }
        if (sh_path && dL_dsh) {
            float *row = slab + lane * kShRow;
#pragma unroll
            for (int k = 0; k < 48; ++k)
                if (k < M3) row[k] = out[k];
        }
    }
    if (sh_path && dL_dsh) {
        __builtin_amdgcn_wave_barrier();          // the slab is private to the wavefront: program order suffices
        if (accum) {                              // a later group of a large batch: add to what the earlier groups wrote
            const int nvec = rows > 0 ? rows * M3 / 4 : 0;
            float4 *dst = reinterpret_cast<float4 *>(dL_dsh + (size_t)M3 * row0);
            for (int v = lane; v < nvec; v += 64) {
                const int e = 4 * v, r = e / M3, c = e - r * M3;
                const float4 x = *reinterpret_cast<const float4 *>(slab + r * kShRow + c);
                float4 y = dst[v];
                y.x += x.x; y.y += x.y; y.z += x.z; y.w += x.w;
                dst[v] = y;
            }
        } else if (M3 == 48 && rows == 64) sh_rows48_store<64>(slab, dL_dsh + (size_t)48 * row0, lane);
        else if (rows > 0) sh_slab_store(slab, dL_dsh + (size_t)M3 * row0, rows, M3, lane);
    }
}

// View-sharded training: the SH gradient of one view is rank-1 per Gaussian,
//   dL/dsh[i][k][c] = basis_k(normalize(mean_i - campos_v)) * g_v[i][c],      g_v = clamp-masked dL/dcolour,
// so the ranks exchange the (P,3) factor g_v (all-gather) instead of summing (P,M,3) blocks (all-reduce) and every rank
// rebuilds   dL_dsh = scale * sum_v basis(dir_v) (x) g_v   here.  One thread per Gaussian, 48 accumulators in
// registers, rows leave through the per-wavefront LDS slab (16 B/lane contiguous stores).
__global__ __launch_bounds__(kBlock) void sh_grad_from_views_kernel(
    int P, int M, int sh_degree, int n_views, const float *__restrict__ means3D, int64_t means_stride /* floats between the views' means (0: shared) */,
    const float *__restrict__ g_views, int64_t g_stride, const float *__restrict__ campos_views, int64_t campos_stride, float scale,
    float *__restrict__ dL_dsh) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_sh = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * kBlock + tid;
    const int M3 = 3 * M;
    const bool staged = sh_staged(M);
    const int row0 = blockIdx.x * kBlock + wave * 64;
    const int rows = min(64, P - row0);
    float *slab = s_sh + wave * kShSlab;
    if (i < P) {
        float out[48];
#pragma unroll
        for (int k = 0; k < 48; ++k) out[k] = 0.f;
        const int nb = (sh_degree + 1) * (sh_degree + 1);
        for (int v = 0; v < n_views; ++v) {
            const float *g = g_views + v * g_stride + 3 * (size_t)i;
            const float g0 = g[0], g1 = g[1], g2 = g[2];
            if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;        // culled / invisible / clamped in this view
            const V3 mean = ld3(means3D + v * means_stride, i);
            const float *cp = campos_views + v * campos_stride;
            const V3 d0 = mean - v3(cp[0], cp[1], cp[2]);
            const float inv = 1.0f / sqrtf(dot(d0, d0));
            float B[16];
            sh_basis(sh_degree, d0.x * inv, d0.y * inv, d0.z * inv, B);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < nb) {
                    out[3 * k] += B[k] * g0; out[3 * k + 1] += B[k] * g1; out[3 * k + 2] += B[k] * g2;
                }
        }
        float *row = staged ? slab + lane * kShRow : nullptr;
#pragma unroll
        for (int k = 0; k < 48; ++k)
            if (k < M3) {
                if (staged) row[k] = out[k] * scale;
                else dL_dsh[(size_t)M3 * i + k] = out[k] * scale;
            }
    }
    if (staged) {
        __syncthreads();
        if (rows > 0) sh_slab_store(slab, dL_dsh + (size_t)M3 * row0, rows, M3, lane);
    }
}

__global__ __launch_bounds__(kBlock) void mark_visible_kernel(int P, const float *__restrict__ means3D,
                                                              const float *__restrict__ viewmatrix,
                                                              uint8_t *__restrict__ visible) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    visible[i] = xform_point(viewmatrix, ld3(means3D, i)).z > kNear ? 1 : 0;
}

}  // namespace d3ga

using namespace d3ga;

static int validate(const d3ga_raster_params *prm) {
    if (!prm) return D3GA_E_NULL;
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0 || prm->M < 0 || prm->M > 16) return D3GA_E_SIZE;
    if (prm->sh_degree < 0 || prm->sh_degree > 3) return D3GA_E_CONFIG;
    if (prm->n_views < 0) return D3GA_E_SIZE;
    if (((prm->W + kTile - 1) / kTile) > 65535 || (int64_t)((prm->H + kTile - 1) / kTile) * n_views_of(prm) > 65535) return D3GA_E_SIZE;
    if ((int64_t)prm->P * n_views_of(prm) >= (1ll << 31)) return D3GA_E_SIZE;
    return D3GA_OK;
}

extern "C" int d3ga_raster_preprocess(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                                      const float *colors_precomp, const float *opacities, const float *scales,
                                      const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                                      const float *projmatrix, const float *campos, void *geom, void *binning,
                                      int64_t d_capacity, int32_t *radii, d3ga_stream_t stream) {
    D3GA_TRY(validate(prm));
    if (!geom || !binning || !viewmatrix || !projmatrix || !campos) return D3GA_E_NULL;
    if (d_capacity < 0) return D3GA_E_SIZE;
    hipStream_t s = (hipStream_t)stream;
    const int views = n_views_of(prm), gyv = tiles_y(prm->H);
    const int64_t tiles = (int64_t)tiles_x(prm->W) * gyv * views;
    BinBuf bin = carve_bin(binning, tiles, d_capacity);
    // counters + tile_count are adjacent: one memset
    D3GA_HIP(zero_async(bin.counters, 256 + align256(4 * tiles), s));
    if (prm->P == 0) return D3GA_OK;          // empty scene: every per-Gaussian tensor is empty (NULL)
    if ((shs != nullptr) == (colors_precomp != nullptr)) return D3GA_E_CONFIG;
    const bool sr = scales != nullptr && rotations != nullptr;
    if (sr == (cov3D_precomp != nullptr)) return D3GA_E_CONFIG;
    if (shs && (prm->sh_degree + 1) * (prm->sh_degree + 1) > prm->M) return D3GA_E_CONFIG;
    if (!means3D || !opacities || !radii) return D3GA_E_NULL;
    const GeomBuf g = carve_geom(geom, (int64_t)prm->P * views);
    const size_t win = (size_t)kWinTiles * 4;
    const size_t lds = (shs && prm->M > 0 && (3 * prm->M) % 4 == 0) ? (kShHalfLdsBytes > win ? kShHalfLdsBytes : win) : win;
    // (the kernels' `staged` condition, on the host: with it and a backward to follow the forward leaves GeomBuf::dcol)
    const bool want_j = D3GA_PRE_DCOL && shs && prm->M > 0 && (3 * prm->M) % 4 == 0 && 3 * prm->M <= 48 && !prm->forward_only;
    const int cam_stride = prm->tanfovx > 0.f ? 3 : 5;       // camera slots carry the two tangents behind the position
    // a batch of views writes view v's records at v P + i of the batch's buffers: grouped launches below when the SH row can be
    // shared, else one launch per view (a streaming kernel at the copy rate gains nothing from a taller grid)
    const size_t pv = (views > 1 && prm->per_view_geometry) ? (size_t)prm->P : 0;      // records between the views' geometry (0: shared)
    const dim3 grid((prm->P + kBlock - 1) / kBlock), block(kBlock);
    int v0 = 0;
    // k cameras of ONE set of Gaussians with staged SH colours: groups of up to four views per pass (preprocess_views_kernel: the
    // 12 M-byte coefficient row, the mean and the covariance are read once per group instead of once per view)
    const bool grouped = views > 1 && shs && prm->M > 0 && (3 * prm->M) % 4 == 0 && 3 * prm->M <= 48;
#define D3GA_PRE_VIEWS(KVV)                                                                                                        \
    do {                                                                                                                           \
        ViewCams<KVV> vc;                                                                                                          \
        for (int q = 0; q < KVV; ++q) {                                                                                            \
            vc.vm[q] = viewmatrix + 16 * (size_t)(v0 + q); vc.pm[q] = projmatrix + 16 * (size_t)(v0 + q);                          \
            vc.cp[q] = campos + (size_t)cam_stride * (v0 + q);                                                                     \
        }                                                                                                                          \
        if (want_j) hipLaunchKernelGGL((preprocess_views_kernel<true, KVV>), grid, block, lds, s, *prm, means3D + 3 * pv * v0, shs, colors_precomp,  \
                                       opacities, scales ? scales + 3 * pv * v0 : nullptr, rotations ? rotations + 4 * pv * v0 : nullptr, cov3D_precomp ? cov3D_precomp + 6 * pv * v0 : nullptr, vc, geom_view(g, prm->P, v0), bin.tile_count,  \
                                       bin.counters, radii + (size_t)prm->P * v0, v0 * gyv, gyv, pv);                              \
        else hipLaunchKernelGGL((preprocess_views_kernel<false, KVV>), grid, block, lds, s, *prm, means3D + 3 * pv * v0, shs, colors_precomp,    \
                                opacities, scales ? scales + 3 * pv * v0 : nullptr, rotations ? rotations + 4 * pv * v0 : nullptr, cov3D_precomp ? cov3D_precomp + 6 * pv * v0 : nullptr, vc, geom_view(g, prm->P, v0), bin.tile_count,         \
                                bin.counters, radii + (size_t)prm->P * v0, v0 * gyv, gyv, pv);                                     \
        v0 += KVV;                                                                                                                 \
    } while (0)
    while (grouped && views - v0 >= 2) {
        const int left = views - v0;
        if (left >= 4 && left != 5) D3GA_PRE_VIEWS(4);          // (5 = 3 + 2: no single view left over)
        else if (left == 3 || left == 5) D3GA_PRE_VIEWS(3);
        else D3GA_PRE_VIEWS(2);
    }
#undef D3GA_PRE_VIEWS
    for (int v = v0; v < views; ++v) {
        const GeomBuf gv = geom_view(g, prm->P, v);
        const float *vm = viewmatrix + 16 * (size_t)v, *pm = projmatrix + 16 * (size_t)v, *cp = campos + (size_t)cam_stride * v;
        const float *mv = means3D + 3 * pv * v, *sv = scales ? scales + 3 * pv * v : nullptr, *rq = rotations ? rotations + 4 * pv * v : nullptr;
        const float *cv = cov3D_precomp ? cov3D_precomp + 6 * pv * v : nullptr;
        int32_t *rv = radii + (size_t)prm->P * v;
        if (want_j)
            hipLaunchKernelGGL(preprocess_kernel<true>, grid, block, lds, s, *prm, mv, shs,
                               colors_precomp, opacities, sv, rq, cv, vm, pm, cp, gv, bin.tile_count, bin.counters, rv, v * gyv);
        else
            hipLaunchKernelGGL(preprocess_kernel<false>, grid, block, lds, s, *prm, mv, shs,
                               colors_precomp, opacities, sv, rq, cv, vm, pm, cp, gv, bin.tile_count, bin.counters, rv, v * gyv);
    }
    return check_launch(s, prm->debug & 0xff);
}

extern "C" int d3ga_raster_recolor(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                                   const float *colors_precomp, const float *campos, const void *geom_src,
                                   void *geom_dst, d3ga_stream_t stream) {
    D3GA_TRY(validate(prm));
    if (n_views_of(prm) > 1) return D3GA_E_CONFIG;
    if (prm->P == 0) return D3GA_OK;
    if (!geom_src || !geom_dst || geom_src == geom_dst) return D3GA_E_NULL;
    if ((shs != nullptr) == (colors_precomp != nullptr)) return D3GA_E_CONFIG;
    if (shs && (!means3D || !campos)) return D3GA_E_NULL;
    if (shs && (prm->sh_degree + 1) * (prm->sh_degree + 1) > prm->M) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const GeomBuf src = carve_geom(const_cast<void *>(geom_src), prm->P), dst = carve_geom(geom_dst, prm->P);
    const size_t lds = (shs && prm->M > 0 && (3 * prm->M) % 4 == 0) ? kShHalfLdsBytes : 0;
    const bool want_j = D3GA_PRE_DCOL && shs && prm->M > 0 && (3 * prm->M) % 4 == 0 && 3 * prm->M <= 48 && !prm->forward_only;
    if (want_j)
        hipLaunchKernelGGL(recolor_kernel<true>, dim3((prm->P + kBlock - 1) / kBlock), dim3(kBlock), lds, s, *prm, means3D, shs,
                           colors_precomp, campos, src, dst);
    else
        hipLaunchKernelGGL(recolor_kernel<false>, dim3((prm->P + kBlock - 1) / kBlock), dim3(kBlock), lds, s, *prm, means3D, shs,
                           colors_precomp, campos, src, dst);
    return check_launch(s, prm->debug & 0xff);
}

extern "C" int d3ga_raster_preprocess_bwd(const d3ga_raster_params *prm, const float *means3D, const float *shs,
                                          const float *scales, const float *rotations, const float *cov3D_precomp,
                                          const float *viewmatrix, const float *projmatrix, const float *campos,
                                          const void *geom, const float *acc, float *dL_dmeans3D, float *dL_dmeans2D,
                                          float *dL_dopacity, float *dL_dsh, float *dL_dcolors, float *dL_dcov3D,
                                          float *dL_dscales, float *dL_drots, d3ga_stream_t stream) {
    D3GA_TRY(validate(prm));
    if (prm->P == 0) return D3GA_OK;
    if (!means3D || !viewmatrix || !projmatrix || !campos || !geom || !acc || !dL_dmeans3D) return D3GA_E_NULL;
    if (dL_dsh && !shs) return D3GA_E_NULL;
    if (shs && !dL_dsh && !dL_dcolors) return D3GA_E_NULL;        // SH path: full block or factored (P,3) output
    if ((dL_dscales != nullptr) != (dL_drots != nullptr)) return D3GA_E_CONFIG;
    if (dL_dscales && (!scales || !rotations)) return D3GA_E_NULL;
    // the covariance the forward worked with: built from (scales, rotations) -> the geometry record holds it; precomputed -> the
    // forward kept NO copy (ABI 101), the caller's tensor is read again: a NULL here would mean uninitialised records
    if (!cov3D_precomp && !(scales && rotations)) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    const int views = n_views_of(prm);
    const GeomBuf g = carve_geom(const_cast<void *>(geom), (int64_t)prm->P * views);
    const size_t lds = (shs && prm->M > 0 && (3 * prm->M) % 4 == 0) ? kShLdsBytes : 0;
    if (views == 1) {
        hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((prm->P + kBlock - 1) / kBlock), dim3(kBlock), lds, s, *prm, means3D,
                           shs, scales, rotations, viewmatrix, projmatrix, campos, g, acc, dL_dmeans3D, dL_dmeans2D,
                           dL_dopacity, dL_dsh, dL_dcolors, dL_dcov3D, dL_dscales, dL_drots, cov3D_precomp, 0);
        return check_launch(s, prm->debug);
    }
    const int cam_stride = prm->tanfovx > 0.f ? 3 : 5;
    // k views that SHARE their geometry: one pass over the Gaussians walks the views (preprocess_bwd_views_kernel), up to eight per
    // launch -- when the SH gradient is wanted as the (P,M,3) block (not as the factors of the camera-sharded exchange) and, for SH
    // colours, the forward left its direction Jacobian
    {
        const bool staged_sh = shs && prm->M > 0 && (3 * prm->M) % 4 == 0 && 3 * prm->M <= 48;
        const bool looped = shs ? (dL_dsh != nullptr && staged_sh && D3GA_PRE_DCOL && !prm->forward_only) : true;
        const size_t pvl = prm->per_view_geometry ? (size_t)prm->P : 0;
        if (looped) {
            for (int v0 = 0; v0 < views; v0 += kMaxGroup) {
                const int kv = views - v0 < kMaxGroup ? views - v0 : kMaxGroup;
                ViewCamsN vc;
                for (int q = 0; q < kMaxGroup; ++q) {
                    const int v = v0 + (q < kv ? q : 0);
                    vc.vm[q] = viewmatrix + 16 * (size_t)v; vc.pm[q] = projmatrix + 16 * (size_t)v; vc.cp[q] = campos + (size_t)cam_stride * v;
                }
                const size_t o = (size_t)prm->P * v0, og = pvl * v0;
                hipLaunchKernelGGL(preprocess_bwd_views_kernel, dim3((prm->P + kBlock - 1) / kBlock), dim3(kBlock), shs ? kShLdsBytes : 0, s, *prm, kv,
                                   means3D + 3 * og, shs != nullptr, scales ? scales + 3 * og : nullptr, rotations ? rotations + 4 * og : nullptr,
                                   cov3D_precomp ? cov3D_precomp + 6 * og : nullptr, vc, geom_view(g, prm->P, v0), acc + D3GA_ACC_STRIDE * o,
                                   dL_dmeans3D + 3 * og, dL_dmeans2D ? dL_dmeans2D + 3 * o : nullptr, dL_dopacity, dL_dsh, dL_dcolors,
                                   dL_dcov3D ? dL_dcov3D + 6 * og : nullptr, dL_dscales ? dL_dscales + 3 * og : nullptr,
                                   dL_drots ? dL_drots + 4 * og : nullptr, v0 > 0, pvl);
            }
            return check_launch(s, prm->debug);
        }
    }
    // otherwise: one launch per view on ITS records; view 0 writes the gradients of the view-independent
    if (shs && !dL_dcolors) return D3GA_E_NULL;
    const size_t pv = prm->per_view_geometry ? (size_t)prm->P : 0;      // a batch of frames: every view has its own geometry and geometry gradients
    if (prm->factor_rows != 0 && prm->factor_rows < prm->P) return D3GA_E_SIZE;
    const size_t fr = prm->factor_rows > 0 ? (size_t)prm->factor_rows : (size_t)prm->P;      // rows between the views' SH factors
    for (int v = 0; v < views; ++v) {
        const size_t o = (size_t)prm->P * v, og = pv * v;
        hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((prm->P + kBlock - 1) / kBlock), dim3(kBlock), lds, s, *prm, means3D + 3 * og,
                           shs, scales ? scales + 3 * og : nullptr, rotations ? rotations + 4 * og : nullptr, viewmatrix + 16 * (size_t)v,
                           projmatrix + 16 * (size_t)v, campos + (size_t)cam_stride * v, geom_view(g, prm->P, v), acc + D3GA_ACC_STRIDE * o,
                           dL_dmeans3D + 3 * og, dL_dmeans2D ? dL_dmeans2D + 3 * o : nullptr, dL_dopacity, (float *)nullptr,
                           shs ? dL_dcolors + 3 * fr * v : dL_dcolors, dL_dcov3D ? dL_dcov3D + 6 * og : nullptr,
                           dL_dscales ? dL_dscales + 3 * og : nullptr, dL_drots ? dL_drots + 4 * og : nullptr,
                           cov3D_precomp ? cov3D_precomp + 6 * og : nullptr, v > 0 ? (pv ? 1 : 3) : 0);
    }
    D3GA_TRY(check_launch(s, prm->debug));
    if (shs && dL_dsh) {
        const size_t lds2 = ((3 * prm->M) % 4 == 0) ? kShLdsBytes : 0;
        hipLaunchKernelGGL(sh_grad_from_views_kernel, dim3((prm->P + kBlock - 1) / kBlock), dim3(kBlock), lds2, s, prm->P, prm->M, prm->sh_degree,
                           views, means3D, (int64_t)(3 * pv), dL_dcolors, (int64_t)(3 * fr), campos, (int64_t)cam_stride, 1.0f, dL_dsh);
        return check_launch(s, prm->debug);
    }
    return D3GA_OK;
}

extern "C" int d3ga_sh_grad_from_views(int32_t P, int32_t M, int32_t sh_degree, int32_t n_views, const float *means3D,
                                       const float *g_views, int64_t g_stride, const float *campos_views,
                                       int64_t campos_stride, float scale, float *dL_dsh, d3ga_stream_t stream) {
    if (P < 0 || M < 1 || M > 16 || n_views < 0) return D3GA_E_SIZE;
    if (sh_degree < 0 || sh_degree > 3 || (sh_degree + 1) * (sh_degree + 1) > M) return D3GA_E_CONFIG;
    if (P == 0) return D3GA_OK;
    if (!means3D || !dL_dsh || (n_views > 0 && (!g_views || !campos_views))) return D3GA_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = ((3 * M) % 4 == 0) ? kShLdsBytes : 0;
    hipLaunchKernelGGL(sh_grad_from_views_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), lds, s, P, M, sh_degree,
                       n_views, means3D, (int64_t)0, g_views, g_stride, campos_views, campos_stride, scale, dL_dsh);
    return check_launch(s, 0);
}

extern "C" int d3ga_raster_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, uint8_t *visible,
                                        d3ga_stream_t stream) {
    if (P < 0) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!means3D || !viewmatrix || !visible) return D3GA_E_NULL;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, P,
                       means3D, viewmatrix, visible);
    return check_launch((hipStream_t)stream, 0);
}

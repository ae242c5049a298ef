// mlp.hip -- the dense layer of the reference's field networks (models/mlp.py:39-232:
// DeformationField / CanonicalField / ColorField are all  z -> [Linear(128) + leaky_relu(0.1)] x (1 + n_layers) -> Linear).
//
// One kernel template does every GEMM of the forward and of the input-gradient chain of the backward:
//     Y[r][n] = act_out( sum_k X[r][k] * W[k][n] + bias[n] ) [ (.) lrelu'(M[r][n]) ]
// (forward: X = activations, bias + leaky_relu epilogue, and one SIGN BIT per output element on the side; backward:
// X = dPre of the layer, W its transposed weights, and the epilogue multiplies by the leaky_relu derivative of the
// layer below, read from those bits -- 4 bytes per row and column block instead of 128 -- so that what is stored is that
// layer's dPre: the operand of its weight gradient and of the next input-gradient GEMM; no masked copy is ever made)
// with f32-EQUIVALENT arithmetic on the bf16 matrix cores: every f32 operand is split exactly into three bf16 pieces
// (x = x0 + x1 + x2: round to nearest, subtract, repeat -- the remainders are exact, 3 x 8 significand bits cover f32's
// 24), and the six products x_i w_j with i + j <= 2 are accumulated in f32 by v_mfma_f32_32x32x16_bf16.  The dropped
// products are below 2^-24 |x||w|, i.e. below the rounding of an f32 fmaf chain: measured max error against f64 on a
// 128 -> 128 layer 1.48e-6 (fmaf chain: 1.54e-6).  Six 32-cycle instructions per 16 k instead of eight 64-cycle
// v_mfma_f32_32x32x2_f32: 2.7x less matrix time, which turns the layer from matrix-bound (196 us at 500k rows) into
// HBM-bound.
//
// Work decomposition: a persistent workgroup (16 wavefronts) owns 512 rows per step, a wavefront 32 rows x all output
// columns (NB blocks of 32).  K is walked in half-chunks of 16 (one MFMA k-step): the wavefront moves 32 rows x 16 k with
// loads that cover 16 rows x 64 bytes each, splits them, and stores the three bf16 planes into a private LDS buffer in
// operand order (lane l of the MFMA reads the 8 k of row l & 31, k-half l >> 5, with one 16-byte read per plane).  Two
// register slots (even / odd half-chunks) are re-issued as soon as they are committed, so two half-chunks are always in
// flight.  The weight planes (prepared once per weight version by d3ga_mlp_pack_weights) sit in LDS for the whole
// persistent workgroup in operand order [plane][k-step][k-half][column block][column & 31][8 k].
// Weight gradients (dW = dPre^T X, a reduction over all rows) are wgrad_kernel below.
#include "d3ga_internal.h"
#include <utility>

namespace d3ga {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int kMlpThreads = 1024;    // 16 wavefronts share one set of weight planes in LDS (4 wavefronts per SIMD)
constexpr int kMlpRows = 512;        // rows per workgroup step (32 per wavefront)
constexpr int kMlpMaxK = 128;        // K <= 128
constexpr int kMlpAHalf = 640;       // bytes: 32 rows x 16 B of one k-half (+128: the two halves start 32 banks apart)
constexpr int kMlpAPlane = 2 * kMlpAHalf;
constexpr int kMlpABytes = 3 * kMlpAPlane;          // a wavefront's operand buffer: 3 planes x 2 k-halves x 32 rows x 8 bf16
__host__ __device__ constexpr int mlp_panel_units(int K, int n_out) { return 3 * ((K + 15) / 16) * 2 * ((n_out + 31) / 32) * 32; }

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <class F>
__device__ __forceinline__ void static_for_16(F &&f) { static_for_impl(f, std::make_integer_sequence<int, 16>{}); }
template <class F>
__device__ __forceinline__ void static_for_4(F &&f) { static_for_impl(f, std::make_integer_sequence<int, 4>{}); }

// two f32 -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32); element 0 in the low half
__device__ __forceinline__ uint32_t bf16_pack(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
// x = p0 + p1 + p2 exactly (each piece a bf16; the subtractions are exact)
// The residuals come from v_dot2c_f32_bf16 (acc += a.lo b.lo + a.hi b.hi) with b = (-1, 0) / (0, -1): x - float(piece) in ONE
// instruction instead of unpack + subtract -- 7 VALU per pair of values instead of 11.  On this hardware an MFMA and a VALU
// instruction of different wavefronts do not overlap on a SIMD (tools/micro/mfma_valu_overlap.hip: 72 + 97 -> 162 us), so every
// VALU instruction of these kernels costs matrix time.  Exact: the product with -1 is exact, the other one is a true zero, and
// x - piece is representable (the piece is x rounded to 8 significant bits).
#ifndef D3GA_SPLIT_DOT2
#define D3GA_SPLIT_DOT2 1
#endif
__device__ __forceinline__ void bf16_split2(float x, float y, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
#if D3GA_SPLIT_DOT2
    // (-1, 0) and (0, -1) as packed bf16 in SGPRs the compiler cannot see through: written as constants it encodes (-1, 0) as
    // the INLINE constant -1.0, which this instruction does not read as a bf16 pair (every result was garbage)
    uint32_t k_lo = 0x0000BF80u, k_hi = 0xBF800000u;
    asm volatile("" : "+s"(k_lo), "+s"(k_hi));
    const bf16x2_t m_lo = __builtin_bit_cast(bf16x2_t, k_lo), m_hi = __builtin_bit_cast(bf16x2_t, k_hi);
    p0 = bf16_pack(x, y);
    const float r0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, p0), m_lo, x, false);
    const float r1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, p0), m_hi, y, false);
    p1 = bf16_pack(r0, r1);
    const float s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, p1), m_lo, r0, false);
    const float s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, p1), m_hi, r1, false);
    p2 = bf16_pack(s0, s1);
#else
    p0 = bf16_pack(x, y);
    const float r0 = x - bf16_lo(p0), r1 = y - bf16_hi(p0);
    p1 = bf16_pack(r0, r1);
    p2 = bf16_pack(r0 - bf16_lo(p1), r1 - bf16_hi(p1));
#endif
}

// Weight planes for linear_kernel: unit (plane, kk, half, nb, n32) = the 8 bf16 pieces w[k = 16 kk + 8 half + j][n = 32 nb + n32],
// j = 0..7, of weight(k, n) = W[k * ld_k + n * ld_n]; zero past K / n_out.  One thread per (kk, half, nb, n32).
__global__ __launch_bounds__(kBlock) void pack_weights_kernel(int K, int n_out, const float *__restrict__ W, int64_t ld_k,
                                                              int64_t ld_n, uint4 *__restrict__ panel) {
    const int KK = (K + 15) / 16, NB = (n_out + 31) / 32;
    const int u = blockIdx.x * kBlock + threadIdx.x;
    if (u >= KK * 2 * NB * 32) return;
    const int n32 = u & 31, nb = (u >> 5) % NB, kh = (u >> 5) / NB;          // kh = 2 kk + half
    const int n = 32 * nb + n32, k0 = 8 * kh;
    uint32_t q[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ka = k0 + 2 * j, kb = ka + 1;
        const float wa = (ka < K && n < n_out) ? W[ka * ld_k + n * ld_n] : 0.f;
        const float wb = (kb < K && n < n_out) ? W[kb * ld_k + n * ld_n] : 0.f;
        bf16_split2(wa, wb, q[0][j], q[1][j], q[2][j]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) panel[pl * (KK * 2 * NB * 32) + u] = make_uint4(q[pl][0], q[pl][1], q[pl][2], q[pl][3]);
}

// One half-chunk (16 columns) of a wavefront's 32 rows, global -> registers: lane l covers the 16 bytes (l & 3) of the
// 64-byte piece of row 16*j + (l >> 2), j = 0..1.  slot_issue only ISSUES the loads (raw values, addresses clamped to stay
// in range): nothing may consume them here -- a select or the mask multiply right after the load makes the compiler wait
// for HBM on the spot and the prefetch degenerates into a blocking load.  slot_commit, called two half-chunks later,
// zeroes what lies past the matrix, splits the values into the three bf16 planes and stores them into the wavefront's
// LDS buffer.
template <bool VEC>
__device__ __forceinline__ void slot_issue(float4 (&v)[2], int P, int K, int row0, int k0, int lane,
                                           const float *__restrict__ X) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = row0 + 16 * j + (lane >> 2), k = k0 + 4 * (lane & 3);
        const int rc = r < P ? r : P - 1;
        if constexpr (VEC) {                                           // K % 4 == 0: rows are 16-byte aligned
            const uint32_t o = (uint32_t)rc * (uint32_t)K + (uint32_t)(k < K ? k : 0);
            v[j] = *reinterpret_cast<const float4 *>(X + o);
        } else {
            float e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = X[(uint32_t)rc * (uint32_t)K + (uint32_t)(k + c < K ? k + c : 0)];
            v[j] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
}

template <bool RAGGED>
__device__ __forceinline__ void slot_commit(char *s_a, const float4 (&v)[2], int P, int K, int row0, int k0, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = row0 + 16 * j + (lane >> 2), k = k0 + 4 * (lane & 3);
        float4 x = v[j];
        const bool ok_r = RAGGED ? r < P : true;                       // !RAGGED: the launcher guarantees P % 32 == 0
        if (RAGGED || (K & 15)) {                                      // (uniform) nothing to zero when K fills its half-chunks
            if (!(ok_r && k < K)) x.x = 0.f;
            if (!(ok_r && k + 1 < K)) x.y = 0.f;
            if (!(ok_r && k + 2 < K)) x.z = 0.f;
            if (!(ok_r && k + 3 < K)) x.w = 0.f;
        }
        uint32_t a0, a1, a2, b0, b1, b2;
        bf16_split2(x.x, x.y, a0, a1, a2);
        bf16_split2(x.z, x.w, b0, b1, b2);
        // operand order: [plane][k-half = (l & 3) >> 1][row][8 bf16]; this lane's 4 k are the (l & 1)-th 8 bytes of the unit
        char *d = s_a + ((lane & 3) >> 1) * kMlpAHalf + (16 * j + (lane >> 2)) * 16 + (lane & 1) * 8;
        *reinterpret_cast<uint2 *>(d) = make_uint2(a0, b0);
        *reinterpret_cast<uint2 *>(d + kMlpAPlane) = make_uint2(a1, b1);
        *reinterpret_cast<uint2 *>(d + 2 * kMlpAPlane) = make_uint2(a2, b2);
    }
}

template <int NB, bool VEC, bool EMASK, bool RAGGED>
__global__ __launch_bounds__(kMlpThreads) void linear_kernel(int P, int K, int n_store, const float *__restrict__ X,
                                                             const uint4 *__restrict__ Wp, const float *__restrict__ bias,
                                                             float out_slope, uint32_t *__restrict__ sign_out,
                                                             const uint32_t *__restrict__ mask_bits, float emask_slope,
                                                             float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) char smem_mlp[];    // weight planes | 16 x operand buffer
    const int KK = (K + 15) / 16;                                      // MFMA k-steps (K padded to 16)
    const int plane_units = KK * 2 * NB * 32;
    uint4 *s_w = reinterpret_cast<uint4 *>(smem_mlp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char *s_a = smem_mlp + 3 * plane_units * 16 + wave * kMlpABytes;
    const int half = lane >> 5, l32 = lane & 31;
    for (int u = tid; u < 3 * plane_units; u += kMlpThreads) s_w[u] = Wp[u];       // contiguous copy
    __syncthreads();
    const int ntiles = (P + kMlpRows - 1) / kMlpRows;
    // this lane's bias values, once (a global load inside the epilogue would force a vmcnt(0) in front of the stores)
    float bias_r[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias_r[nb] = (bias && l32 + 32 * nb < n_store) ? bias[l32 + 32 * nb] : 0.f;
    float4 va[2], vb[2];
    {   // the two slots in issue order (the waits below count on it)
        const int row0 = blockIdx.x * kMlpRows + wave * 32;
        slot_issue<VEC>(va, P, K, row0, 0, lane, X);
        __builtin_amdgcn_sched_barrier(0);
        slot_issue<VEC>(vb, P, K, row0, 16, lane, X);
        __builtin_amdgcn_sched_barrier(0);
    }
    const char *a_rd = s_a + half * kMlpAHalf + l32 * 16;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * kMlpRows + wave * 32;
        const int nrow0 = row0 + (int)gridDim.x * kMlpRows;            // past the last tile: slot_issue clamps the rows
        if (!RAGGED && row0 >= P) continue;                            // wave-uniform: the last tile may be partly empty
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        auto mma = [&](int kk) __attribute__((always_inline)) {
            bf16x8_t a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                a[pl] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(a_rd + pl * kMlpAPlane));
            const uint4 *wrd = s_w + ((kk * 2 + half) * NB) * 32 + l32;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                bf16x8_t w[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) w[pl] = __builtin_bit_cast(bf16x8_t, wrd[pl * plane_units + nb * 32]);
                // smallest products first
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], w[0], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], w[1], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[2], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], w[0], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[1], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[0], acc[nb], 0, 0, 0);
            }
        };
        for (int kk = 0; kk < KK; kk += 2) {
            // even half-chunk: commit slot a, re-issue it (two half-chunks ahead, or the next tile's first), multiply
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_wave_barrier();                           // the previous half-chunk's LDS reads are done
            slot_commit<RAGGED>(s_a, va, P, K, row0, 16 * kk, lane);
            {
                const bool same = kk + 2 < KK;
                slot_issue<VEC>(va, P, K, same ? row0 : nrow0, same ? 16 * (kk + 2) : 0, lane, X);
            }
            __builtin_amdgcn_wave_barrier();
            mma(kk);
            // odd half-chunk (absent in the last pair when KK is odd: its slot is still re-issued, never committed)
            __builtin_amdgcn_sched_barrier(0);
            const bool has_b = kk + 1 < KK;
            if (has_b) {
                __builtin_amdgcn_wave_barrier();
                slot_commit<RAGGED>(s_a, vb, P, K, row0, 16 * (kk + 1), lane);
            }
            {
                const bool same = kk + 3 < KK;
                slot_issue<VEC>(vb, P, K, same ? row0 : nrow0, same ? 16 * (kk + 3) : 16, lane, X);
            }
            if (has_b) {
                __builtin_amdgcn_wave_barrier();
                mma(kk + 1);
            }
        }
        // epilogue: C/D layout of the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5);
        // every store instruction covers two full 128-byte row segments.
        // Sign bits: the ballot of (y > 0) for register r IS the pair of 32-bit words (columns of block nb) of rows
        // dr(r) and dr(r) + 4; lane r (and 32 + r) collects them (v_writelane) and the sixteen lanes r < 16 of each half
        // store NB words each.  EMASK reads the same words back (one load per lane), broadcasts them with v_readlane and
        // uses the 64-bit pair directly as the lane mask of the select.
        constexpr bool full = !RAGGED;                                 // launcher: P % 32 == 0 and n_store == 32*NB
        const uint32_t ybase0 = (uint32_t)(row0 + 4 * half) * (uint32_t)n_store + (uint32_t)l32;
        const int my_row = row0 + 4 * half + (l32 & 3) + 8 * ((l32 >> 2) & 3);        // the row lane l32 < 16 keeps words of
        const bool word_lane = l32 < 16 && (full || my_row < P);
        uint32_t mw[NB];
        if constexpr (EMASK) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) mw[nb] = word_lane ? mask_bits[(uint32_t)my_row * (uint32_t)NB + nb] : 0u;
        }
        uint32_t sw[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = l32 + 32 * nb;
            const float b = bias_r[nb];
            sw[nb] = 0u;
            static_for_16([&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;                 // a constant expression: v_writelane's lane select
                const int dr = (r & 3) + 8 * (r >> 2);
                float y = acc[nb][r] + b;
                y = y > 0.f ? y : out_slope * y;
                if constexpr (EMASK) {
                    const uint64_t keep = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mw[nb], 32 + r) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)mw[nb], r);
                    const float ys = y * emask_slope;
                    // (s_nop: gfx950 wants two wait states between a VALU write of an SGPR -- the v_readlanes -- and a VALU
                    // read of it as a lane mask; inside inline asm nobody inserts them)
                    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(y) : "v"(ys), "v"(y), "s"(keep));
                }
                if (sign_out) {                                        // (uniform)
                    const uint64_t pos = __ballot(y > 0.f);
                    // (the s_nop: v_writelane reading an SGPR the VALU has just written -- the ballot -- needs wait
                    // states the assembler does not insert inside inline asm; without them lanes got stale words)
                    asm("s_nop 3\n\tv_writelane_b32 %0, %1, %2\n\tv_writelane_b32 %0, %3, %4"
                        : "+v"(sw[nb]) : "s"((uint32_t)pos), "n"(r), "s"((uint32_t)(pos >> 32)), "n"(32 + r));
                }
                if (full || (row0 + 4 * half + dr < P && n < n_store)) Y[ybase0 + 32 * nb + (uint32_t)dr * (uint32_t)n_store] = y;
            });
        }
        if (sign_out && word_lane) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) sign_out[(uint32_t)my_row * (uint32_t)NB + nb] = sw[nb];
        }
    }
}

// Weight gradient dW (N,K) += dPre^T (N x rows) . X (rows x K): the contraction runs over the ROWS, so the MFMA operands
// are COLUMNS of the row-major arrays -- lane l needs 8 consecutive rows of one column as one bf16x8.  A workgroup owns
// a contiguous row range and walks it in chunks of 64 rows.  Wavefronts 0-7 fetch dPre, 8-15 fetch X: wavefront w owns
// the 8 rows of row group w & 7, lane l the columns 2l, 2l+1 (every load instruction is one full 512-byte row), so after
// the loads a lane HOLDS its two columns' 8 rows: it splits them (same exact 3-way bf16 split as above) and writes
// operand-ready 16-byte units [plane][row group][column] -- the transpose costs nothing.  Then one wavefront per 32x32
// block of dW (NBn x NBk of the 16) multiplies: 4 k-steps x 6 products per chunk.  The next chunk's loads are in flight
// meanwhile.  Partial results meet in global memory through float atomics (dW zeroed by the launcher); the dPre loaders
// also accumulate the bias gradient from their f32 values.  (History: per-wavefront global fetches 311 us per 128x128
// layer at 500k rows; LDS-staged f32 MFMA 202 us.)
constexpr int kWgRows = 64;
constexpr int kWgThreads = 1024;     // 16 wavefronts: all of them move data, NBn x NBk of them own a block of dW
constexpr int kWgPlaneUnits = 8 * 128;              // 16-byte units per plane of one operand: [row group][column]
constexpr int kWgOperandBytes = 3 * kWgPlaneUnits * 16;

// A loader job: this lane fetches 8 consecutive rows (row group g of the chunk) of the columns c, c+1 (nc = 2) or c
// (nc = 1) of one operand: global -> registers (issue; raw values, clamped addresses) -> LDS units (commit).
struct WgJob {
    const float *src;    // operand (row-major, `width` columns); nullptr: no job
    uint4 *s_op;         // its planes in LDS
    int width, g, c, nc;
    bool is_a;           // dPre: its f32 values also feed the bias gradient
};
template <bool VEC2>
__device__ __forceinline__ void wg_issue(float2 (&v)[8], const WgJob &job, int r0, int r_end) {
    const int c = job.c, w = job.width;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rr = min(r0 + 8 * job.g + i, r_end - 1);
        const uint32_t o = (uint32_t)rr * (uint32_t)w;
        if (VEC2 && job.nc == 2) v[i] = *reinterpret_cast<const float2 *>(job.src + o + (c < w ? c : 0));
        else v[i] = make_float2(job.src[o + (c < w ? c : 0)], job.src[o + (c + 1 < w ? c + 1 : 0)]);
    }
}
__device__ __forceinline__ void wg_commit(const float2 (&v)[8], const WgJob &job, int r0, int r_end, float (&colsum)[2]) {
    const int c = job.c;
    float x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool ok = r0 + 8 * job.g + i < r_end;
        x[i] = (ok && c < job.width) ? v[i].x : 0.f;
        y[i] = (ok && job.nc == 2 && c + 1 < job.width) ? v[i].y : 0.f;
        colsum[0] += x[i];
        colsum[1] += y[i];
    }
    uint32_t px[3][4], py[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bf16_split2(x[2 * q], x[2 * q + 1], px[0][q], px[1][q], px[2][q]);
        bf16_split2(y[2 * q], y[2 * q + 1], py[0][q], py[1][q], py[2][q]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        uint4 *d = job.s_op + pl * kWgPlaneUnits + job.g * 128 + c;
        d[0] = make_uint4(px[pl][0], px[pl][1], px[pl][2], px[pl][3]);
        if (job.nc == 2) d[1] = make_uint4(py[pl][0], py[pl][1], py[pl][2], py[pl][3]);
    }
}

// the same commit for the wavefront-specialised kernel below (4 row groups per plane instead of 8)
__device__ __forceinline__ void wg_commit_ws(const float2 (&v)[8], const WgJob &job, int r0, int r_end, float (&colsum)[2]) {
    const int c = job.c;
    float x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool ok = r0 + 8 * job.g + i < r_end;
        x[i] = (ok && c < job.width) ? v[i].x : 0.f;
        y[i] = (ok && c + 1 < job.width) ? v[i].y : 0.f;
        colsum[0] += x[i];
        colsum[1] += y[i];
    }
    uint32_t px[3][4], py[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bf16_split2(x[2 * q], x[2 * q + 1], px[0][q], px[1][q], px[2][q]);
        bf16_split2(y[2 * q], y[2 * q + 1], py[0][q], py[1][q], py[2][q]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        uint4 *d = job.s_op + pl * (4 * 128) + job.g * 128 + c;
        d[0] = make_uint4(px[pl][0], px[pl][1], px[pl][2], px[pl][3]);
        d[1] = make_uint4(py[pl][0], py[pl][1], py[pl][2], py[pl][3]);
    }
}

// Loader assignment (all wave-uniform).  Both operands wide: wavefronts 0-7 take dPre, 8-15 take X, row group w & 7, lane l
// the columns 2l, 2l+1.  One operand narrow (<= 16 columns: the first and last layers of the fields): the WIDE one is
// spread over all 16 wavefronts (row group w & 7, column 64 (w >> 3) + l) so that as many bytes stay in flight as in
// the square case, and wavefront 0 fetches the whole narrow chunk as a second job (row group l >> 3, columns 2 (l & 7)).
template <bool VECA, bool VECB, bool MIXED>
__global__ __launch_bounds__(kWgThreads) void wgrad_kernel(int P, int N, int K, int NBk, int n_blocks, int rows_per_block,
                                                           const float *__restrict__ dpre, const float *__restrict__ X,
                                                           float *__restrict__ dW, float *__restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) char smem_wg[];     // dPre planes | X planes
    uint4 *s_a = reinterpret_cast<uint4 *>(smem_wg), *s_b = reinterpret_cast<uint4 *>(smem_wg + kWgOperandBytes);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool worker = wave < n_blocks;                               // wave-uniform: owns a 32x32 block of dW
    const int nb = wave / NBk, kb = wave - nb * NBk;
    const int half = lane >> 5, l32 = lane & 31;
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(P, r_begin + rows_per_block);
    if (r_begin >= r_end) return;
    const bool narrow_a = MIXED && N <= 16, narrow_b = MIXED && K <= 16;      // !MIXED: the launcher saw two wide operands
    WgJob j1 = {nullptr, nullptr, 0, 0, 0, 0, false}, j2 = j1;
    const WgJob wide2_a = {dpre, s_a, N, wave & 7, 2 * lane, 2, true}, wide2_b = {X, s_b, K, wave & 7, 2 * lane, 2, false};
    const WgJob wide1_a = {dpre, s_a, N, wave & 7, 64 * (wave >> 3) + lane, 1, true};
    const WgJob wide1_b = {X, s_b, K, wave & 7, 64 * (wave >> 3) + lane, 1, false};
    const WgJob small_a = {dpre, s_a, N, lane >> 3, 2 * (lane & 7), 2, true}, small_b = {X, s_b, K, lane >> 3, 2 * (lane & 7), 2, false};
    if (!narrow_a && !narrow_b) j1 = wave < 8 ? wide2_a : wide2_b;
    else if (narrow_a && !narrow_b) { j1 = wide1_b; if (wave == 0) j2 = small_a; }
    else if (!narrow_a && narrow_b) { j1 = wide1_a; if (wave == 0) j2 = small_b; }
    else { if (wave == 0) j1 = small_a; if (wave == 8) j1 = small_b; }
    float2 v1[8], v2[8];
    float cs1[2] = {0.f, 0.f}, cs2[2] = {0.f, 0.f};
    auto issue = [&](int r0) __attribute__((always_inline)) {
        if (j1.src) { if (j1.is_a) wg_issue<VECA>(v1, j1, r0, r_end); else wg_issue<VECB>(v1, j1, r0, r_end); }
        if constexpr (MIXED) { if (j2.src) wg_issue<false>(v2, j2, r0, r_end); }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (narrow_a || narrow_b) {    // columns no job writes must read as zero (a narrow job covers 16 columns, MFMA blocks are 32)
        for (int u = tid; u < 2 * 3 * kWgPlaneUnits; u += kWgThreads) reinterpret_cast<uint4 *>(smem_wg)[u] = make_uint4(0, 0, 0, 0);
    }
    issue(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += kWgRows) {
        __syncthreads();                                               // previous chunk's LDS reads are done
        if (j1.src) wg_commit(v1, j1, r0, r_end, cs1);
        if constexpr (MIXED) { if (j2.src) wg_commit(v2, j2, r0, r_end, cs2); }
        __syncthreads();
        issue(min(r0 + kWgRows, r_end - 1));                           // prefetch while the MFMAs below run (clamped rows)
        if (worker) {
            const uint4 *pa = s_a + half * 128 + 32 * nb + l32, *pb = s_b + half * 128 + 32 * kb + l32;
#pragma unroll
            for (int ks = 0; ks < kWgRows / 16; ++ks) {
                bf16x8_t a[3], b[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a[pl] = __builtin_bit_cast(bf16x8_t, pa[pl * kWgPlaneUnits + 2 * ks * 128]);
                    b[pl] = __builtin_bit_cast(bf16x8_t, pb[pl * kWgPlaneUnits + 2 * ks * 128]);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
            }
        }
    }
    // bias gradient: every dPre loader lane holds the sums of its columns over its row groups; meet in LDS (float
    // atomics on 128 words), then one global atomic per column
    if (db) {
        __syncthreads();
        float *s_sum = reinterpret_cast<float *>(smem_wg);
        if (tid < 128) s_sum[tid] = 0.f;
        __syncthreads();
        if (j1.src && j1.is_a) { if (j1.c < N) atomicAdd(s_sum + j1.c, cs1[0]); if (j1.nc == 2 && j1.c + 1 < N) atomicAdd(s_sum + j1.c + 1, cs1[1]); }
        if (MIXED && j2.src && j2.is_a) { if (j2.c < N) atomicAdd(s_sum + j2.c, cs2[0]); if (j2.c + 1 < N) atomicAdd(s_sum + j2.c + 1, cs2[1]); }
        __syncthreads();
        if (tid < N) atomicAdd(db + tid, s_sum[tid]);
    }
    if (!worker) return;
    // C/D layout: col = lane & 31 -> k, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> n
    const int k = 32 * kb + l32;
    if (k < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (nn < N) atomicAdd(dW + (uint32_t)nn * (uint32_t)K + (uint32_t)k, acc[r]);
        }
    }
}

// Wavefront-specialised variant for two wide operands (the square layers): wavefronts 0-7 only MOVE data (global ->
// registers -> split -> LDS), wavefronts 8-15 only MULTIPLY.  Chunks of 32 rows, two LDS buffers: while the loaders
// commit chunk c into buffer c & 1 the multipliers work on chunk c - 1 in the other buffer, one barrier per chunk.  (In
// wgrad_kernel every wavefront does both and the two phases are separated by barriers: measured, the 3100-cycle multiply
// phase of each chunk simply ADDS to the load / commit skeleton -- 161 vs 118 us per 128x128 layer at 500k rows.)
// Loader w: operand w >> 2 (dPre | X), row group w & 3, lane l the columns 2l, 2l+1, two chunks of loads in flight.
// Multiplier m: column block nb = m / NBk2 of dPre against the k blocks 2 (m % NBk2), +1 of X (the A operand is read once
// for both).
constexpr int kWsRows = 32;
constexpr int kWsPlaneUnits = 4 * 128;                              // [row group][column]
constexpr int kWsOperandUnits = 3 * kWsPlaneUnits;
constexpr int kWsBufferUnits = 2 * kWsOperandUnits;                 // dPre planes | X planes
template <bool VECA, bool VECB>
__global__ __launch_bounds__(kWgThreads) void wgrad_ws_kernel(int P, int N, int K, int NBk, int NBn, int rows_per_block,
                                                              const float *__restrict__ dpre, const float *__restrict__ X,
                                                              float *__restrict__ dW, float *__restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) char smem_ws[];     // two buffers x (dPre planes | X planes)
    uint4 *s_buf = reinterpret_cast<uint4 *>(smem_ws);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(P, r_begin + rows_per_block);
    if (r_begin >= r_end) return;
    const int n_chunks = (r_end - r_begin + kWsRows - 1) / kWsRows;
    const bool loader = wave < 8;
    // loader state
    const bool is_a = wave < 4;
    WgJob job = {is_a ? dpre : X, nullptr, is_a ? N : K, wave & 3, 2 * lane, 2, is_a};
    float2 v0[8], v1[8];
    float cs[2] = {0.f, 0.f};
    // multiplier state
    const int m = wave - 8, NBk2 = (NBk + 1) / 2;
    const bool worker = !loader && m < NBn * NBk2;
    const int nb = worker ? m / NBk2 : 0, kb0 = worker ? 2 * (m - nb * NBk2) : 0;
    const bool two = kb0 + 1 < NBk;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    auto fetch = [&](float2 (&v)[8], int c) __attribute__((always_inline)) {
        const int r0 = min(r_begin + c * kWsRows, r_end - 1);          // past the range: clamped rows, never committed
        if (is_a) wg_issue<VECA>(v, job, r0, r_end); else wg_issue<VECB>(v, job, r0, r_end);
    };
    auto commit = [&](const float2 (&v)[8], int c) __attribute__((always_inline)) {
        WgJob j = job;
        j.s_op = s_buf + (c & 1) * kWsBufferUnits + (is_a ? 0 : kWsOperandUnits);
        wg_commit_ws(v, j, r_begin + c * kWsRows, r_end, cs);
    };
    auto multiply = [&](int c) __attribute__((always_inline)) {
        const uint4 *pa = s_buf + (c & 1) * kWsBufferUnits + half * 128 + 32 * nb + l32;
        const uint4 *pb = s_buf + (c & 1) * kWsBufferUnits + kWsOperandUnits + half * 128 + 32 * kb0 + l32;
#pragma unroll
        for (int ks = 0; ks < kWsRows / 16; ++ks) {
            bf16x8_t a[3], b[3], d[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[pl] = __builtin_bit_cast(bf16x8_t, pa[pl * kWsPlaneUnits + 2 * ks * 128]);
                b[pl] = __builtin_bit_cast(bf16x8_t, pb[pl * kWsPlaneUnits + 2 * ks * 128]);
                d[pl] = __builtin_bit_cast(bf16x8_t, pb[pl * kWsPlaneUnits + 2 * ks * 128 + (two ? 32 : 0)]);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], d[0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], d[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], d[2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], d[0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], d[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], d[0], acc1, 0, 0, 0);
        }
    };
    if (loader) {
        fetch(v0, 0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(v1, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // iteration c: loaders commit chunk c (even chunks live in v0, odd in v1) and re-issue that register set for chunk
    // c + 2; multipliers work on chunk c - 1.  One barrier per iteration; one more round drains the pipeline.
    for (int c = 0; c <= n_chunks; c += 2) {
        if (loader) {
            if (c < n_chunks) { commit(v0, c); fetch(v0, c + 2); }
        } else if (worker && c >= 1) multiply(c - 1);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 <= n_chunks) {                                       // (uniform)
            if (loader) {
                if (c + 1 < n_chunks) { commit(v1, c + 1); fetch(v1, c + 3); }
            } else if (worker) multiply(c);
            __syncthreads();
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // bias gradient: the dPre loaders hold the sums of their columns over their row groups
    if (db) {
        __syncthreads();
        float *s_sum = reinterpret_cast<float *>(smem_ws);
        if (tid < 128) s_sum[tid] = 0.f;
        __syncthreads();
        if (loader && is_a) { if (job.c < N) atomicAdd(s_sum + job.c, cs[0]); if (job.c + 1 < N) atomicAdd(s_sum + job.c + 1, cs[1]); }
        __syncthreads();
        if (tid < N) atomicAdd(db + tid, s_sum[tid]);
    }
    if (!worker) return;
    // C/D layout: col = lane & 31 -> k, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> n
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        if (blk == 1 && !two) break;
        const int k = 32 * (kb0 + blk) + l32;
        if (k < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (nn < N) atomicAdd(dW + (uint32_t)nn * (uint32_t)K + (uint32_t)k, blk ? acc1[r] : acc0[r]);
            }
        }
    }
}



// ---------------------------------------------------------------------------------------------------------------------
// Round 3: ONE kernel per field network (forward) -- the activations never leave the registers.
//
// The per-layer kernel above writes every 128-wide activation to HBM and the next launch reads it back (512 MB per
// hidden layer at 500k rows: the forward is HBM-bound).  Here a wavefront keeps ITS 32 rows through all the layers:
//   * the product is taken TRANSPOSED, Y^T = W X^T: the first MFMA operand is a 32 x 16 tile of the weights (rows =
//     output features), the second one the activations (columns = the wavefront's 32 batch rows).  The result tile then
//     has the batch row on the lane (n = lane & 31) and 16 features in its registers, m = 32 t + 8 (r >> 2) + 4 (lane >> 5)
//     + (r & 3) -- and that is, up to a fixed permutation of the features, exactly the layout the SECOND operand of the next
//     layer wants (lane = batch row, 8 consecutive k per k-step and lane half).  The permutation is baked into the order
//     in which the next layer's weights are packed (chain panel: k-slot j of lane half h in k-step s is input feature
//     16 s + 8 (j >> 2) + 4 h + (j & 3)), so register r of output tile t simply IS slot 16 t + r of the next layer's input:
//     no transpose, no LDS round trip, no HBM round trip between the layers;
//   * same exact 3-way bf16 split and the same six products (smallest first) as the per-layer kernel: the activations are
//     split in registers right before they are used, the weights once per weight version by d3ga_mlp_pack_chain;
//   * shapes: every layer but the last is 128 wide (the reference's fields: n_nodes = 128, models/mlp.py:50-69), the
//     first takes K0 <= 128 inputs, the last has NTL = ceil(N / 32) output tiles (a template parameter).  Everything
//     else goes through the per-layer kernels (the host side decides);
//   * a workgroup = 4 wavefronts (one per SIMD) = 128 rows per step; two workgroups per CU run out of phase and cover each
//     other's epilogues and waits.  The weights stream through LDS in chunks of two k-steps (3 planes x 2 x NT KB <= 24 KB),
//     two slots, loaded by LDS-DMA (global_load_lds_dwordx4: no registers) one chunk ahead of the MFMAs; one barrier per
//     chunk;
//   * what the backward needs is still written: every layer's output (the input of the next layer's weight gradient) and
//     one sign bit per activation -- the same arrays, in the same layout, as the per-layer forward produces, so the
//     existing backward (input-gradient GEMMs + wgrad) runs unchanged.  The output tile goes through a per-wavefront LDS
//     tile so that the stores are 128-byte rows, not 32 scattered 16-byte pieces.  What disappears is every READ of an
//     activation in the forward.
struct ChainLayer {
    const uint4 *panel;      // d3ga_mlp_pack_chain: [plane][k-step][tile][lane] units of 8 bf16, then 128 floats of bias
    float *out;              // (P, N) layer output (after the activation)
    uint32_t *sign;          // (P, ceil(N / 32)) sign words or null
    const uint32_t *mask;    // (P, ceil(N / 32)) sign words of ANOTHER chain or null: out (.)= bit ? 1 : mask_slope  (the
                             // backward's input-gradient chain: this layer's output is the pre-activation gradient below it)
    int K, N;
    float slope;             // leaky_relu slope behind this layer (1: none)
    float mask_slope;
};
constexpr int kChainMaxLayers = 8;
struct ChainArgs { ChainLayer layer[kChainMaxLayers]; int L; int abl; };
#ifndef D3GA_CHAIN_WAVES
#define D3GA_CHAIN_WAVES 8
#endif
constexpr int kChainWaves = D3GA_CHAIN_WAVES;
constexpr int kChainThreads = 64 * kChainWaves;    // 4 wavefronts (one per SIMD), 32 rows each
constexpr int kChainRows = 32 * kChainWaves;
// k-steps of a chain panel: padded to an even number (a chunk is always two k-steps; the padding weights are zero)
__host__ __device__ constexpr int chain_ksteps(int K) { return 2 * ((K + 31) / 32); }
__host__ __device__ constexpr int chain_panel_units(int K, int N) { return 3 * chain_ksteps(K) * ((N + 31) / 32) * 64; }

// chain panel of one layer: unit (plane, s, t, lane = 32 h + i) = the 8 bf16 pieces of weight(f_j, o), o = 32 t + i,
// f_j = 16 s + 8 (j >> 2) + 4 h + (j & 3), weight(k, n) = W[k * ld_k + n * ld_n]; zero past K / N.
// The panel ends in room for the layer's bias: 128 floats, zero past N (one 512-byte piece for the kernel's LDS copy), written
// by every d3ga_mlp_chain_fwd call (chain_bias_kernel): a bias is an activation-like input (the fields fold the pose into it),
// not part of the weight version the panel is cached under.
__global__ __launch_bounds__(kBlock) void pack_chain_kernel(int K, int N, const float *__restrict__ W, int64_t ld_k, int64_t ld_n,
                                                            uint4 *__restrict__ panel) {
    const int KS = chain_ksteps(K), NT = (N + 31) / 32;
    const int u = blockIdx.x * kBlock + threadIdx.x;
    if (u >= KS * NT * 64) return;
    const int lane = u & 63, t = (u >> 6) % NT, sk = (u >> 6) / NT;
    const int h = lane >> 5, o = 32 * t + (lane & 31);
    uint32_t q[3][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int ja = 2 * jj, jb = ja + 1;
        const int fa = 16 * sk + 8 * (ja >> 2) + 4 * h + (ja & 3), fb = 16 * sk + 8 * (jb >> 2) + 4 * h + (jb & 3);
        const float wa = (fa < K && o < N) ? W[fa * ld_k + o * ld_n] : 0.f;
        const float wb = (fb < K && o < N) ? W[fb * ld_k + o * ld_n] : 0.f;
        bf16_split2(wa, wb, q[0][jj], q[1][jj], q[2][jj]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) panel[(size_t)pl * (KS * NT * 64) + u] = make_uint4(q[pl][0], q[pl][1], q[pl][2], q[pl][3]);
}

struct ChainBiasArgs { const float *bias[kChainMaxLayers]; float *tail[kChainMaxLayers]; int N[kChainMaxLayers]; };
__global__ __launch_bounds__(128) void chain_bias_kernel(ChainBiasArgs a) {
    const int l = blockIdx.x, u = threadIdx.x;
    a.tail[l][u] = (a.bias[l] && u < a.N[l]) ? a.bias[l][u] : 0.f;
}

#ifndef D3GA_CHAIN_CS
#define D3GA_CHAIN_CS 2
#endif
constexpr int kChainCS = D3GA_CHAIN_CS;                   // k-steps per chunk (2 or 4; layer 0's last chunk may hold 2 of 4)
static_assert(kChainCS == 2 || kChainCS == 4, "chunk = 2 or 4 k-steps");
constexpr int kChainSlotUnits = 3 * kChainCS * 4 * 64;   // one chunk: [plane][k-steps][NT tiles][lane] units, <= 24 / 48 KB
constexpr int kChainSlots = 2;                           // the current chunk and the next one
constexpr int kChainStageLd = 36;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_void;
typedef __attribute__((address_space(3))) float lds_f32;
typedef float f32x4_ev __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_ev __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) Float4U { float v[4]; };      // a 16-byte load that is only 4-byte aligned
// 16 bytes of LDS behind the compiler's back: its wait-count pass puts s_waitcnt vmcnt(0) in front of LDS accesses it relates
// to the LDS-DMA, i.e. every output tile's stores would complete before the next tile starts.  These buffers are ordered
// against the DMA by the workgroup barrier (bias) or are private to the wavefront (the transposition tile).
__device__ __forceinline__ float4 lds_read16_opaque(const float *p) {
    f32x4_ev v;
    const uint32_t a = (uint32_t)(uintptr_t)(lds_f32 *)p;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return make_float4(v[0], v[1], v[2], v[3]);
}
// four of them, one wait
__device__ __forceinline__ void lds_read16x4_opaque(const float *p0, const float *p1, const float *p2, const float *p3, float4 (&o)[4]) {
    f32x4_ev v0, v1, v2, v3;
    const uint32_t a0 = (uint32_t)(uintptr_t)(lds_f32 *)p0, a1 = (uint32_t)(uintptr_t)(lds_f32 *)p1,
                   a2 = (uint32_t)(uintptr_t)(lds_f32 *)p2, a3 = (uint32_t)(uintptr_t)(lds_f32 *)p3;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
    o[0] = make_float4(v0[0], v0[1], v0[2], v0[3]);
    o[1] = make_float4(v1[0], v1[1], v1[2], v1[3]);
    o[2] = make_float4(v2[0], v2[1], v2[2], v2[3]);
    o[3] = make_float4(v3[0], v3[1], v3[2], v3[3]);
}
__device__ __forceinline__ void lds_write16_opaque(float *p, float4 v) {
    const uint32_t a = (uint32_t)(uintptr_t)(lds_f32 *)p;
    const f32x4_ev w = {v.x, v.y, v.z, v.w};
    asm volatile("ds_write_b128 %0, %1" : : "v"(a), "v"(w) : "memory");
}

// One LDS-DMA piece: 64 lanes x 16 bytes from `src` (per lane) to LDS at `dst` (wave-uniform) + lane x 16.  Inline assembly
// on purpose: the compiler's wait-count pass treats the builtin form conservatively (s_waitcnt vmcnt(0) between two DMAs and
// in front of LDS accesses it cannot tell apart), which serialises exactly what this is for.  The kernel waits for the DMA
// itself (chain_sync).  Extra outstanding operations the compiler does not know of only make its own vmcnt waits stricter.
__device__ __forceinline__ void lds_dma16(const void *src, const void *dst, int lanes32 = 0) {
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const lds_f32 *)dst);
    uint32_t m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(la), "v"(src) : "memory");
}
// workgroup barrier that also waits for this wavefront's DMA (and everything else it has in flight)
__device__ __forceinline__ void chain_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// the two halves of it, for the end of a layer: the DMA wait goes BEFORE the epilogue, so that the barrier behind the
// epilogue does not wait for the epilogue's stores
__device__ __forceinline__ void chain_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void chain_barrier_only() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS-DMA of chunk c (k-steps 2c, 2c + 1) of a layer's panel (KS k-steps, NT tiles) into a slot: 6 NT pieces of 1 KB, dealt
// round robin to the 4 wavefronts; with c == 0 also the bias tail (512 bytes).  No registers involved.
template <int NT, int NS>
__device__ __forceinline__ void chain_issue(const uint4 *panel, int KS, int s0, uint4 *slot, float *bias_dst, int wave, int lane) {
    constexpr int pieces = 3 * NS * NT;                    // 1 KB each: [plane][k-step s0 .. s0 + NS)[tile]
#pragma unroll
    for (int i = 0; i < (pieces + kChainWaves - 1) / kChainWaves; ++i) {
        const int q = wave + kChainWaves * i;              // (scalar)
        if (q < pieces) {
            const int pl = q / (NS * NT), rem = q - pl * (NS * NT);
            lds_dma16(panel + ((size_t)(pl * KS + s0) * NT + rem) * 64 + lane, slot + q * 64);
        }
    }
    if (s0 == 0 && wave == 0 && lane < 32) lds_dma16(panel + (size_t)3 * KS * NT * 64 + lane, bias_dst);
}

// The MFMAs of one chunk: k-steps S0 .. S0 + NS - 1 of the layer against the activations in registers 8 S0 .. 8 (S0 + NS) - 1.
template <int NT, int S0, int NS, class Mid>
__device__ __forceinline__ void chain_chunk(const uint4 *slot, const float (&act)[64], f32x16 (&acc)[4], int lane, Mid &&mid) {
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        if (sl == NS / 2) mid();                           // (half way: the second half of the wavefronts starts its DMA here)
        uint32_t b0[4], b1[4], b2[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bf16_split2(act[8 * (S0 + sl) + 2 * jj], act[8 * (S0 + sl) + 2 * jj + 1], b0[jj], b1[jj], b2[jj]);
        const bf16x8_t x0 = __builtin_bit_cast(bf16x8_t, make_uint4(b0[0], b0[1], b0[2], b0[3]));
        const bf16x8_t x1 = __builtin_bit_cast(bf16x8_t, make_uint4(b1[0], b1[1], b1[2], b1[3]));
        const bf16x8_t x2 = __builtin_bit_cast(bf16x8_t, make_uint4(b2[0], b2[1], b2[2], b2[3]));
        constexpr int pstride = NS * NT * 64;              // units per plane in the slot
        if constexpr (NT == 1) {
            const uint4 *wa = slot + sl * 64 + lane;
            const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, wa[0]), a1 = __builtin_bit_cast(bf16x8_t, wa[pstride]),
                           a2 = __builtin_bit_cast(bf16x8_t, wa[2 * pstride]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x2, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x1, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x0, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x1, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x0, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x0, acc[0], 0, 0, 0);
        } else {
            if constexpr (NT & 1) {                        // odd tile count: the last tile on its own
                const uint4 *wa = slot + (sl * NT + NT - 1) * 64 + lane;
                const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, wa[0]), a1 = __builtin_bit_cast(bf16x8_t, wa[pstride]),
                               a2 = __builtin_bit_cast(bf16x8_t, wa[2 * pstride]);
                acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x2, acc[NT - 1], 0, 0, 0);
                acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x1, acc[NT - 1], 0, 0, 0);
                acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x0, acc[NT - 1], 0, 0, 0);
                acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x1, acc[NT - 1], 0, 0, 0);
                acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x0, acc[NT - 1], 0, 0, 0);
                acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x0, acc[NT - 1], 0, 0, 0);
            }
#pragma unroll
            for (int tp = 0; tp + 1 < NT; tp += 2) {       // two tiles at a time: their accumulators are independent
                const uint4 *wa = slot + (sl * NT + tp) * 64 + lane, *wb = wa + 64;
                const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, wa[0]), a1 = __builtin_bit_cast(bf16x8_t, wa[pstride]),
                               a2 = __builtin_bit_cast(bf16x8_t, wa[2 * pstride]);
                const bf16x8_t c0 = __builtin_bit_cast(bf16x8_t, wb[0]), c1 = __builtin_bit_cast(bf16x8_t, wb[pstride]),
                               c2 = __builtin_bit_cast(bf16x8_t, wb[2 * pstride]);
                // smallest products first (weights = first operand: output features on the rows)
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x2, acc[tp], 0, 0, 0);
                acc[tp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, x2, acc[tp + 1], 0, 0, 0);
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x1, acc[tp], 0, 0, 0);
                acc[tp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, x1, acc[tp + 1], 0, 0, 0);
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x0, acc[tp], 0, 0, 0);
                acc[tp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, x0, acc[tp + 1], 0, 0, 0);
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x1, acc[tp], 0, 0, 0);
                acc[tp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, x1, acc[tp + 1], 0, 0, 0);
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x0, acc[tp], 0, 0, 0);
                acc[tp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, x0, acc[tp + 1], 0, 0, 0);
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x0, acc[tp], 0, 0, 0);
                acc[tp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, x0, acc[tp + 1], 0, 0, 0);
            }
        }
    }
}

// Layer epilogue: bias, activation, sign bits; register r of tile t = feature 32 t + 8 (r >> 2) + 4 h + (r & 3) of row n.
// KEEP: the outputs become the next layer's input registers.
// BWD (the backward's input-gradient chain): no bias, no activation, no sign bits out; the output is multiplied by
// (bit ? 1 : mask_slope) from the forward's sign words of the layer below (mw) where the layer has a mask.
template <int NT, bool KEEP, bool BWD>
__device__ __forceinline__ void chain_epilogue(const ChainLayer &Ly, const float *bias_l, float *s_stage, int P, int row0, int lane,
                                               const f32x16 (&acc)[4], float (&act)[64], const uint32_t (&mw)[4], int abl) {
    const int h = lane >> 5, n = lane & 31;
    const float slope = Ly.slope, mslope = Ly.mask_slope;
    const bool masked = Ly.mask != nullptr;
    const int N = Ly.N;
    const bool vec = (N & 3) == 0;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(Ly.out, 0, (int)((uint32_t)P * (uint32_t)N * 4u), 0x00020000);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        uint32_t word = 0u;
        float4 bqs[4];
        if constexpr (!BWD)
            lds_read16x4_opaque(bias_l + 32 * t + 4 * h, bias_l + 32 * t + 8 + 4 * h, bias_l + 32 * t + 16 + 4 * h, bias_l + 32 * t + 24 + 4 * h, bqs);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v;
                if constexpr (BWD) {
                    v = acc[t][4 * q + c];
                    if (masked) v = (mw[t] & (1u << (8 * q + 4 * h + c))) ? v : mslope * v;       // (uniform branch)
                } else {
                    const float4 bq = bqs[q];
                    v = acc[t][4 * q + c] + (c == 0 ? bq.x : (c == 1 ? bq.y : (c == 2 ? bq.z : bq.w)));
                    const bool pos = v > 0.f;
                    word |= pos ? (1u << (8 * q + 4 * h + c)) : 0u;
                    v = pos ? v : slope * v;
                }
                y[c] = v;
                if constexpr (KEEP) act[16 * t + 4 * q + c] = v;       // = slot 16 t + r of the next layer's second operand
            }
            // rows are on the lanes here: a direct store would touch 32 lines per instruction.  Through the wavefront's LDS
            // tile instead ([row][feature]), read back with the features on the lanes
            lds_write16_opaque(s_stage + n * kChainStageLd + 8 * q + 4 * h, make_float4(y[0], y[1], y[2], y[3]));
        }
        float4 yvs[4];
        {
            const float *sp = s_stage + (lane >> 3) * kChainStageLd + 4 * (lane & 7);
            lds_read16x4_opaque(sp, sp + 8 * kChainStageLd, sp + 16 * kChainStageLd, sp + 24 * kChainStageLd, yvs);
        }
        if (vec) {                                           // (uniform) bounds-checked buffer stores: rows past P fall out in hardware
            const uint32_t voff = (((uint32_t)(row0 + (lane >> 3)) * (uint32_t)N) + 32 * t + 4 * (lane & 7)) * 4u;
            if (!(abl & 1)) {
#pragma unroll
                for (int i = 0; i < 4; ++i)             // 8 rows per instruction, 128 contiguous bytes each
                    if (32 * t + 4 * (lane & 7) < N)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_ev, yvs[i]), rsrc, voff + (uint32_t)(8 * i * N * 4), 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = 8 * i + (lane >> 3), m0 = 32 * t + 4 * (lane & 7);
                const float4 yv = yvs[i];
                if (row0 + rr < P && !(abl & 1)) {
                    float *dst = Ly.out + (size_t)(row0 + rr) * N + m0;
                    if (m0 < N) dst[0] = yv.x;
                    if (m0 + 1 < N) dst[1] = yv.y;
                    if (m0 + 2 < N) dst[2] = yv.z;
                    if (m0 + 3 < N) dst[3] = yv.w;
                }
            }
        }
        if (!BWD && Ly.sign) {                               // (uniform) the two lane halves hold complementary bits of the word
            word |= (uint32_t)__shfl_xor((int)word, 32);
            if (h == 0 && row0 + n < P && !(abl & 16)) Ly.sign[(size_t)(row0 + n) * NT + t] = word;
        }
    }
}

template <int NTL, bool BWD>
__global__ __launch_bounds__(kChainThreads, 2) void chain_fwd_kernel(int P, int K0, const float *__restrict__ X, ChainArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem_chain[];
    uint4 *const s_slot0 = reinterpret_cast<uint4 *>(smem_chain);
    float *s_bias = reinterpret_cast<float *>(smem_chain + (size_t)kChainSlots * kChainSlotUnits * 16);       // [3][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, n = lane & 31;
    const int nblocks = (P + kChainRows - 1) / kChainRows;
    float *s_stage = s_bias + 3 * 128 + wave * (32 * kChainStageLd);      // the wavefront's transposition tile: 32 rows x (32 + 4) floats
    const int L = args.L, abl = args.abl;
    const int KS0 = chain_ksteps(K0), nch0 = (KS0 + kChainCS - 1) / kChainCS;
    constexpr int kNch = 8 / kChainCS;                     // chunks of a 128-input layer
    const bool early = wave < kChainWaves / 2 || (abl & 32);           // (abl 32: every wavefront at the start of the chunk)
    auto slot = [&](int g) { return s_slot0 + (size_t)g * kChainSlotUnits; };
    // bias buffers: a layer's bias arrives with its first chunk, i.e. during the last chunk of the layer before -- for layer 0
    // of the NEXT row block that is the last layer of this one, whose epilogue is still to come.  Layers 0 .. L - 2 alternate
    // between two buffers, the last layer has its own.
    auto bias_of = [&](int l) { return s_bias + (l == L - 1 ? 2 : (l & 1)) * 128; };
    // chunk (l, c) = k-steps [CS c, CS c + ns): layers 0 .. L - 2 have 4 output tiles, the last one NTL; layer 0 has nch0
    // chunks (its last one may hold 2 k-steps of 4), the others 8 / CS full ones
    auto issue = [&](int l, int c, int g) {
        if (abl & 2) return;
        const uint4 *panel = args.layer[l].panel;
        const int KS = l == 0 ? KS0 : 8;
        const bool full = kChainCS == 2 || KS - kChainCS * c >= kChainCS;
        if (l == L - 1) chain_issue<NTL, kChainCS>(panel, KS, kChainCS * c, slot(g), bias_of(l), wave, lane);
        else if (full) chain_issue<4, kChainCS>(panel, KS, kChainCS * c, slot(g), bias_of(l), wave, lane);
        else chain_issue<4, 2>(panel, KS, kChainCS * c, slot(g), bias_of(l), wave, lane);
    };
    if ((int)blockIdx.x < nblocks) issue(0, 0, 0);
    chain_sync();                                          // chunk 0 is in place
    int g = 0;
    for (int rb = (int)blockIdx.x; rb < nblocks; rb += (int)gridDim.x) {
        const int row0 = rb * kChainRows + wave * 32;
        const uint32_t rowc = (uint32_t)(row0 + n < P ? row0 + n : P - 1);
        float act[64];
        {   // the wavefront's 32 input rows, straight into the second-operand order of layer 0.  Branch-free (every load in
            // flight before the first wait): a group of 4 features that runs past the row is read from the row's last 16
            // bytes instead (K0 >= 4) and shifted into place
            const float *xr = X + (size_t)rowc * (uint32_t)K0;
            if (K0 < 4) {                                  // (uniform) up to three inputs: element loads
#pragma unroll
                for (int u = 0; u < 64; ++u) act[u] = 0.f;
                const float e0 = xr[0], e1 = xr[K0 > 1 ? 1 : 0], e2 = xr[K0 > 2 ? 2 : 0];
                act[0] = h == 0 ? e0 : 0.f;
                act[1] = (h == 0 && K0 > 1) ? e1 : 0.f;
                act[2] = (h == 0 && K0 > 2) ? e2 : 0.f;
            } else
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                Float4U raw[8];
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    const int f0 = 8 * (8 * half + v) + 4 * h;
                    raw[v] = *reinterpret_cast<const Float4U *>(xr + min(f0, K0 - 4));
                }
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    const int u = 8 * half + v, f0 = 8 * u + 4 * h, sh = f0 - min(f0, K0 - 4);       // 0 inside the row
                    const float r0 = raw[v].v[0], r1 = raw[v].v[1], r2 = raw[v].v[2], r3 = raw[v].v[3];
                    act[4 * u + 0] = sh == 0 ? r0 : (sh == 1 ? r1 : (sh == 2 ? r2 : (sh == 3 ? r3 : 0.f)));
                    act[4 * u + 1] = sh == 0 ? r1 : (sh == 1 ? r2 : (sh == 2 ? r3 : 0.f));
                    act[4 * u + 2] = sh == 0 ? r2 : (sh == 1 ? r3 : 0.f);
                    act[4 * u + 3] = sh == 0 ? r3 : 0.f;
                }
            }
        }
        const bool more_blocks = rb + (int)gridDim.x < nblocks;
        f32x16 acc[4];
        uint32_t mw[4] = {0u, 0u, 0u, 0u};                 // the row's mask words of the current layer (backward chain)
        auto load_mask = [&](const ChainLayer &Ly, int nt) {          // issued one chunk before the epilogue that uses them
            if (!Ly.mask) return;
            const uint32_t *mp = Ly.mask + (size_t)rowc * nt;
            if (nt == 4) { const uint4 q = *reinterpret_cast<const uint4 *>(mp); mw[0] = q.x; mw[1] = q.y; mw[2] = q.z; mw[3] = q.w; }
            else for (int t = 0; t < nt; ++t) mw[t] = mp[t];
        };
        // ---- layers 0 .. L - 2: 128 outputs
        for (int l = 0; l + 1 < L; ++l) {
            const int nch = l == 0 ? nch0 : kNch;
            const int KS = l == 0 ? KS0 : 8;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            static_for_4([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if (c < kNch && c < nch) {                  // (uniform)
                    // the next chunk goes into the other slot: every wavefront finished reading it before the barrier that
                    // ended the previous chunk
                    // the DMA instructions hold the issuing wavefront for a long time (measured: ~100 us of the kernel whatever
                    // the source, the depth of the prefetch or the piece size); the two wavefronts of a SIMD (w and w + 4)
                    // therefore issue theirs half a chunk apart, so that one of them always feeds the matrix pipe
                    auto next = [&]() {
                        if (c + 1 < nch) issue(l, c + 1, g ^ 1);
                        else issue(l + 1, 0, g ^ 1);
                    };
                    if (early) next();
                    if (c + 1 >= nch) load_mask(args.layer[l], 4);
                    const bool full = kChainCS == 2 || KS - kChainCS * c >= kChainCS;      // (layer 0's last chunk: 2 of 4 k-steps)
                    if (abl & 4) { if (!early) next(); }
                    else if (full) chain_chunk<4, kChainCS * c, kChainCS>(slot(g), act, acc, lane, [&]() { if (!early) next(); });
                    else chain_chunk<4, kChainCS * c, 2>(slot(g), act, acc, lane, [&]() { if (!early) next(); });
                    if (c + 1 < nch) {
                        chain_sync();      // this slot is free again, and the next chunk's DMA has landed
                        g ^= 1;
                    }
                }
            });
            chain_wait_dma();         // the next layer's first chunk (and bias): this wavefront's pieces have landed
            chain_epilogue<4, true, BWD>(args.layer[l], bias_of(l), s_stage, P, row0, lane, acc, act, mw, abl);
            chain_barrier_only();     // ... everybody's have, and the layer's last slot is free; the stores stay in flight
            g ^= 1;
        }
        // ---- the last layer: NTL output tiles, 128 inputs
        {
            const int l = L - 1;
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            static_for_4([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (c < kNch) {
                    auto next = [&]() {
                        if (c + 1 < kNch) issue(l, c + 1, g ^ 1);
                        else if (more_blocks) issue(0, 0, g ^ 1);
                    };
                    if (early) next();
                    if (c + 1 >= kNch) load_mask(args.layer[l], NTL);
                    if (!(abl & 4)) chain_chunk<NTL, kChainCS * c, kChainCS>(slot(g), act, acc, lane, [&]() { if (!early) next(); });
                    else if (!early) next();
                    if (c + 1 < kNch) {
                        chain_sync();
                        g ^= 1;
                    }
                }
            });
            chain_wait_dma();
            chain_epilogue<NTL, false, BWD>(args.layer[l], bias_of(l), s_stage, P, row0, lane, acc, act, mw, abl);
            chain_barrier_only();
            g ^= 1;
        }
    }
}

}  // namespace d3ga

using namespace d3ga;

extern "C" int64_t d3ga_mlp_panel_bytes(int32_t K, int32_t n_out) {
    if (K < 1 || K > kMlpMaxK || n_out < 1 || n_out > 128) return D3GA_E_SIZE;
    return (int64_t)mlp_panel_units(K, n_out) * 16;
}

extern "C" int d3ga_mlp_pack_weights(int32_t K, int32_t n_out, const float *W, int64_t ld_k, int64_t ld_n, void *panel,
                                     d3ga_stream_t stream) {
    if (K < 1 || K > kMlpMaxK || n_out < 1 || n_out > 128) return D3GA_E_SIZE;
    if (!W || !panel) return D3GA_E_NULL;
    if ((uintptr_t)panel & 15) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const int units = mlp_panel_units(K, n_out) / 3;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((units + kBlock - 1) / kBlock), dim3(kBlock), 0, s, K, n_out, W, ld_k, ld_n,
                       reinterpret_cast<uint4 *>(panel));
    return check_launch(s, 0);
}

// Y (P, n_out) = act( X (P,K) * W + bias ) [ (.) (mask bit ? 1 : mask_slope) ];  sign_out: bit (Y > 0) per element
// panel: the bf16 weight planes written by d3ga_mlp_pack_weights for the same (K, n_out).
extern "C" int d3ga_mlp_linear(int32_t P, int32_t K, int32_t n_out, const float *X, const void *panel, const float *bias,
                               float out_slope, uint32_t *sign_out, const uint32_t *mask_bits, float mask_slope, float *Y,
                               d3ga_stream_t stream) {
    if (P < 0 || K < 1 || K > kMlpMaxK || n_out < 1 || n_out > 128) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!X || !panel || !Y) return D3GA_E_NULL;
    if ((((uintptr_t)X | (uintptr_t)panel) & 15) != 0) return D3GA_E_CONFIG;
    if ((int64_t)P * K >= (1ll << 31) || (int64_t)P * n_out >= (1ll << 31)) return D3GA_E_SIZE;      // 32-bit element offsets
    hipStream_t s = (hipStream_t)stream;
    const int NB = (n_out + 31) / 32;
    const uint4 *Wp = reinterpret_cast<const uint4 *>(panel);
    const size_t lds = (size_t)mlp_panel_units(K, n_out) * 16 + (size_t)(kMlpThreads / 64) * kMlpABytes;
    const bool vec = (K % 4) == 0;
#define D3GA_MLP_LAUNCH3(NBV, VECV, MASKV, RAGV, PV, XV, SV, MV, YV)                                                  \
    do {                                                                                                              \
        static bool attr[64] = {};                                                                                    \
        int dev = 0;                                                                                                  \
        D3GA_HIP(hipGetDevice(&dev));                                                                                 \
        if (dev >= 0 && dev < 64 && !attr[dev]) {                                                                     \
            D3GA_HIP(hipFuncSetAttribute((const void *)linear_kernel<NBV, VECV, MASKV, RAGV>,                         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize,                                  \
                                         mlp_panel_units(kMlpMaxK, 32 * NBV) * 16 + (kMlpThreads / 64) * kMlpABytes));  \
            attr[dev] = true;                                                                                         \
        }                                                                                                             \
        const int nt = ((PV) + kMlpRows - 1) / kMlpRows;                                                              \
        hipLaunchKernelGGL((linear_kernel<NBV, VECV, MASKV, RAGV>), dim3(nt < 256 ? nt : 256), dim3(kMlpThreads), lds, \
                           s, (PV), K, n_out, (XV), Wp, bias, out_slope, (SV), (MV), mask_slope, (YV));               \
    } while (0)
    // rows [0, P_full): no bounds checks at all (straight-line loads and stores); the ragged remainder (< 32 rows), or
    // everything when n_out is not a multiple of 32, goes through the bounds-checked instantiation
    const int P_full = (n_out % 32 == 0) ? P - P % 32 : 0;
    const int P_rest = P - P_full;
    const size_t xo = (size_t)P_full * K, yo = (size_t)P_full * n_out, wo = (size_t)P_full * NB;
#define D3GA_MLP_LAUNCH2(NBV, VECV, MASKV)                                                                            \
    do {                                                                                                              \
        if (P_full > 0) D3GA_MLP_LAUNCH3(NBV, VECV, MASKV, false, P_full, X, sign_out, mask_bits, Y);                 \
        if (P_rest > 0)                                                                                               \
            D3GA_MLP_LAUNCH3(NBV, VECV, MASKV, true, P_rest, X + xo, sign_out ? sign_out + wo : nullptr,              \
                             mask_bits ? mask_bits + wo : nullptr, Y + yo);                                           \
    } while (0)
#define D3GA_MLP_LAUNCH(NBV)                                                                                          \
    do {                                                                                                              \
        if (vec && mask_bits) D3GA_MLP_LAUNCH2(NBV, true, true);                                                       \
        else if (vec) D3GA_MLP_LAUNCH2(NBV, true, false);                                                             \
        else if (mask_bits) D3GA_MLP_LAUNCH2(NBV, false, true);                                                           \
        else D3GA_MLP_LAUNCH2(NBV, false, false);                                                                     \
    } while (0)
    switch (NB) {
        case 1: D3GA_MLP_LAUNCH(1); break;
        case 2: D3GA_MLP_LAUNCH(2); break;
        case 3: D3GA_MLP_LAUNCH(3); break;
        default: D3GA_MLP_LAUNCH(4); break;
    }
#undef D3GA_MLP_LAUNCH
#undef D3GA_MLP_LAUNCH2
#undef D3GA_MLP_LAUNCH3
    return check_launch(s, 0);
}

// dW (N,K) (+)= dPre^T . X and (optionally) db (N) (+)= column sums of dPre
static int mlp_wgrad_launch(int32_t P, int32_t N, int32_t K, const float *dpre, const float *X, float *dW, float *db,
                            bool zero_first, d3ga_stream_t stream) {
    if (P < 0 || N < 1 || N > 128 || K < 1 || K > 128) return D3GA_E_SIZE;
    if (!dW) return D3GA_E_NULL;
    if ((int64_t)P * N >= (1ll << 31) || (int64_t)P * K >= (1ll << 31)) return D3GA_E_SIZE;
    hipStream_t s = (hipStream_t)stream;
    if (zero_first) {
        D3GA_HIP(zero_async(dW, sizeof(float) * (size_t)N * K, s));
        if (db) D3GA_HIP(zero_async(db, sizeof(float) * (size_t)N, s));
    }
    if (P == 0) return D3GA_OK;
    if (!dpre || !X) return D3GA_E_NULL;
    const int NBn = (N + 31) / 32, NBk = (K + 31) / 32;
    int grid = 256;                                                   // row ranges; every workgroup ends with N*K atomics
    int rows = (P + grid - 1) / grid;
    rows = ((rows + kWgRows - 1) / kWgRows) * kWgRows;                 // whole chunks
    grid = (P + rows - 1) / rows;
    const size_t lds = 2 * (size_t)kWgOperandBytes;
    const bool va = N % 2 == 0 && ((uintptr_t)dpre & 7) == 0, vb = K % 2 == 0 && ((uintptr_t)X & 7) == 0;   // float2 loads
    const bool mixed = N <= 16 || K <= 16;
    // D3GA_KNOB_WGRAD_WS: 0 the barrier-phased kernel everywhere; 1 (default) the wavefront-specialised kernel for two wide operands
    // AND for a narrow dPre against a wide activation (the fields' output layers, N <= 16: its idle lanes load a clamped column,
    // its spare multipliers leave at once -- 98 -> 84 us at N = 11, 92 -> 81 at N = 4, 500k rows, tools/time_wgrad.py; the other
    // way round, a wide dPre against a narrow input, the second loader layout of the barrier-phased kernel stays ahead: 109 vs 117);
    // 2: the wavefront-specialised kernel for every shape
    const int ws_knob = debug_knob(D3GA_KNOB_WGRAD_WS);
    const bool use_ws = ws_knob != 0;
    const bool ws_for_mixed = ws_knob == 2 || (ws_knob == 1 && N <= 16 && K > 16);
#define D3GA_WG(VA, VB, MX)                                                                                           \
    do {                                                                                                              \
        static bool attr[64] = {};                                                                                    \
        int dev = 0;                                                                                                  \
        D3GA_HIP(hipGetDevice(&dev));                                                                                 \
        if (dev >= 0 && dev < 64 && !attr[dev]) {                                                                     \
            D3GA_HIP(hipFuncSetAttribute((const void *)wgrad_kernel<VA, VB, MX>,                                      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
            attr[dev] = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((wgrad_kernel<VA, VB, MX>), dim3(grid), dim3(kWgThreads), lds, s, P, N, K, NBk, NBn * NBk, \
                           rows, dpre, X, dW, db);                                                                    \
    } while (0)
#define D3GA_WS(VA, VB)                                                                                               \
    do {                                                                                                              \
        static bool attr[64] = {};                                                                                    \
        int dev = 0;                                                                                                  \
        D3GA_HIP(hipGetDevice(&dev));                                                                                 \
        const size_t lds_ws = 2 * (size_t)kWsBufferUnits * 16;                                                        \
        if (dev >= 0 && dev < 64 && !attr[dev]) {                                                                     \
            D3GA_HIP(hipFuncSetAttribute((const void *)wgrad_ws_kernel<VA, VB>,                                       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ws));                   \
            attr[dev] = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((wgrad_ws_kernel<VA, VB>), dim3(grid), dim3(kWgThreads), lds_ws, s, P, N, K, NBk, NBn, rows, \
                           dpre, X, dW, db);                                                                          \
    } while (0)
#define D3GA_WG2(VA, VB)                                                                                              \
    do {                                                                                                              \
        if (mixed && !ws_for_mixed) D3GA_WG(VA, VB, true);                                                            \
        else if (use_ws) D3GA_WS(VA, VB);                                                                             \
        else D3GA_WG(VA, VB, false);                                                                                  \
    } while (0)
    if (va && vb) D3GA_WG2(true, true);
    else if (va) D3GA_WG2(true, false);
    else if (vb) D3GA_WG2(false, true);
    else D3GA_WG2(false, false);
#undef D3GA_WG2
#undef D3GA_WS
#undef D3GA_WG
    return check_launch(s, 0);
}

extern "C" int d3ga_mlp_wgrad(int32_t P, int32_t N, int32_t K, const float *dpre, const float *X, float *dW, float *db,
                              d3ga_stream_t stream) {
    return mlp_wgrad_launch(P, N, K, dpre, X, dW, db, true, stream);
}

extern "C" int d3ga_mlp_wgrad_acc(int32_t P, int32_t N, int32_t K, const float *dpre, const float *X, float *dW, float *db,
                                  d3ga_stream_t stream) {
    return mlp_wgrad_launch(P, N, K, dpre, X, dW, db, false, stream);
}


extern "C" int64_t d3ga_mlp_chain_panel_bytes(int32_t K, int32_t n_out) {
    if (K < 1 || K > 128 || n_out < 1 || n_out > 128) return D3GA_E_SIZE;
    return (int64_t)d3ga::chain_panel_units(K, n_out) * 16 + 512;
}

extern "C" int d3ga_mlp_pack_chain(int32_t K, int32_t n_out, const float *W, int64_t ld_k, int64_t ld_n, void *panel,
                                   d3ga_stream_t stream) {
    if (K < 1 || K > 128 || n_out < 1 || n_out > 128) return D3GA_E_SIZE;
    if (!W || !panel) return D3GA_E_NULL;
    if ((uintptr_t)panel & 15) return D3GA_E_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const int units = d3ga::chain_panel_units(K, n_out) / 3;
    hipLaunchKernelGGL(d3ga::pack_chain_kernel, dim3((units + d3ga::kBlock - 1) / d3ga::kBlock), dim3(d3ga::kBlock), 0, s, K, n_out, W,
                       ld_k, ld_n, reinterpret_cast<uint4 *>(panel));
    return d3ga::check_launch(s, 0);
}

extern "C" int d3ga_mlp_chain_fwd(int32_t P, int32_t K0, const float *X, int32_t L, const int32_t *Ks, const int32_t *Ns,
                                  void *const *panels, const float *const *biases, const float *slopes, float *const *outs,
                                  uint32_t *const *signs, const uint32_t *const *masks, const float *mask_slopes,
                                  d3ga_stream_t stream) {
    using namespace d3ga;
    if (P < 0 || L < 1 || L > kChainMaxLayers || K0 < 1 || K0 > 128) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!X || !Ks || !Ns || !panels || !slopes || !outs || !signs) return D3GA_E_NULL;
    ChainArgs a;
    ChainBiasArgs ba;
    a.L = L;
    int k_prev = K0;
    for (int l = 0; l < L; ++l) {
        if (Ks[l] != k_prev || Ns[l] < 1 || Ns[l] > 128) return D3GA_E_SIZE;      // layer l consumes what layer l - 1 produced
        if (!panels[l] || !outs[l]) return D3GA_E_NULL;
        if (((uintptr_t)panels[l] | (uintptr_t)outs[l]) & 15) return D3GA_E_CONFIG;
        // byte offsets in 32 bits -- INCLUDING the rows >= P of the last row block, which rely on the hardware bounds check:
        // their offsets must not wrap past 2^32 into the buffer's valid range
        if (((int64_t)P + d3ga::kChainRows) * Ns[l] >= (1ll << 30) || ((int64_t)P + d3ga::kChainRows) * Ks[l] >= (1ll << 30)) return D3GA_E_SIZE;
        a.layer[l] = ChainLayer{reinterpret_cast<const uint4 *>(panels[l]), outs[l], signs[l], masks ? masks[l] : nullptr, Ks[l], Ns[l],
                                slopes[l], (masks && masks[l] && mask_slopes) ? mask_slopes[l] : 1.f};
        if (masks && masks[l] && (Ns[l] + 31) / 32 == 4 && ((uintptr_t)masks[l] & 15)) return D3GA_E_CONFIG;
        ba.bias[l] = biases ? biases[l] : nullptr;
        ba.tail[l] = reinterpret_cast<float *>(reinterpret_cast<char *>(panels[l]) + (size_t)chain_panel_units(Ks[l], Ns[l]) * 16);
        ba.N[l] = Ns[l];
        k_prev = Ns[l];
    }
    // shapes the kernel is built for (everything else: the per-layer kernels): >= 2 layers, all but the last 128 wide
    if (L < 2) return D3GA_E_CONFIG;
    for (int l = 0; l + 1 < L; ++l) if (Ns[l] != 128) return D3GA_E_CONFIG;
    const int ntl = (Ns[L - 1] + 31) / 32;
    const int abl = debug_knob(D3GA_KNOB_CHAIN_ABL);      // timing ablations (wrong results)
    a.abl = abl;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)kChainSlots * kChainSlotUnits * 16 + (3 * 128 + (kChainThreads / 64) * 32 * kChainStageLd) * sizeof(float);
    static bool attr[64] = {};
    int dev = 0;
    D3GA_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr[dev]) {
        const void *ks[8] = {(const void *)chain_fwd_kernel<1, false>, (const void *)chain_fwd_kernel<2, false>, (const void *)chain_fwd_kernel<3, false>,
                             (const void *)chain_fwd_kernel<4, false>, (const void *)chain_fwd_kernel<1, true>, (const void *)chain_fwd_kernel<2, true>,
                             (const void *)chain_fwd_kernel<3, true>, (const void *)chain_fwd_kernel<4, true>};
        for (const void *k : ks) D3GA_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr[dev] = true;
    }
    // the backward's chain: no bias, no activation, no sign output anywhere -- its own instantiation without that arithmetic
    bool bwd = masks != nullptr;
    for (int l = 0; l < L && bwd; ++l) bwd = !signs[l] && slopes[l] == 1.f && !(biases && biases[l]);
    // this call's biases into the panels' tails.  (The BWD instantiation never reads a bias -- the tail still rides along with the
    // panel's DMA, whatever it holds: no launch for it, 5.7 us x 3 chains per colour step.)
    if (!bwd) hipLaunchKernelGGL(chain_bias_kernel, dim3(L), dim3(128), 0, s, ba);
    const int nblocks = (P + kChainRows - 1) / kChainRows;
    const int grid_cap = debug_knob(D3GA_KNOB_CHAIN_GRID) > 0 ? debug_knob(D3GA_KNOB_CHAIN_GRID) : 2048 / kChainWaves;
    const dim3 grid(nblocks < grid_cap ? nblocks : grid_cap), block(kChainThreads);
    if (masks && !bwd)                                     // masks together with bias / activation / sign output: not built
        for (int l = 0; l < L; ++l) if (masks[l]) return D3GA_E_CONFIG;
    if (bwd) {
        if (ntl == 1) hipLaunchKernelGGL((chain_fwd_kernel<1, true>), grid, block, lds, s, P, K0, X, a);
        else if (ntl == 2) hipLaunchKernelGGL((chain_fwd_kernel<2, true>), grid, block, lds, s, P, K0, X, a);
        else if (ntl == 3) hipLaunchKernelGGL((chain_fwd_kernel<3, true>), grid, block, lds, s, P, K0, X, a);
        else hipLaunchKernelGGL((chain_fwd_kernel<4, true>), grid, block, lds, s, P, K0, X, a);
    } else {
        if (ntl == 1) hipLaunchKernelGGL((chain_fwd_kernel<1, false>), grid, block, lds, s, P, K0, X, a);
        else if (ntl == 2) hipLaunchKernelGGL((chain_fwd_kernel<2, false>), grid, block, lds, s, P, K0, X, a);
        else if (ntl == 3) hipLaunchKernelGGL((chain_fwd_kernel<3, false>), grid, block, lds, s, P, K0, X, a);
        else hipLaunchKernelGGL((chain_fwd_kernel<4, false>), grid, block, lds, s, P, K0, X, a);
    }
    return check_launch(s, 0);
}

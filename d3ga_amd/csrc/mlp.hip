// mlp.hip -- the dense layer of the reference's field networks (models/mlp.py:39-110:
// DeformationField / CanonicalField / ColorField are all  z -> [Linear(128) + leaky_relu(0.1)] x (1 + n_layers) -> Linear).
//
// One kernel template does every GEMM of the forward and of the input-gradient chain of the backward:
//     Y[r][n] = act_out( sum_k A[r][k] * Wt[k][n] + bias[n] ),        A = X            (forward)
//                                                                     A = dY (.) lrelu'(H)   (backward; A is also stored)
// with exact f32 arithmetic on the matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf chain;
// 157 TFLOP/s peak on MI355X = the f32 vector peak, but no operand shuffling and no VALU slots spent on the products).
//
// Work decomposition: a workgroup (8 wavefronts) owns 256 rows, a wavefront 32 rows x all output columns (NB blocks of
// 32).  K is walked in chunks of 32: the wavefront moves its 32 x 32 chunk with full-segment loads into a private LDS
// buffer (next chunk prefetched into registers), then lane l takes row l & 31 and, of the chunk, k = 16 (l >> 5) + s
// (the 32x32x2 instruction contracts lanes 0-31's k with lanes 32-63's k, any pairing is a valid order of the sum).  The
// weight panel Wt (32*ceil(K/32) x 32*NB, zero padded, prepared by the host layer) sits in LDS for the whole
// persistent workgroup in the interleaved order [k][n & 31][n >> 5]: the NB operands a lane needs for one k-step are
// one 16-byte LDS read.
// Weight gradients (dW = dPre^T X, a reduction over all rows) are plain library GEMMs in the host layer (hipBLASLt).
#include "d3ga_internal.h"

namespace d3ga {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kMlpThreads = 1024;    // 16 wavefronts share one weight panel in LDS (138 KB: 4 wavefronts per SIMD)
constexpr int kMlpRows = 512;        // rows per workgroup (32 per wavefront)
constexpr int kMlpMaxK = 128;        // K <= 128
constexpr int kMlpXPitch = 36;       // floats per row of a wavefront's 32 x 32 activation chunk in LDS (16-byte aligned rows)

// One K-chunk (32 columns) of a wavefront's 32 rows, global -> registers in the COOPERATIVE order: lane l covers the
// 16 bytes (l & 7) of the 128-byte segment of row 8*j + (l >> 3), j = 0..3 -- every load instruction touches eight full
// 128-byte segments.  (A lane reading its own row straight from global memory touches 64 different cache lines per
// instruction and re-fetches every line 8 times from L2: measured 1.4 TB/s, the first version of this kernel.)
// chunk_issue only ISSUES the loads (raw values into registers, addresses clamped to stay in range): nothing may consume
// them here -- a select or the mask multiply right after the load makes the compiler wait for HBM on the spot and the
// "prefetch" degenerates into a blocking load (measured: launch time = MFMA time + HBM time).  chunk_commit, called one
// chunk later, applies the leaky_relu mask, zeroes what lies past the matrix, writes the masked operand back (a_out) and
// stores the chunk into the wavefront's LDS buffer.
template <bool VEC, bool MASK>
__device__ __forceinline__ void chunk_issue(float4 (&v)[4], float4 (&m)[4], int P, int K, int row0, int kc, int lane,
                                            const float *__restrict__ X, const float *__restrict__ mask) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = row0 + 8 * j + (lane >> 3), k = kc + 4 * (lane & 7);
        const int rc = r < P ? r : P - 1;
        if constexpr (VEC) {                                           // K % 4 == 0: rows are 16-byte aligned
            const uint32_t o = (uint32_t)rc * (uint32_t)K + (uint32_t)(k < K ? k : 0);
            v[j] = *reinterpret_cast<const float4 *>(X + o);
            if constexpr (MASK) m[j] = *reinterpret_cast<const float4 *>(mask + o);
        } else {
            float e[4], f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t o = (uint32_t)rc * (uint32_t)K + (uint32_t)(k + c < K ? k + c : 0);
                e[c] = X[o];
                if constexpr (MASK) f[c] = mask[o];
            }
            v[j] = make_float4(e[0], e[1], e[2], e[3]);
            if constexpr (MASK) m[j] = make_float4(f[0], f[1], f[2], f[3]);
        }
    }
}

template <bool VEC, bool MASK, bool RAGGED>
__device__ __forceinline__ void chunk_commit(float *s_x, float4 (&v)[4], float4 (&m)[4], int P, int K, int row0, int kc,
                                             int lane, float mask_slope, float *__restrict__ a_out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = row0 + 8 * j + (lane >> 3), k = kc + 4 * (lane & 7);
        float4 x = v[j];
        if constexpr (MASK) {
            x.x *= m[j].x > 0.f ? 1.f : mask_slope; x.y *= m[j].y > 0.f ? 1.f : mask_slope;
            x.z *= m[j].z > 0.f ? 1.f : mask_slope; x.w *= m[j].w > 0.f ? 1.f : mask_slope;
        }
        const bool ok_r = RAGGED ? r < P : true;                       // !RAGGED: the launcher guarantees P % 32 == 0
        if (RAGGED || (K & 31)) {                                      // (uniform) nothing to zero when K fills its chunks
            if (!(ok_r && k < K)) x.x = 0.f;
            if (!(ok_r && k + 1 < K)) x.y = 0.f;
            if (!(ok_r && k + 2 < K)) x.z = 0.f;
            if (!(ok_r && k + 3 < K)) x.w = 0.f;
        }
        if constexpr (MASK) {
            if (a_out && ok_r) {
                const uint32_t o = (uint32_t)r * (uint32_t)K + (uint32_t)k;
                if (VEC) { if (k < K) *reinterpret_cast<float4 *>(a_out + o) = x; }
                else {
                    if (k < K) a_out[o] = x.x;
                    if (k + 1 < K) a_out[o + 1] = x.y;
                    if (k + 2 < K) a_out[o + 2] = x.z;
                    if (k + 3 < K) a_out[o + 3] = x.w;
                }
            }
        }
        *reinterpret_cast<float4 *>(s_x + (8 * j + (lane >> 3)) * kMlpXPitch + 4 * (lane & 7)) = x;
    }
}

template <int NB, bool VEC, bool MASK, bool RAGGED>
__global__ __launch_bounds__(kMlpThreads) void linear_kernel(int P, int K, int n_store, const float *__restrict__ X,
                                                             const float *__restrict__ mask, float mask_slope,
                                                             float *__restrict__ a_out, const float *__restrict__ Wt,
                                                             const float *__restrict__ bias, float out_slope,
                                                             float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // weight panel [KP][32*NB] | 8 x activation chunk
    constexpr int N32 = 32 * NB;
    const int KP = 32 * ((K + 31) / 32);                               // panel rows (K padded to the chunk size)
    float *s_w = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *s_x = smem + KP * N32 + wave * (32 * kMlpXPitch);           // this wavefront's [32 rows][32 k] chunk, padded
    const int half = lane >> 5, l32 = lane & 31;
    {   // weight panel: contiguous copy, 16 bytes per thread and step
        const int nvec = KP * N32 / 4;
        for (int v = tid; v < nvec; v += kMlpThreads) reinterpret_cast<float4 *>(s_w)[v] = reinterpret_cast<const float4 *>(Wt)[v];
    }
    __syncthreads();
    const int ntiles = (P + kMlpRows - 1) / kMlpRows;
    // this lane's bias values, once (a global load inside the epilogue would force a vmcnt(0) in front of the stores)
    float bias_r[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias_r[nb] = (bias && l32 + 32 * nb < n_store) ? bias[l32 + 32 * nb] : 0.f;
    float4 nxt[4], nxm[4];
    chunk_issue<VEC, MASK>(nxt, nxm, P, K, blockIdx.x * kMlpRows + wave * 32, 0, lane, X, mask);
    auto do_tile = [&](int tile) __attribute__((always_inline)) {
        const int row0 = tile * kMlpRows + wave * 32;
        if (!RAGGED && row0 >= P) return;                              // wave-uniform: the last tile may be partly empty
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        for (int kc = 0; kc < KP; kc += 32) {
            __builtin_amdgcn_wave_barrier();                           // previous chunk's LDS reads are done
            chunk_commit<VEC, MASK, RAGGED>(s_x, nxt, nxm, P, K, row0, kc, lane, mask_slope, a_out);
            // prefetch (always exactly one chunk_issue, no branch): the next chunk of this tile, or -- BEFORE this tile's
            // epilogue stores are issued -- the first chunk of the workgroup's next tile (past the last tile: clamped rows)
            {
                const bool same = kc + 32 < KP;
                chunk_issue<VEC, MASK>(nxt, nxm, P, K, same ? row0 : row0 + (int)gridDim.x * kMlpRows, same ? kc + 32 : 0,
                                       lane, X, mask);
            }
            __builtin_amdgcn_wave_barrier();
            // A operand: row l32, k = kc + 16*half + s  (the instruction contracts lanes 0-31's k with lanes 32-63's k)
            float a[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 t = *reinterpret_cast<const float4 *>(s_x + l32 * kMlpXPitch + 16 * half + 4 * j);
                a[4 * j] = t.x; a[4 * j + 1] = t.y; a[4 * j + 2] = t.z; a[4 * j + 3] = t.w;
            }
            // panel layout [k][l32][nb]: the NB operands of one k-step are contiguous for a lane (one 16-byte LDS read)
            const float *wrow = s_w + ((kc + 16 * half) * 32 + l32) * NB;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float bv[NB];
                if constexpr (NB == 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(wrow + s * N32);
                    bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
                } else if constexpr (NB == 2) {
                    const float2 t = *reinterpret_cast<const float2 *>(wrow + s * N32);
                    bv[0] = t.x; bv[1] = t.y;
                } else {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) bv[nb] = wrow[s * N32 + nb];
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bv[nb], acc[nb], 0, 0, 0);
            }
        }
        // epilogue: C/D layout of the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5);
        // every store instruction writes two full 128-byte row segments
        constexpr bool full = !RAGGED;                                 // launcher: P % 32 == 0 and n_store == 32*NB
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = l32 + 32 * nb;
            const float b = bias_r[nb];
            const uint32_t ybase = (uint32_t)(row0 + 4 * half) * (uint32_t)n_store + (uint32_t)n;
            if constexpr (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    float y = acc[nb][r] + b;
                    y = y > 0.f ? y : out_slope * y;
                    Y[ybase + (uint32_t)dr * (uint32_t)n_store] = y;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    float y = acc[nb][r] + b;
                    y = y > 0.f ? y : out_slope * y;
                    if (row0 + 4 * half + dr < P && n < n_store) Y[ybase + (uint32_t)dr * (uint32_t)n_store] = y;
                }
            }
        }
    };
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += gridDim.x) do_tile(tile);
}

// Weight gradient dW (N,K) += dPre^T (N x rows) . X (rows x K): the contraction runs over the ROWS.  A workgroup owns a
// contiguous row range and walks it in chunks of 64 rows: both operands of the chunk are fetched ONCE with full-width
// loads into registers (next chunk) / LDS (current chunk), then one wavefront per 32x32 block of dW (NBn x NBk wavefronts)
// reads dPre[row][32 nb + (l & 31)] and X[row][32 kb + (l & 31)] for row = 2 s + (l >> 5) from LDS.  (First version: every
// wavefront fetched its operands from global memory itself -- each 128-byte segment was requested by four wavefronts at
// different times, 2 GB of L2 traffic per 128x128 layer: 311 us at 500k rows.)  Partial results meet in global memory
// through float atomics (dW zeroed by the launcher); the kb == 0 wavefronts also accumulate the bias gradient.
constexpr int kWgRows = 64;
constexpr int kWgThreads = 1024;     // 16 wavefronts: NBn x NBk of them own a block of dW, all of them move data

// one operand array of a chunk: global -> registers (issue) -> LDS (commit).  VEC: 16-byte pieces (width % 4 == 0),
// two per thread; else scalars, eight per thread (64 x 128 floats / 1024 threads).
template <bool VEC>
struct WgPiece { float4 v[VEC ? 2 : 1]; float s[VEC ? 1 : 8]; };
template <bool VEC>
__device__ __forceinline__ void wg_issue(WgPiece<VEC> &p, const float *__restrict__ src, int width, int r0, int r_end, int tid) {
    if constexpr (VEC) {
        const int w4 = width / 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * kWgThreads, r = i / w4, c = i - r * w4;
            const int rr = min(r0 + min(r, kWgRows - 1), r_end - 1);
            p.v[u] = *reinterpret_cast<const float4 *>(src + (uint32_t)rr * (uint32_t)width + 4 * c);
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * kWgThreads, r = i / width, c = i - r * width;
            const int rr = min(r0 + min(r, kWgRows - 1), r_end - 1);
            p.s[u] = src[(uint32_t)rr * (uint32_t)width + c];
        }
    }
}
template <bool VEC>
__device__ __forceinline__ void wg_commit(const WgPiece<VEC> &p, float *dst, int pitch, int width, int r0, int r_end, int tid) {
    if constexpr (VEC) {                                               // rows past the range are stored as zeros
        const int w4 = width / 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * kWgThreads, r = i / w4, c = i - r * w4;
            if (r < kWgRows) *reinterpret_cast<float4 *>(dst + r * pitch + 4 * c) = (r0 + r < r_end) ? p.v[u] : make_float4(0, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * kWgThreads, r = i / width, c = i - r * width;
            if (r < kWgRows) dst[r * pitch + c] = (r0 + r < r_end) ? p.s[u] : 0.f;
        }
    }
}

template <bool VECA, bool VECB>
__global__ __launch_bounds__(kWgThreads) void wgrad_kernel(int P, int N, int K, int NBk, int n_blocks, int rows_per_block,
                                                           const float *__restrict__ dpre, const float *__restrict__ X,
                                                           float *__restrict__ dW, float *__restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // s_a [64][NP] | s_b [64][KP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nthreads = kWgThreads;
    const bool worker = wave < n_blocks;                               // wave-uniform: owns a 32x32 block of dW
    const int nb = wave / NBk, kb = wave - nb * NBk;
    const int half = lane >> 5, l32 = lane & 31;
    const int NP = 32 * ((N + 31) / 32), KP = 32 * ((K + 31) / 32);
    float *s_a = smem, *s_b = smem + kWgRows * NP;
    const int n = 32 * nb + l32, k = 32 * kb + l32;
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(P, r_begin + rows_per_block);
    WgPiece<VECA> pa_r;
    WgPiece<VECB> pb_r;
    auto issue = [&](int r0) __attribute__((always_inline)) {
        wg_issue<VECA>(pa_r, dpre, N, r0, r_end, tid);
        wg_issue<VECB>(pb_r, X, K, r0, r_end, tid);
    };
    auto commit = [&](int r0) __attribute__((always_inline)) {
        wg_commit<VECA>(pa_r, s_a, NP, N, r0, r_end, tid);
        wg_commit<VECB>(pb_r, s_b, KP, K, r0, r_end, tid);
    };
    // zero the pad columns once (they are never written by commit)
    for (int i = tid; i < kWgRows * (NP + KP); i += nthreads) smem[i] = 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    if (r_begin < r_end) issue(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += kWgRows) {
        __syncthreads();                                               // previous chunk's LDS reads are done
        commit(r0);
        __syncthreads();
        if (r0 + kWgRows < r_end) issue(r0 + kWgRows);                 // prefetch while the MFMAs below run
        if (worker) {
            const float *pa = s_a + half * NP + n, *pb = s_b + half * KP + k;
#pragma unroll 8
            for (int s2 = 0; s2 < kWgRows / 2; ++s2) {
                const float a = pa[2 * s2 * NP], b = pb[2 * s2 * KP];
                bsum += a;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
    }
    if (!worker) return;
    // C/D layout: col = lane & 31 -> k, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> n
    if (k < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (nn < N) atomicAdd(dW + (uint32_t)nn * (uint32_t)K + (uint32_t)k, acc[r]);
        }
    }
    if (db && kb == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (half == 0 && n < N) atomicAdd(db + n, bsum);
    }
}

}  // namespace d3ga

using namespace d3ga;

// Y (P, n_out) = act( A (P,K) * Wt + bias ),  A = X, or X (.) lrelu'(mask) when mask != NULL (A then also written to a_out).
// Wt: (2*ceil(K/2), 32*ceil(n_out/32)) row-major, zero padded: Wt[k][n] = weight of input k for output n.
extern "C" int d3ga_mlp_linear(int32_t P, int32_t K, int32_t n_out, const float *X, const float *mask, float mask_slope,
                               float *a_out, const float *Wt, const float *bias, float out_slope, float *Y,
                               d3ga_stream_t stream) {
    if (P < 0 || K < 1 || K > kMlpMaxK || n_out < 1 || n_out > 128) return D3GA_E_SIZE;
    if (P == 0) return D3GA_OK;
    if (!X || !Wt || !Y) return D3GA_E_NULL;
    if (a_out && !mask) return D3GA_E_CONFIG;
    if ((((uintptr_t)X | (uintptr_t)Wt | (uintptr_t)mask | (uintptr_t)a_out) & 15) != 0) return D3GA_E_CONFIG;
    if ((int64_t)P * K >= (1ll << 31) || (int64_t)P * n_out >= (1ll << 31)) return D3GA_E_SIZE;      // 32-bit element offsets
    hipStream_t s = (hipStream_t)stream;
    const int KP = 32 * ((K + 31) / 32), NB = (n_out + 31) / 32;
    const size_t lds = ((size_t)KP * 32 * NB + (size_t)(kMlpThreads / 64) * 32 * kMlpXPitch) * sizeof(float);
    const bool vec = (K % 4) == 0;
#define D3GA_MLP_LAUNCH3(NBV, VECV, MASKV, RAGV, PV, XV, MV, AV, YV)                                                  \
    do {                                                                                                              \
        static bool attr[64] = {};                                                                                    \
        int dev = 0;                                                                                                  \
        D3GA_HIP(hipGetDevice(&dev));                                                                                 \
        if (dev >= 0 && dev < 64 && !attr[dev]) {                                                                     \
            D3GA_HIP(hipFuncSetAttribute((const void *)linear_kernel<NBV, VECV, MASKV, RAGV>,                         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize,                                  \
                                         (kMlpMaxK * 32 * NBV + (kMlpThreads / 64) * 32 * kMlpXPitch) * (int)sizeof(float))); \
            attr[dev] = true;                                                                                         \
        }                                                                                                             \
        const int nt = ((PV) + kMlpRows - 1) / kMlpRows;                                                              \
        hipLaunchKernelGGL((linear_kernel<NBV, VECV, MASKV, RAGV>), dim3(nt < 256 ? nt : 256), dim3(kMlpThreads), lds, \
                           s, (PV), K, n_out, (XV), (MV), mask_slope, (AV), Wt, bias, out_slope, (YV));               \
    } while (0)
    // rows [0, P_full): no bounds checks at all (straight-line loads and stores); the ragged remainder (< 32 rows), or
    // everything when n_out is not a multiple of 32, goes through the bounds-checked instantiation
    const int P_full = (n_out % 32 == 0) ? P - P % 32 : 0;
    const int P_rest = P - P_full;
    const size_t xo = (size_t)P_full * K, yo = (size_t)P_full * n_out;
#define D3GA_MLP_LAUNCH2(NBV, VECV, MASKV)                                                                            \
    do {                                                                                                              \
        if (P_full > 0) D3GA_MLP_LAUNCH3(NBV, VECV, MASKV, false, P_full, X, mask, a_out, Y);                         \
        if (P_rest > 0)                                                                                               \
            D3GA_MLP_LAUNCH3(NBV, VECV, MASKV, true, P_rest, X + xo, mask ? mask + xo : nullptr,                      \
                             a_out ? a_out + xo : nullptr, Y + yo);                                                   \
    } while (0)
#define D3GA_MLP_LAUNCH(NBV)                                                                                          \
    do {                                                                                                              \
        if (vec && mask) D3GA_MLP_LAUNCH2(NBV, true, true);                                                           \
        else if (vec) D3GA_MLP_LAUNCH2(NBV, true, false);                                                             \
        else if (mask) D3GA_MLP_LAUNCH2(NBV, false, true);                                                            \
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

// dW (N,K) = dPre^T . X and (optionally) db (N) = column sums of dPre; both outputs are zeroed by the call.
extern "C" int d3ga_mlp_wgrad(int32_t P, int32_t N, int32_t K, const float *dpre, const float *X, float *dW, float *db,
                              d3ga_stream_t stream) {
    if (P < 0 || N < 1 || N > 128 || K < 1 || K > 128) return D3GA_E_SIZE;
    if (!dW) return D3GA_E_NULL;
    if ((int64_t)P * N >= (1ll << 31) || (int64_t)P * K >= (1ll << 31)) return D3GA_E_SIZE;
    hipStream_t s = (hipStream_t)stream;
    D3GA_HIP(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, s));
    if (db) D3GA_HIP(hipMemsetAsync(db, 0, sizeof(float) * (size_t)N, s));
    if (P == 0) return D3GA_OK;
    if (!dpre || !X) return D3GA_E_NULL;
    const int NBn = (N + 31) / 32, NBk = (K + 31) / 32;
    int grid = 256;                                                   // row ranges; every workgroup ends with N*K atomics
    int rows = (P + grid - 1) / grid;
    rows = ((rows + kWgRows - 1) / kWgRows) * kWgRows;                 // whole chunks
    grid = (P + rows - 1) / rows;
    const size_t lds = (size_t)kWgRows * (32 * NBn + 32 * NBk) * sizeof(float);
    const bool va = N % 4 == 0, vb = K % 4 == 0;                       // 16-byte pieces need 16-byte aligned rows
#define D3GA_WG(VA, VB)                                                                                               \
    hipLaunchKernelGGL((wgrad_kernel<VA, VB>), dim3(grid), dim3(kWgThreads), lds, s, P, N, K, NBk, NBn * NBk, rows, dpre, \
                       X, dW, db)
    if (va && vb) D3GA_WG(true, true);
    else if (va) D3GA_WG(true, false);
    else if (vb) D3GA_WG(false, true);
    else D3GA_WG(false, false);
#undef D3GA_WG
    return check_launch(s, 0);
}

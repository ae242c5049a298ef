// valu_issue.hip -- calibration of VALU / DPP / transcendental / LDS-atomic issue cost on gfx950.
//
// Every wavefront runs ITER iterations of a block of 64 INDEPENDENT instructions of one kind (16 registers x 4), so the
// stream is issue-bound, never dependency-bound.  Launched with 1, 2, 4, 8 wavefronts per SIMD on every CU (one
// workgroup per CU, pinned by a 96 KB LDS allocation).  Reported per kind and occupancy:
//   cyc/inst/SIMD  = shader cycles (s_memtime, median wave) x ... / (instructions per wave x waves per SIMD)
//                    i.e. how many cycles of ONE SIMD's issue capacity one wave-instruction occupies,
//   ns/inst/SIMD   = the same from HIP events (wall), which also gives the effective clock.
// Run under `rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` (tools/gpu_valu.sh) to get
// SQ_ACTIVE_INST_VALU per wave-instruction: the factor bench.py needs to turn that counter into issue cycles.
//
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue.hip -o tools/micro/bin/valu_issue
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Kind { K_FMA, K_MUL, K_PKFMA, K_ADD_DPP, K_MOV_DPP, K_EXP, K_RCP, K_CNDMASK, K_DSADD, K_DSREAD, K_MIX, K_EMPTY, K_CND_SGPR, K_CMP, K_CMPCND, K_BFI, K_MED3, K_DSADD_ROT, K_DSADD_U32, K_DSADD_RTN, K_DSADD_16L, K_DSRMW, K_COUNT };
static const char *kNames[K_COUNT] = {"v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_add_f32_dpp(row_shr:1)", "v_mov_b32_dpp(quad_perm)",
                                      "v_exp_f32", "v_rcp_f32", "v_cndmask_b32", "ds_add_f32", "ds_read_b128(bcast)",
                                      "mix(8fma+4dpp+2exp+2cnd)", "empty-loop", "v_cndmask_b32_e64(sgpr mask)",
                                      "v_cmp_ge_f32_e64->sgpr", "4x(cmp,fma,fma,cndmask)", "v_bfi_b32", "v_med3_f32", "ds_add_f32(16 rotating addresses)", "ds_add_u32(rotating)",
                                     "ds_add_rtn_u32(rotating)", "ds_add_f32(rotating, 16 of 64 lanes)", "ds_read_b32+v_add+ds_write_b32(rotating)"};
static const int kPerBlock[K_COUNT] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 0, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};

template <int KIND>
__global__ __launch_bounds__(1024) void issue_kernel(float *out, unsigned long long *cyc, int iters, float seed) {
    extern __shared__ float lds[];
    float v[16];
    float p[32];      // second half for the packed ops (register pairs)
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * (float)(i + threadIdx.x % 7);
#pragma unroll
    for (int i = 0; i < 32; ++i) p[i] = seed * 0.5f + 0.002f * (float)i;
    const float a = 0.9999f, b = 1e-6f;
    const unsigned lds_addr = (threadIdx.x * 4u) & 0xffffu;      // one dword per lane: conflict-free
    lds[threadIdx.x] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if constexpr (KIND == K_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                R16(X)
#undef X
            } else if constexpr (KIND == K_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                R16(X)
#undef X
            } else if constexpr (KIND == K_PKFMA) {
                typedef float f2 __attribute__((ext_vector_type(2)));
#define X(i) { f2 t = {p[2 * i], p[2 * i + 1]}; f2 aa = {a, a}, bb = {b, b}; \
               asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(aa), "v"(bb)); p[2 * i] = t.x; p[2 * i + 1] = t.y; }
                R16(X)
#undef X
            } else if constexpr (KIND == K_ADD_DPP) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]));
                R16(X)
#undef X
            } else if constexpr (KIND == K_MOV_DPP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
                R16(X)
#undef X
            } else if constexpr (KIND == K_EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                R16(X)
#undef X
            } else if constexpr (KIND == K_RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                R16(X)
#undef X
            } else if constexpr (KIND == K_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : );
                R16(X)
#undef X
            } else if constexpr (KIND == K_DSADD) {
#define X(i) asm volatile("ds_add_f32 %0, %1" : : "v"(lds_addr), "v"(v[i]) : "memory");
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (KIND == K_DSREAD) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 t0, t1, t2, t3;
                const unsigned ra = (threadIdx.x >> 4) * 528u;     // one address per 16-lane row (broadcast inside the row)
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*32" : "=v"(t##i) : "v"(ra) : "memory");
                X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                v[0] += t0.x + t1.y + t2.z + t3.w;
            } else if constexpr (KIND == K_CND_SGPR) {
                const unsigned long long m = 0x5555aaaa3333ccccull ^ (unsigned long long)iters;     // any SGPR-pair lane mask
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "s"(m));
                R16(X)
#undef X
            } else if constexpr (KIND == K_CMP) {
                unsigned long long m0, m1, m2, m3;
#define X(i) asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m##i) : "v"(v[i]), "v"(a));
                X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
                asm volatile("" :: "s"(m0), "s"(m1), "s"(m2), "s"(m3));
            } else if constexpr (KIND == K_CMPCND) {
                unsigned long long m0, m1, m2, m3;
#define X(i) asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m##i) : "v"(v[i]), "v"(a)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[4 + i]) : "v"(a), "v"(b)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[8 + i]) : "v"(a), "v"(b)); \
             asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(v[12 + i]) : "s"(m##i));
                X(0) X(1) X(2) X(3)
#undef X
            } else if constexpr (KIND == K_BFI) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                R16(X)
#undef X
            } else if constexpr (KIND == K_MED3) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                R16(X)
#undef X
            } else if constexpr (KIND == K_DSADD_ROT) {
                // round 3: the same-address stream above measures a dependent chain per lane; here consecutive instructions
                // hit 16 different dwords per lane (offset i * 4096), the pattern of an accumulator indexed by list position
#define X(i) asm volatile("ds_add_f32 %0, %1 offset:" #i "*4096" : : "v"(lds_addr & 0xfffu), "v"(v[i]) : "memory");
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (KIND == K_DSADD_U32) {
#define X(i) asm volatile("ds_add_u32 %0, %1 offset:" #i "*4096" : : "v"(lds_addr & 0xfffu), "v"(v[i]) : "memory");
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (KIND == K_DSADD_RTN) {
#define X(i) asm volatile("ds_add_rtn_u32 %0, %1, %0 offset:" #i "*4096" : "+v"(v[i]) : "v"(lds_addr & 0xfffu) : "memory");
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (KIND == K_DSADD_16L) {
                if ((threadIdx.x & 63u) < 16u) {
#define X(i) asm volatile("ds_add_f32 %0, %1 offset:" #i "*4096" : : "v"(lds_addr & 0xfffu), "v"(v[i]) : "memory");
                    R16(X)
#undef X
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (KIND == K_DSRMW) {
                float t0, t1, t2, t3;
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*4096" : "=v"(t##i) : "v"(lds_addr & 0xfffu) : "memory");
                X(0) X(1) X(2) X(3)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                t0 += v[0]; t1 += v[1]; t2 += v[2]; t3 += v[3];
#define X(i) asm volatile("ds_write_b32 %0, %1 offset:" #i "*4096" : : "v"(lds_addr & 0xfffu), "v"(t##i) : "memory");
                X(0) X(1) X(2) X(3)
#undef X
                // 4 read + 4 write per rep, counted as 16 per rep below: multiply the reported cost by 4 for one RMW of a dword... see note
            } else if constexpr (KIND == K_MIX) {
                // the proportions of the entry-per-lane compositing backward: 8 plain, 4 DPP, 2 transcendental, 2 selects
#define F(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define D(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]));
#define E(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define C(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a));
                F(0) F(1) D(8) F(2) F(3) E(12) D(9) F(4) C(14) F(5) D(10) F(6) E(13) F(7) D(11) C(15)
#undef F
#undef D
#undef E
#undef C
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += p[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
static void run(float *out, unsigned long long *cyc, int num_cu) {
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        const int per_cu = wps > 4 ? 2 : 1;                  // a workgroup holds at most 16 wavefronts
        const int threads = 64 * 4 * (wps / per_cu);
        const size_t lds = per_cu == 1 ? 96 * 1024 : 64 * 1024;      // LDS pins exactly per_cu workgroups on a CU (160 KB)
        hipFuncSetAttribute((const void *)issue_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(num_cu * per_cu), dim3(threads), lds, 0, out, cyc, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(num_cu * per_cu), dim3(threads), lds, 0, out, cyc, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const int waves = num_cu * 4 * wps;
        std::vector<unsigned long long> h(waves);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double n_inst = (double)iters * 4.0 * (kPerBlock[KIND] / 4.0);     // wave-instructions of the measured kind per wave
        const double med = (double)h[waves / 2];
        if (KIND == K_EMPTY)
            printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"median_wave_cycles\": %.0f}\n", kNames[KIND], wps, ms, med);
        else
            printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"median_wave_cycles\": %.0f, "
                   "\"memtime_ticks_per_inst_per_simd\": %.3f, \"ns_per_inst_per_simd\": %.4f, \"wave_insts_per_wave\": %.0f}\n",
                   kNames[KIND], wps, ms, med, med / (n_inst * wps), (double)ms * 1e6 / (n_inst * wps), n_inst);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int num_cu = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"note\": \"s_memtime ticks at a fixed rate; ns is wall\"}\n", prop.gcnArchName,
           num_cu, prop.clockRate);
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * (size_t)num_cu * 2048);
    hipMalloc(&cyc, sizeof(unsigned long long) * (size_t)num_cu * 32);
    const int only = argc > 1 ? atoi(argv[1]) : -1;
#define RUN(K) if (only < 0 || only == K) run<K>(out, cyc, num_cu);
    RUN(K_FMA) RUN(K_MUL) RUN(K_PKFMA) RUN(K_ADD_DPP) RUN(K_MOV_DPP) RUN(K_EXP) RUN(K_RCP) RUN(K_CNDMASK) RUN(K_DSADD) RUN(K_DSREAD)
    RUN(K_MIX) RUN(K_EMPTY) RUN(K_CND_SGPR) RUN(K_CMP) RUN(K_CMPCND) RUN(K_BFI) RUN(K_MED3)
    RUN(K_DSADD_ROT) RUN(K_DSADD_U32) RUN(K_DSADD_RTN) RUN(K_DSADD_16L) RUN(K_DSRMW)
    hipDeviceSynchronize();
    return 0;
}

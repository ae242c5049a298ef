// hostcheck.cpp -- TEST-ONLY: runs the product's per-element arithmetic headers (d3ga_math.h, raster_pre_body.h)
// on the CPU so that `pytest -m "not gpu"` can compare the formulas the gfx950 kernels execute against the
// oracle on a box without a GPU.  Never loaded by d3ga_amd/ (the product fails loudly without the HIP library).
#include <cstdint>
#include <cstring>

#include "../../d3ga_amd/csrc/raster_pre_body.h"

using namespace d3ga;

static DeformIn gather(int i, const float *tp, const int32_t *tetras, const int32_t *tid, const float *barys,
                       const float *cg, const float *scales, const float *rots, int vid[4]) {
    DeformIn in;
    const int t = tid[i];
    for (int k = 0; k < 4; ++k) vid[k] = tetras[4 * t + k];
    in.x0 = ld3(tp, vid[0]); in.x1 = ld3(tp, vid[1]); in.x2 = ld3(tp, vid[2]); in.x3 = ld3(tp, vid[3]);
    for (int k = 0; k < 4; ++k) in.bary[k] = barys[4 * (size_t)i + k];
    for (int k = 0; k < 9; ++k) in.G.m[k] = cg[9 * (size_t)i + k];
    for (int k = 0; k < 3; ++k) in.s[k] = scales[3 * (size_t)i + k];
    for (int k = 0; k < 4; ++k) in.q[k] = rots[4 * (size_t)i + k];
    return in;
}

extern "C" {

void hc_deform_fwd(int P, const float *tp, const int32_t *tetras, const int32_t *tid, const float *barys,
                   const float *cg, const float *scales, const float *rots, float *means, float *cov6) {
    for (int i = 0; i < P; ++i) {
        int vid[4];
        DeformIn in = gather(i, tp, tetras, tid, barys, cg, scales, rots, vid);
        deform_fwd(in, means + 3 * (size_t)i, cov6 + 6 * (size_t)i);
    }
}

void hc_deform_bwd(int P, int V, const float *tp, const int32_t *tetras, const int32_t *tid, const float *barys,
                   const float *cg, const float *scales, const float *rots, const float *gm, const float *gc,
                   float *g_tp, float *g_barys, float *g_scales, float *g_rots) {
    memset(g_tp, 0, sizeof(float) * 3 * (size_t)V);
    for (int i = 0; i < P; ++i) {
        int vid[4];
        DeformIn in = gather(i, tp, tetras, tid, barys, cg, scales, rots, vid);
        DeformGrad o;
        deform_bwd(in, gm + 3 * (size_t)i, gc + 6 * (size_t)i, o);
        for (int k = 0; k < 4; ++k) g_barys[4 * (size_t)i + k] = o.gbary[k];
        for (int k = 0; k < 3; ++k) g_scales[3 * (size_t)i + k] = o.gs[k];
        for (int k = 0; k < 4; ++k) g_rots[4 * (size_t)i + k] = o.gq[k];
        const V3 gx[4] = {o.gx0, o.gx1, o.gx2, o.gx3};
        for (int k = 0; k < 4; ++k) {
            g_tp[3 * vid[k]] += gx[k].x; g_tp[3 * vid[k] + 1] += gx[k].y; g_tp[3 * vid[k] + 2] += gx[k].z;
        }
    }
}

void hc_fem_fwd(int T, const float *tp, const int32_t *tetras, const float *Dn_inv, float *energy) {
    for (int t = 0; t < T; ++t) {
        M3 D;
        for (int k = 0; k < 9; ++k) D.m[k] = Dn_inv[9 * (size_t)t + k];
        energy[t] = fem_energy_fwd(ld3(tp, tetras[4 * t]), ld3(tp, tetras[4 * t + 1]), ld3(tp, tetras[4 * t + 2]),
                                   ld3(tp, tetras[4 * t + 3]), D);
    }
}

void hc_fem_bwd(int T, int V, const float *tp, const int32_t *tetras, const float *Dn_inv, const float *g,
                float *g_tp) {
    memset(g_tp, 0, sizeof(float) * 3 * (size_t)V);
    for (int t = 0; t < T; ++t) {
        M3 D;
        for (int k = 0; k < 9; ++k) D.m[k] = Dn_inv[9 * (size_t)t + k];
        V3 gx[4];
        fem_energy_bwd(ld3(tp, tetras[4 * t]), ld3(tp, tetras[4 * t + 1]), ld3(tp, tetras[4 * t + 2]),
                       ld3(tp, tetras[4 * t + 3]), D, g[t], gx);
        for (int k = 0; k < 4; ++k) {
            const int v = tetras[4 * t + k];
            g_tp[3 * v] += gx[k].x; g_tp[3 * v + 1] += gx[k].y; g_tp[3 * v + 2] += gx[k].z;
        }
    }
}

// preprocess forward: outputs depth (P), xy (P,2), conic_o (P,4), rgb (P,3), radii (P), rect (P,4), clamped (P), cov3D (P,6)
void hc_preprocess(const d3ga_raster_params *prm, const float *means3D, const float *shs, const float *colors,
                   const float *opacities, const float *scales, const float *rots, const float *cov3D_precomp,
                   const float *view, const float *proj, const float *campos, float *depth, float *xy, float *conic_o,
                   float *rgb, int32_t *radii, int32_t *rect, uint8_t *clamped, float *cov3D) {
    for (int i = 0; i < prm->P; ++i) {
        const PreOut o = preprocess_one(*prm, i, means3D, shs ? shs + (size_t)3 * prm->M * i : nullptr, colors, opacities,
                                        scales, rots, cov3D_precomp, view, proj, campos);
        depth[i] = o.sp.depth; xy[2 * i] = o.sp.px; xy[2 * i + 1] = o.sp.py;
        for (int k = 0; k < 3; ++k) conic_o[4 * i + k] = o.sp.conic[k];
        conic_o[4 * i + 3] = o.opacity;
        for (int k = 0; k < 3; ++k) rgb[3 * i + k] = o.rgb[k];
        radii[i] = o.sp.radius;
        for (int k = 0; k < 4; ++k) rect[4 * i + k] = o.sp.visible ? o.sp.rect[k] : 0;
        clamped[i] = o.clampmask;
        for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = o.c6[k];
    }
}

void hc_preprocess_bwd(const d3ga_raster_params *prm, const float *means3D, const float *shs, const float *scales,
                       const float *rots, const float *view, const float *proj, const float *campos,
                       const int32_t *radii, const float *cov3D, const uint8_t *clamped, const float *acc,
                       float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dopacity, float *dL_dsh, float *dL_dcolors,
                       float *dL_dcov3D, float *dL_dscales, float *dL_drots, const float *act_opacity, int use_dcol) {
    // use_dcol: the round-5 split -- d(colour)/d(direction) as the FORWARD forms it (sh_accumulate_jacobian), handed to the
    // backward in place of the coefficients (preprocess_bwd_one must then not read them: it is given a row of NaNs)
    const float zeros[12] = {0};
    for (int i = 0; i < prm->P; ++i) {
        const bool vis = radii[i] > 0;
        if (use_dcol && shs) {
            float B[16], x, y, z;
            sh_view_dir(means3D, i, campos, x, y, z);
            sh_basis(prm->sh_degree, x, y, z, B);
            ShColJ cj = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const int nb = (prm->sh_degree + 1) * (prm->sh_degree + 1);
            cj = sh_accumulate_jacobian(B, x, y, z, shs + (size_t)3 * prm->M * i, nb, cj);
            float acc3[3] = {0.f, 0.f, 0.f};
            sh_accumulate(B, shs + (size_t)3 * prm->M * i, 0, 16, nb, acc3);
            if (acc3[0] != cj.a0 || acc3[1] != cj.a1 || acc3[2] != cj.a2) { dL_dmeans3D[0] = NAN; return; }   // same colour, bit for bit
            float poison[48];
            for (int k = 0; k < 48; ++k) poison[k] = NAN;
            preprocess_bwd_one(*prm, i, vis, means3D, poison, scales, rots, view, proj, campos, cov3D + 6 * (size_t)i, clamped[i],
                               vis ? acc + D3GA_ACC_STRIDE * (size_t)i : zeros, dL_dmeans3D, dL_dmeans2D, dL_dopacity,
                               dL_dsh ? dL_dsh + (size_t)3 * prm->M * i : nullptr, dL_dcolors, dL_dcov3D, dL_dscales, dL_drots,
                               act_opacity ? act_opacity[i] : 0.f, true, cj);
            continue;
        }
        preprocess_bwd_one(*prm, i, vis, means3D, shs ? shs + (size_t)3 * prm->M * i : nullptr, scales, rots, view, proj,
                           campos, cov3D + 6 * (size_t)i, clamped[i], vis ? acc + D3GA_ACC_STRIDE * (size_t)i : zeros, dL_dmeans3D,
                           dL_dmeans2D, dL_dopacity, dL_dsh ? dL_dsh + (size_t)3 * prm->M * i : nullptr, dL_dcolors,
                           dL_dcov3D, dL_dscales, dL_drots, act_opacity ? act_opacity[i] : 0.f);
    }
}

}  // extern "C"

// raster_pre_body.h -- per-Gaussian bodies of the preprocess forward / backward kernels, written as
// host+device functions over plain pointers so that tests/hostcheck can run exactly this code on the CPU.
#pragma once
#include "../../include/d3ga.h"
#include "d3ga_math.h"

namespace d3ga {

D3GA_HD V3 ld3(const float *p, size_t i) { return v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }

struct PreOut {
    Splat sp;
    float c6[6];
    float rgb[3];
    float opacity;
    uint8_t clampmask;
};

// The unit view direction of Gaussian i: normalize(mean - campos), as in R1.  Without contraction, like sh_basis: the
// inference forward and the forward that also leaves d(colour)/d(direction) must evaluate the same basis values.
D3GA_HD void sh_view_dir(const float *means3D, int i, const float *campos, float &x, float &y, float &z) {
    D3GA_NO_CONTRACT
    const V3 m = ld3(means3D, i);
    const float dx = m.x - campos[0], dy = m.y - campos[1], dz = m.z - campos[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    x = dx * inv; y = dy * inv; z = dz * inv;
}
// SH basis of Gaussian i's view direction
D3GA_HD void sh_view_basis(const d3ga_raster_params &prm, const float *means3D, int i, const float *campos, float B[16]) {
    float x, y, z;
    sh_view_dir(means3D, i, campos, x, y, z);
    sh_basis(prm.sh_degree, x, y, z, B);
}
// acc[c] += sum_k B[k] * coeff[k][c]  AND  J[3 dir + c] = sum_k dY_k/d(dir)(x, y, z) * coeff[k][c] -- the derivative of the
// (unclamped, un-offset) SH colour w.r.t. the unit direction -- in ONE walk over the row: every coefficient is read once and
// the three derivative values of a basis function are formed where they are used (no 3 x 16 gradient arrays: the forward's
// staging kernel has no registers for them).  Same polynomials as sh_basis_grad.
struct ShColJ { float a0, a1, a2, j0, j1, j2, j3, j4, j5, j6, j7, j8; };      // by value: as arrays behind pointers these landed in scratch memory
D3GA_HD ShColJ sh_accumulate_jacobian(const float B[16], float x, float y, float z, const float *row, int nb, ShColJ o) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    o.j0 = o.j1 = o.j2 = o.j3 = o.j4 = o.j5 = o.j6 = o.j7 = o.j8 = 0.f;
#define D3GA_SHJ(K, GX, GY, GZ)                                                                             \
    if (K < nb) {                                                                                           \
        const float r0 = row[3 * K], r1 = row[3 * K + 1], r2 = row[3 * K + 2];                              \
        o.a0 = fmaf(B[K], r0, o.a0); o.a1 = fmaf(B[K], r1, o.a1); o.a2 = fmaf(B[K], r2, o.a2);   /* as sh_accumulate, bit for bit */                                            \
        const float gx_ = (GX), gy_ = (GY), gz_ = (GZ);                                                     \
        o.j0 += gx_ * r0; o.j1 += gx_ * r1; o.j2 += gx_ * r2;                                               \
        o.j3 += gy_ * r0; o.j4 += gy_ * r1; o.j5 += gy_ * r2;                                               \
        o.j6 += gz_ * r0; o.j7 += gz_ * r1; o.j8 += gz_ * r2;                                               \
    }
    if (0 < nb) { o.a0 = fmaf(B[0], row[0], o.a0); o.a1 = fmaf(B[0], row[1], o.a1); o.a2 = fmaf(B[0], row[2], o.a2); }
    D3GA_SHJ(1, 0.f, -kC1, 0.f)
    D3GA_SHJ(2, 0.f, 0.f, kC1)
    D3GA_SHJ(3, -kC1, 0.f, 0.f)
    D3GA_SHJ(4, kC2_0 * y, kC2_0 * x, 0.f)
    D3GA_SHJ(5, 0.f, kC2_1 * z, kC2_1 * y)
    D3GA_SHJ(6, -2.f * kC2_2 * x, -2.f * kC2_2 * y, 4.f * kC2_2 * z)
    D3GA_SHJ(7, kC2_3 * z, 0.f, kC2_3 * x)
    D3GA_SHJ(8, 2.f * kC2_4 * x, -2.f * kC2_4 * y, 0.f)
    D3GA_SHJ(9, kC3_0 * 6.f * xy, kC3_0 * 3.f * (xx - yy), 0.f)
    D3GA_SHJ(10, kC3_1 * yz, kC3_1 * xz, kC3_1 * xy)
    D3GA_SHJ(11, kC3_2 * -2.f * xy, kC3_2 * (4.f * zz - xx - 3.f * yy), kC3_2 * 8.f * yz)
    D3GA_SHJ(12, kC3_3 * -6.f * xz, kC3_3 * -6.f * yz, kC3_3 * 3.f * (2.f * zz - xx - yy))
    D3GA_SHJ(13, kC3_4 * (4.f * zz - 3.f * xx - yy), kC3_4 * -2.f * xy, kC3_4 * 8.f * xz)
    D3GA_SHJ(14, kC3_5 * 2.f * xz, kC3_5 * -2.f * yz, kC3_5 * (xx - yy))
    D3GA_SHJ(15, kC3_6 * 3.f * (xx - yy), kC3_6 * -6.f * xy, 0.f)
#undef D3GA_SHJ
    return o;
}
// acc[c] += sum_{k in [k0, k1)} B[k] * coeff[k][c];  `part` points at coefficient k0 of the row (3 floats per coefficient)
D3GA_HD void sh_accumulate(const float B[16], const float *part, int k0, int k1, int nb, float acc[3]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {           // fixed trip count: keeps B[] in registers
        if (k >= k0 && k < k1 && k < nb) {
            const float *c = part + 3 * (k - k0);
            acc[0] = fmaf(B[k], c[0], acc[0]); acc[1] = fmaf(B[k], c[1], acc[1]); acc[2] = fmaf(B[k], c[2], acc[2]);   // explicit: the same in every caller
        }
    }
}

// R1 for Gaussian i.  Exactly one of (sh_row|sh_acc|colors_precomp), ((scales,rotations)|cov3D_precomp) is non-null.
// sh_row points at THIS Gaussian's 3*M SH floats (in global memory or in an LDS staging row); alternatively pl.sh
// holds the already evaluated sum_k Y_k(dir) * coeff_k (see sh_view_basis / sh_accumulate).
// pl.c6 / pl.op: this Gaussian's covariance row and raw opacity if the caller has loaded them already (the kernel issues
// those loads before it stages the SH rows).  The opacity is read unconditionally: behind the visibility test it would be
// one more dependent memory round trip per wavefront.
// What the caller has in registers already, BY VALUE (round 4: as nullable pointers to locals these lived in scratch
// memory -- 69 scratch instructions in the kernel and a dependent memory round trip in front of the colour).
struct PreLoaded {
    bool has_sh = false, has_c6 = false;   // uniform over the launch
    float sh[3] = {0.f, 0.f, 0.f};         // sum_k Y_k(dir) * coeff_k, already evaluated (sh_view_basis / sh_accumulate)
    float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float op = 0.f;                        // raw opacity (with has_c6)
};
D3GA_HD PreOut preprocess_one(const d3ga_raster_params &prm, int i, const float *means3D, const float *sh_row,
                              const float *colors_precomp, const float *opacities, const float *scales,
                              const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                              const float *projmatrix, const float *campos, const PreLoaded pl = PreLoaded()) {
    PreOut o;
    const V3 mean = ld3(means3D, i);
    const float raw_opacity = pl.has_c6 ? pl.op : opacities[i];
    if (pl.has_c6) {
        for (int k = 0; k < 6; ++k) o.c6[k] = pl.c6[k];
    } else if (cov3D_precomp) {
        for (int k = 0; k < 6; ++k) o.c6[k] = cov3D_precomp[6 * (size_t)i + k];
    } else {
        const float s[3] = {scales[3 * (size_t)i], scales[3 * (size_t)i + 1], scales[3 * (size_t)i + 2]};
        const float q[4] = {rotations[4 * (size_t)i], rotations[4 * (size_t)i + 1], rotations[4 * (size_t)i + 2],
                            rotations[4 * (size_t)i + 3]};
        cov3d_from_scale_rot(s, prm.scale_modifier, q, o.c6);
    }
    o.sp = project_gaussian(mean, o.c6, viewmatrix, projmatrix, prm.W, prm.H, prm.tanfovx, prm.tanfovy, prm.antialiasing != 0);
    o.rgb[0] = o.rgb[1] = o.rgb[2] = 0.f;
    o.clampmask = 0;
    o.opacity = 0.f;
    if (!o.sp.visible) return o;
    o.opacity = raw_opacity;
    if (prm.opacity_activation == D3GA_OPACITY_SIGMOID) o.opacity = 1.0f / (1.0f + expf(-o.opacity));   // cage_net.py:247
    if (!(o.opacity == o.opacity)) {         // NaN opacity: min(0.99, NaN * G) would evaluate to 0.99 -- cull instead
        o.sp.visible = false; o.sp.radius = 0;
        o.sp.rect[0] = o.sp.rect[1] = o.sp.rect[2] = o.sp.rect[3] = 0;
        o.opacity = 0.f;
        return o;
    }
    o.opacity *= o.sp.aa;                    // antialiasing (branch dr_aa): what the compositing stage sees is opacity x h_convolution_scaling
    if (colors_precomp) {
        o.rgb[0] = colors_precomp[3 * (size_t)i]; o.rgb[1] = colors_precomp[3 * (size_t)i + 1];
        o.rgb[2] = colors_precomp[3 * (size_t)i + 2];
    } else {
        float acc[3] = {0.f, 0.f, 0.f};
        if (pl.has_sh) {
            acc[0] = pl.sh[0]; acc[1] = pl.sh[1]; acc[2] = pl.sh[2];
        } else {
            float B[16];
            sh_view_basis(prm, means3D, i, campos, B);
            sh_accumulate(B, sh_row, 0, 16, (prm.sh_degree + 1) * (prm.sh_degree + 1), acc);
        }
        for (int c = 0; c < 3; ++c) {
            const float v = acc[c] + 0.5f;
            if (v < 0.f) o.clampmask |= (uint8_t)(1u << c);
            o.rgb[c] = fmaxf(v, 0.f);
        }
    }
    return o;
}

// R6 for Gaussian i.  a[12] = accumulated screen-space gradients (layout: d3ga.h, d3ga_raster_composite_bwd);
// all-zero and visible=false for culled Gaussians.  Output pointers may be null where not applicable.
// sh_row / dsh_row point at THIS Gaussian's 3*M floats (global memory or an LDS staging row; they may alias).
D3GA_HD void preprocess_bwd_one(const d3ga_raster_params &prm, int i, bool visible, const float *means3D,
                                const float *sh_row, const float *scales, const float *rotations,
                                const float *viewmatrix, const float *projmatrix, const float *campos,
                                const float *c6, uint8_t clampmask, const float *a, float *dL_dmeans3D,
                                float *dL_dmeans2D, float *dL_dopacity, float *dsh_row, float *dL_dcolors,
                                float *dL_dcov3D, float *dL_dscales, float *dL_drots, float act_opacity = 0.f,
                                bool have_j = false, ShColJ jd = ShColJ(),      // have_j: jd.j0..j8 = the forward's d(colour)/d(direction) of this Gaussian (then sh_row is not read)
                                int accum = 0) {      // views > 0 of a batch (d3ga.h n_views): gradients of inputs the views SHARE are added to what the outputs hold -- bit 0: opacity and a precomputed colour, bit 1: the geometry (mean, covariance | scale, rotation)
    float gmean[3] = {0.f, 0.f, 0.f};
    float g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const V3 mean = ld3(means3D, i);
    const int nbM = prm.M;
    float aa = 1.0f;
    if (visible) {
        cov2d_bwd(mean, c6, viewmatrix, prm.W, prm.H, prm.tanfovx, prm.tanfovy, a[3], a[4], a[5], g6, gmean,
                  prm.antialiasing != 0, a[6], act_opacity, &aa);
        project_bwd(mean, projmatrix, a[0], a[1], gmean);
        // inverse depth (branch dr_aa): a[10] = dL/d(1/z) of this Gaussian, z its view-space depth; d(1/z)/dmean = -view_z / z^2
        const float z = viewmatrix[2] * mean.x + viewmatrix[6] * mean.y + viewmatrix[10] * mean.z + viewmatrix[14];
        const float gz = -a[10] / (z * z);
        gmean[0] += viewmatrix[2] * gz; gmean[1] += viewmatrix[6] * gz; gmean[2] += viewmatrix[10] * gz;
    }
    // SH colour path (sh_row != null).  dsh_row == null selects the FACTORED output used by the view-sharded gradient
    // exchange (d3ga_sh_grad_from_views): the clamp-masked dL/dcolour is written to dL_dcolors instead of the rank-1
    // (basis x dL/dcolour) SH block; the view-direction term of dL/dmean is computed either way.
    if (sh_row) {
        float *out = dsh_row;
        if (visible) {
            const float gr[3] = {(clampmask & 1) ? 0.f : a[7], (clampmask & 2) ? 0.f : a[8],
                                 (clampmask & 4) ? 0.f : a[9]};
            const V3 d0 = mean - v3(campos[0], campos[1], campos[2]);
            const float inv = 1.0f / sqrtf(dot(d0, d0));
            const float x = d0.x * inv, y = d0.y * inv, z = d0.z * inv;
            float B[16];
            sh_basis(prm.sh_degree, x, y, z, B);
            const int nb = (prm.sh_degree + 1) * (prm.sh_degree + 1);
            V3 gd = v3(0.f, 0.f, 0.f);
            if (have_j) {                    // dL/d(dir) = J . (clamp-masked dL/dcolour): the coefficients are not read
                gd = v3(jd.j0 * gr[0] + jd.j1 * gr[1] + jd.j2 * gr[2], jd.j3 * gr[0] + jd.j4 * gr[1] + jd.j5 * gr[2],
                        jd.j6 * gr[0] + jd.j7 * gr[1] + jd.j8 * gr[2]);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k < nb) { if (out) { out[3 * k] = B[k] * gr[0]; out[3 * k + 1] = B[k] * gr[1]; out[3 * k + 2] = B[k] * gr[2]; } }
                    else if (k < nbM && out) { out[3 * k] = 0.f; out[3 * k + 1] = 0.f; out[3 * k + 2] = 0.f; }
                }
            } else {
                float Bx[16], By[16], Bz[16];
                sh_basis_grad(prm.sh_degree, x, y, z, Bx, By, Bz);
                const float *sh = sh_row;
#pragma unroll
                for (int k = 0; k < 16; ++k) {   // fixed trip count: keeps the basis arrays in registers
                    if (k < nb) {
                        const float s0 = sh[3 * k], s1 = sh[3 * k + 1], s2 = sh[3 * k + 2];
                        if (out) { out[3 * k] = B[k] * gr[0]; out[3 * k + 1] = B[k] * gr[1]; out[3 * k + 2] = B[k] * gr[2]; }
                        const float w = s0 * gr[0] + s1 * gr[1] + s2 * gr[2];
                        gd.x += Bx[k] * w; gd.y += By[k] * w; gd.z += Bz[k] * w;
                    } else if (k < nbM && out) {
                        out[3 * k] = 0.f; out[3 * k + 1] = 0.f; out[3 * k + 2] = 0.f;
                    }
                }
            }
            const V3 gm = normalize_bwd(d0, gd);
            gmean[0] += gm.x; gmean[1] += gm.y; gmean[2] += gm.z;
            if (dL_dcolors) {
                dL_dcolors[3 * (size_t)i] = gr[0]; dL_dcolors[3 * (size_t)i + 1] = gr[1]; dL_dcolors[3 * (size_t)i + 2] = gr[2];
            }
        } else {
            if (out) for (int k = 0; k < 3 * nbM; ++k) out[k] = 0.f;
            if (dL_dcolors) {
                dL_dcolors[3 * (size_t)i] = 0.f; dL_dcolors[3 * (size_t)i + 1] = 0.f; dL_dcolors[3 * (size_t)i + 2] = 0.f;
            }
        }
    } else if (dL_dcolors) {                 // a precomputed colour is view-independent: summed over a batch's views
        float *o = dL_dcolors + 3 * (size_t)i;
        o[0] = (accum & 1) ? o[0] + (a[7]) : (a[7]); o[1] = (accum & 1) ? o[1] + (a[8]) : (a[8]); o[2] = (accum & 1) ? o[2] + (a[9]) : (a[9]);
    }
    {
        float *o = dL_dmeans3D + 3 * (size_t)i;
        o[0] = (accum & 2) ? o[0] + (gmean[0]) : (gmean[0]); o[1] = (accum & 2) ? o[1] + (gmean[1]) : (gmean[1]); o[2] = (accum & 2) ? o[2] + (gmean[2]) : (gmean[2]);
    }
    if (dL_dmeans2D) {                       // screen-space: per view
        dL_dmeans2D[3 * (size_t)i] = a[0]; dL_dmeans2D[3 * (size_t)i + 1] = a[1]; dL_dmeans2D[3 * (size_t)i + 2] = 0.f;
    }
    // act_opacity: the activated opacity the forward stored (conic_o.w); sigmoid' = s (1 - s)
    // (antialiasing: the stored opacity is opacity x aa; a[6] is the gradient w.r.t. that product)
    if (dL_dopacity) {
        const float op = act_opacity / aa, g_op = a[6] * aa;
        const float g = prm.opacity_activation == D3GA_OPACITY_SIGMOID ? g_op * op * (1.0f - op) : g_op;
        dL_dopacity[i] = (accum & 1) ? dL_dopacity[i] + g : g;
    }
    if (dL_dcov3D) {
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * (size_t)i + k] = (accum & 2) ? dL_dcov3D[6 * (size_t)i + k] + (g6[k]) : (g6[k]);
    }
    if (dL_dscales && dL_drots) {
        float gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
        if (visible) {
            const float s[3] = {scales[3 * (size_t)i], scales[3 * (size_t)i + 1], scales[3 * (size_t)i + 2]};
            const float q[4] = {rotations[4 * (size_t)i], rotations[4 * (size_t)i + 1], rotations[4 * (size_t)i + 2],
                                rotations[4 * (size_t)i + 3]};
            cov3d_from_scale_rot_bwd(s, prm.scale_modifier, q, g6, gs, gq);
        }
        for (int k = 0; k < 3; ++k) dL_dscales[3 * (size_t)i + k] = (accum & 2) ? dL_dscales[3 * (size_t)i + k] + (gs[k]) : (gs[k]);
        for (int k = 0; k < 4; ++k) dL_drots[4 * (size_t)i + k] = (accum & 2) ? dL_drots[4 * (size_t)i + k] + (gq[k]) : (gq[k]);
    }
}

}  // namespace d3ga

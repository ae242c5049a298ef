/*
 * Oracle: CPU tile rasterizer for 3D Gaussians, forward + hand-derived backward (plain C, float32).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product (d3ga_amd/) never links or calls it.
 *
 * *** PARITY UNPINNED *** The reference (facebookresearch/D3GA) calls the un-vendored CUDA package
 * diff_gaussian_rasterization (graphdeco-inria, branch dr_aa, SHA not recorded; /root/reference/.gitmodules:9-12)
 * at renderer.py:79-141 and carries no test/golden for it.  This file restates the published 3DGS algorithm
 * (Kerbl et al. 2023, sec. 4-6, App. A-C) with the constants of SURVEY.md sec. 8a R1-R6, keeping the
 * float32 evaluation order a tile splatter naturally has (per Gaussian -> per tile list -> per pixel).
 * It is validated against oracle/raster_torch.py (independent dense autograd restatement) in tests/.
 *
 * Matrix convention: viewmatrix / projmatrix are the reference's transposed ("row-vector") 4x4 matrices
 * (lib/cameras.py:68-74), flattened row-major: element (r,c) of the stored matrix is m[4*r+c], and a point
 * transforms as p' = [p,1] * M, i.e. p'.x = m[0]x + m[4]y + m[8]z + m[12].
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC (oracle/build_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

typedef struct {
    int P, M, deg, W, H, gx, gy;
    int from_scale_rot, has_sh;
    float scale_modifier, tanfovx, tanfovy;
    float view[16], proj[16], campos[3], bg[3];
    /* per Gaussian */
    float *depth, *xy, *conic_o, *rgb, *cov3D;
    int *radii;
    uint8_t *clamped;
    int *rect; /* 4 per Gaussian */
    /* per tile */
    int64_t D;
    int64_t *tile_start; /* tiles+1 */
    int *point_list;     /* D */
    /* per pixel */
    float *final_T;
    int *n_contrib;
    /* decision margins (see ro_marginal): per pixel the smallest relative distance of any evaluated alpha to the
     * 1/255 cut-off and of any test_T to the 1e-4 cut-off */
    float *margin_alpha, *margin_T;
    int antialiasing;    /* ro_set_antialiasing at ro_forward time */
    float *aa;           /* P: the antialiasing factor h (1 when off) */
    float *op_raw;       /* P: the opacity as given (conic_o[3] = op_raw x h) */
    /* shared alpha decisions (ro_set_alpha_overrides): n pairs sorted by (pixel, Gaussian), 1 = the pair is blended */
    int64_t ovr_n;
    int *ovr_pix, *ovr_gid;
    uint8_t *ovr_ok, *ovr_has; /* ovr_has: H*W, 1 where a pixel has an override */
} ro_ctx;

static void xform4x3(const float *m, const float *p, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform4x4(const float *m, const float *p, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* Sigma = R diag(mod*s)^2 R^T, quaternion (w,x,y,z) used as given (not normalised) */
static void cov3d_from_scale_rot(const float *s, float mod, const float *q, float *c6) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                  2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                  2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
    float sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
    float L[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) L[3 * i + j] = R[3 * i + j] * sc[j];
    float S[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float a = 0.f;
            for (int k = 0; k < 3; k++) a += L[3 * i + k] * L[3 * j + k];
            S[3 * i + j] = a;
        }
    c6[0] = S[0]; c6[1] = S[1]; c6[2] = S[2]; c6[3] = S[4]; c6[4] = S[5]; c6[5] = S[8];
}

/* T = Jac * Wrot (2x3, third row zero); returns t (possibly clamped) */
static void ewa_T(const ro_ctx *c, const float *mean, float *t, float T[6], int *clx, int *cly) {
    xform4x3(c->view, mean, t);
    float fx = c->W / (2.f * c->tanfovx), fy = c->H / (2.f * c->tanfovy);
    float limx = 1.3f * c->tanfovx, limy = 1.3f * c->tanfovy;
    float txtz = t[0] / t[2], tytz = t[1] / t[2];
    *clx = (txtz < -limx || txtz > limx);
    *cly = (tytz < -limy || tytz > limy);
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    float J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
    float J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
    const float *v = c->view; /* world->view rotation Wr[r][c] = v[4*c + r] */
    for (int k = 0; k < 3; k++) {
        float w0 = v[4 * k + 0], w1 = v[4 * k + 1], w2 = v[4 * k + 2];
        T[k] = J00 * w0 + J02 * w2;
        T[3 + k] = J11 * w1 + J12 * w2;
    }
}

static void eval_sh(const ro_ctx *c, int idx, const float *mean, const float *sh, float *rgb, uint8_t *clamped) {
    float d[3] = {mean[0] - c->campos[0], mean[1] - c->campos[1], mean[2] - c->campos[2]};
    float inv = 1.f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
    for (int ch = 0; ch < 3; ch++) {
#define SHC(k) sh[3 * (k) + ch]
        float r = SH_C0 * SHC(0);
        if (c->deg > 0) {
            r = r - SH_C1 * y * SHC(1) + SH_C1 * z * SHC(2) - SH_C1 * x * SHC(3);
            if (c->deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * SHC(4) + SH_C2[1] * yz * SHC(5) + SH_C2[2] * (2.f * zz - xx - yy) * SHC(6) +
                    SH_C2[3] * xz * SHC(7) + SH_C2[4] * (xx - yy) * SHC(8);
                if (c->deg > 2) {
                    r = r + SH_C3[0] * y * (3.f * xx - yy) * SHC(9) + SH_C3[1] * xy * z * SHC(10) +
                        SH_C3[2] * y * (4.f * zz - xx - yy) * SHC(11) +
                        SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * SHC(12) +
                        SH_C3[4] * x * (4.f * zz - xx - yy) * SHC(13) + SH_C3[5] * z * (xx - yy) * SHC(14) +
                        SH_C3[6] * x * (xx - 3.f * yy) * SHC(15);
                }
            }
        }
#undef SHC
        r += 0.5f;
        clamped[3 * idx + ch] = (r < 0.f);
        rgb[ch] = fmaxf(r, 0.f);
    }
}

typedef struct { uint64_t key; int idx; } kv_t;
static int kv_cmp(const void *a, const void *b) {
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

#ifdef _OPENMP
#include <omp.h>
void ro_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int ro_max_threads(void) { return omp_get_max_threads(); }
#else
void ro_set_threads(int n) { (void)n; }
int ro_max_threads(void) { return 1; }
#endif

/* Branch dr_aa [UPSTREAM-RECALL].  ro_set_antialiasing(1) before ro_forward: the opacity the compositing stage sees is
 * opacity x h, h = sqrt(max(0.000025, det(cov2D) / det(cov2D + 0.3 I))) (kept per Gaussian in the context; ro_backward chains it).
 * ro_set_invdepth_grad(g | NULL) before ro_backward: g (H,W) = dL/d(out_invdepth), a fourth channel whose per-Gaussian value is
 * 1 / depth. */
static int g_antialiasing = 0;
static const float *g_dL_dinvdepth = NULL;
void ro_set_antialiasing(int on) { g_antialiasing = on != 0; }
void ro_set_invdepth_grad(const float *g) { g_dL_dinvdepth = g; }

void ro_free(ro_ctx *c) {
    if (!c) return;
    free(c->depth); free(c->xy); free(c->conic_o); free(c->rgb); free(c->cov3D); free(c->radii);
    free(c->clamped); free(c->rect); free(c->tile_start); free(c->point_list); free(c->final_T); free(c->n_contrib);
    free(c->margin_alpha); free(c->margin_T); free(c->aa); free(c->op_raw);
    free(c->ovr_pix); free(c->ovr_gid); free(c->ovr_ok); free(c->ovr_has);
    free(c);
}

int64_t ro_num_rendered(const ro_ctx *c) { return c->D; }
void ro_get_tile_list(const ro_ctx *c, int64_t *tile_start, int *point_list) {
    memcpy(tile_start, c->tile_start, sizeof(int64_t) * ((size_t)c->gx * c->gy + 1));
    memcpy(point_list, c->point_list, sizeof(int) * (size_t)c->D);
}
void ro_get_geom(const ro_ctx *c, float *depth, float *xy, float *conic_o, float *rgb, int *n_contrib, float *final_T) {
    memcpy(depth, c->depth, sizeof(float) * c->P);
    memcpy(xy, c->xy, sizeof(float) * 2 * c->P);
    memcpy(conic_o, c->conic_o, sizeof(float) * 4 * c->P);
    memcpy(rgb, c->rgb, sizeof(float) * 3 * c->P);
    memcpy(n_contrib, c->n_contrib, sizeof(int) * (size_t)c->W * c->H);
    memcpy(final_T, c->final_T, sizeof(float) * (size_t)c->W * c->H);
}

/* Forward.  Any of shs/colors, scales+rots/cov3D_precomp may be NULL (exactly one of each pair given).
 * Outputs: out_color (3,H,W), radii (P), out_invdepth (H,W).  Returns a context for ro_backward. */
ro_ctx *ro_forward(int P, int M, int deg, int W, int H, const float *means3D, const float *shs, const float *colors,
                   const float *opacities, const float *scales, const float *rots, float scale_modifier,
                   const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *campos,
                   float tanfovx, float tanfovy, const float *bg, float *out_color, int *radii_out,
                   float *out_invdepth) {
    ro_ctx *c = (ro_ctx *)calloc(1, sizeof(ro_ctx));
    c->P = P; c->M = M; c->deg = deg; c->W = W; c->H = H;
    c->gx = (W + TILE - 1) / TILE; c->gy = (H + TILE - 1) / TILE;
    c->from_scale_rot = (cov3D_precomp == NULL); c->has_sh = (colors == NULL);
    c->scale_modifier = scale_modifier; c->tanfovx = tanfovx; c->tanfovy = tanfovy;
    memcpy(c->view, viewmatrix, 64); memcpy(c->proj, projmatrix, 64);
    memcpy(c->campos, campos, 12); memcpy(c->bg, bg, 12);
    size_t Pn = P > 0 ? P : 1;
    c->depth = (float *)calloc(Pn, 4); c->xy = (float *)calloc(Pn, 8); c->conic_o = (float *)calloc(Pn, 16);
    c->rgb = (float *)calloc(Pn, 12); c->cov3D = (float *)calloc(Pn, 24); c->radii = (int *)calloc(Pn, 4);
    c->clamped = (uint8_t *)calloc(Pn, 3); c->rect = (int *)calloc(Pn, 16);
    c->antialiasing = g_antialiasing; c->aa = (float *)calloc(Pn, 4); c->op_raw = (float *)calloc(Pn, 4);
    int tiles = c->gx * c->gy;
    int64_t *counts = (int64_t *)calloc((size_t)tiles + 1, 8);

#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        const float *m = means3D + 3 * i;
        float pv[3];
        xform4x3(c->view, m, pv);
        if (pv[2] <= 0.2f) continue;
        float ph[4];
        xform4x4(c->proj, m, ph);
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float px = ph[0] * pw, py = ph[1] * pw;
        float *c6 = c->cov3D + 6 * i;
        if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * i, 24);
        else cov3d_from_scale_rot(scales + 3 * i, scale_modifier, rots + 4 * i, c6);
        float t[3], T[6]; int clx, cly;
        ewa_T(c, m, t, T, &clx, &cly);
        /* cov2D = T Sigma T^T */
        float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
        float TS[6];
        for (int r = 0; r < 2; r++)
            for (int k = 0; k < 3; k++) TS[3 * r + k] = T[3 * r] * S[k] + T[3 * r + 1] * S[3 + k] + T[3 * r + 2] * S[6 + k];
        float a = TS[0] * T[0] + TS[1] * T[1] + TS[2] * T[2];
        float b = TS[0] * T[3] + TS[1] * T[4] + TS[2] * T[5];
        float cc = TS[3] * T[3] + TS[4] * T[4] + TS[5] * T[5];
        float det0 = a * cc - b * b;
        a += 0.3f; cc += 0.3f;
        float det = a * cc - b * b;
        if (det == 0.0f) continue;
        float h_aa = c->antialiasing ? sqrtf(fmaxf(0.000025f, det0 / det)) : 1.0f;
        float det_inv = 1.f / det;
        float conic[3] = {cc * det_inv, -b * det_inv, a * det_inv};
        float mid = 0.5f * (a + cc);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float rad = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix[2] = {((px + 1.0f) * W - 1.0f) * 0.5f, ((py + 1.0f) * H - 1.0f) * 0.5f};
        int rminx = (int)((pix[0] - rad) / TILE), rminy = (int)((pix[1] - rad) / TILE);
        int rmaxx = (int)((pix[0] + rad + TILE - 1) / TILE), rmaxy = (int)((pix[1] + rad + TILE - 1) / TILE);
        rminx = rminx < 0 ? 0 : (rminx > c->gx ? c->gx : rminx);
        rminy = rminy < 0 ? 0 : (rminy > c->gy ? c->gy : rminy);
        rmaxx = rmaxx < 0 ? 0 : (rmaxx > c->gx ? c->gx : rmaxx);
        rmaxy = rmaxy < 0 ? 0 : (rmaxy > c->gy ? c->gy : rmaxy);
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;
        if (c->has_sh) eval_sh(c, i, m, shs + (size_t)3 * M * i, c->rgb + 3 * i, c->clamped);
        else memcpy(c->rgb + 3 * i, colors + 3 * i, 12);
        c->depth[i] = pv[2];
        c->radii[i] = (int)rad;
        c->xy[2 * i] = pix[0]; c->xy[2 * i + 1] = pix[1];
        c->conic_o[4 * i] = conic[0]; c->conic_o[4 * i + 1] = conic[1]; c->conic_o[4 * i + 2] = conic[2];
        c->conic_o[4 * i + 3] = opacities[i] * h_aa;
        c->aa[i] = h_aa; c->op_raw[i] = opacities[i];
        c->rect[4 * i] = rminx; c->rect[4 * i + 1] = rminy; c->rect[4 * i + 2] = rmaxx; c->rect[4 * i + 3] = rmaxy;
    }
    memcpy(radii_out, c->radii, sizeof(int) * P);

    /* tile lists ordered by (depth, index) */
    for (int i = 0; i < P; i++) {
        if (c->radii[i] <= 0) continue;
        const int *r = c->rect + 4 * i;
        for (int y = r[1]; y < r[3]; y++)
            for (int x = r[0]; x < r[2]; x++) counts[y * c->gx + x + 1]++;
    }
    for (int t = 0; t < tiles; t++) counts[t + 1] += counts[t];
    c->tile_start = counts;
    c->D = counts[tiles];
    kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (size_t)(c->D > 0 ? c->D : 1));
    int64_t *cursor = (int64_t *)malloc(8 * ((size_t)tiles + 1));
    memcpy(cursor, counts, 8 * ((size_t)tiles + 1));
    for (int i = 0; i < P; i++) {
        if (c->radii[i] <= 0) continue;
        const int *r = c->rect + 4 * i;
        uint32_t db; memcpy(&db, &c->depth[i], 4);
        for (int y = r[1]; y < r[3]; y++)
            for (int x = r[0]; x < r[2]; x++) {
                int64_t p = cursor[y * c->gx + x]++;
                kv[p].key = db; kv[p].idx = i;
            }
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (int t = 0; t < tiles; t++) {
        int64_t n = counts[t + 1] - counts[t];
        if (n > 1) qsort(kv + counts[t], (size_t)n, sizeof(kv_t), kv_cmp);
    }
    c->point_list = (int *)malloc(4 * (size_t)(c->D > 0 ? c->D : 1));
    for (int64_t k = 0; k < c->D; k++) c->point_list[k] = kv[k].idx;
    free(kv); free(cursor);

    c->final_T = (float *)malloc(4 * (size_t)W * H);
    c->n_contrib = (int *)malloc(4 * (size_t)W * H);
    c->margin_alpha = (float *)malloc(4 * (size_t)W * H);
    c->margin_T = (float *)malloc(4 * (size_t)W * H);
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < tiles; t++) {
        int tx = t % c->gx, ty = t / c->gx;
        int64_t s = counts[t], e = counts[t + 1];
        for (int py = ty * TILE; py < ty * TILE + TILE && py < H; py++)
            for (int px = tx * TILE; px < tx * TILE + TILE && px < W; px++) {
                float T = 1.0f, C[3] = {0, 0, 0}, invd = 0.f;
                float m_alpha = INFINITY, m_T = INFINITY;
                int contributor = 0, last = 0;
                for (int64_t k = s; k < e; k++) {
                    contributor++;
                    int g = c->point_list[k];
                    float dx = c->xy[2 * g] - (float)px, dy = c->xy[2 * g + 1] - (float)py;
                    const float *co = c->conic_o + 4 * g;
                    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    float alpha = fminf(0.99f, co[3] * expf(power));
                    /* distance to the threshold in units of what float32 can resolve HERE: the quadratic form is a sum of
                     * cancelling terms, its rounding error ~3 eps (|A dx^2| + |C dy^2| + |2 B dx dy|) / 2 is the RELATIVE error of
                     * alpha -- 1e-7 per unit of term magnitude, i.e. 0.02 eps_alpha per unit at the eps_alpha = 1e-5 the tests
                     * use: nothing for ordinary splats (terms <= 10), x20 for a frame-filling one 100 px off its centre
                     * (terms ~1e3; fuzz seed 2446: a flip at relative distance 6e-5 there) */
                    float tmag = 0.5f * (fabsf(co[0] * dx * dx) + fabsf(co[2] * dy * dy)) + fabsf(co[1] * dx * dy);
                    m_alpha = fminf(m_alpha, fabsf(alpha * 255.0f - 1.0f) / (1.0f + 0.02f * tmag));
                    if (alpha < 1.0f / 255.0f) continue;
                    float test_T = T * (1 - alpha);
                    m_T = fminf(m_T, fabsf(test_T * 10000.0f - 1.0f));
                    if (test_T < 0.0001f) break;
                    for (int ch = 0; ch < 3; ch++) C[ch] += c->rgb[3 * g + ch] * alpha * T;
                    invd += (1.f / c->depth[g]) * alpha * T;
                    T = test_T;
                    last = contributor;
                }
                size_t pid = (size_t)py * W + px;
                c->final_T[pid] = T;
                c->n_contrib[pid] = last;
                c->margin_alpha[pid] = m_alpha;
                c->margin_T[pid] = m_T;
                for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
                if (out_invdepth) out_invdepth[pid] = invd;
            }
    }
    return c;
}

/* Decision margins.  The algorithm is discontinuous in two places per (pixel, Gaussian): alpha < 1/255 drops the splat, and
 * T (1 - alpha) < 1e-4 ends the pixel.  Two correct float32 implementations (different exp, different FMA contraction
 * in the quadratic form) disagree on such a decision whenever the compared value sits within their rounding difference
 * of the threshold; the pixel then moves by up to alpha T <= 1/255 and so do the gradients of the Gaussians on it.  A
 * pixel is MARGINAL when some alpha it evaluated lies within eps_alpha (relative, scaled up by the magnitude of the terms of
 * its quadratic form where that is large: see the forward) of 1/255 or some test_T within eps_T
 * (relative) of 1e-4; a Gaussian is marginal when it can contribute (alpha >= (1 - eps_alpha)/255) to a marginal pixel
 * -- a flipped decision changes T for everything behind it and the colour accumulated behind everything before it.
 * pix_flag (H*W) and gauss_flag (P) receive 0/1.  Parity tests hold every NON-marginal pixel / Gaussian to the strict
 * bars with no allowance; returns the number of marginal pixels. */
int64_t ro_marginal(const ro_ctx *c, float eps_alpha, float eps_T, uint8_t *pix_flag, uint8_t *gauss_flag) {
    const int W = c->W, H = c->H;
    int64_t n = 0;
    memset(gauss_flag, 0, (size_t)(c->P > 0 ? c->P : 1));
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            size_t pid = (size_t)py * W + px;
            int marg = (c->margin_alpha[pid] < eps_alpha) || (c->margin_T[pid] < eps_T);
            pix_flag[pid] = (uint8_t)marg;
            if (!marg) continue;
            n++;
            int t = (py / TILE) * c->gx + px / TILE;
            for (int64_t k = c->tile_start[t]; k < c->tile_start[t + 1]; k++) {
                int g = c->point_list[k];
                float dx = c->xy[2 * g] - (float)px, dy = c->xy[2 * g + 1] - (float)py;
                const float *co = c->conic_o + 4 * g;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 1e-5f) continue;
                float alpha = fminf(0.99f, co[3] * expf(power));
                if (alpha * 255.0f >= 1.0f - eps_alpha) gauss_flag[g] = 1;
            }
        }
    return n;
}

/* Shared decisions.  ro_backward reads the per-pixel termination (final_T, n_contrib) from the context.
 * A parity test may replace the oracle's own with the ones the implementation under test produced in ITS forward: both
 * backwards then walk the same entries of every pixel, whatever side of T (1 - alpha) < 1e-4 either forward fell on, and
 * the comparison needs no mask on the incoming gradient.  What stays implementation-specific is the per-entry
 * alpha >= 1/255 decision (two float32 exps): ro_alpha_marginal reports the Gaussians it can affect. */
void ro_set_termination(ro_ctx *c, const float *final_T, const int *n_contrib) {
    memcpy(c->final_T, final_T, 4 * (size_t)c->W * c->H);
    memcpy(c->n_contrib, n_contrib, 4 * (size_t)c->W * c->H);
}

/* gauss_flag[g] = 1 iff Gaussian g, on some pixel and at a list position the backward walks there (<= n_contrib, as the
 * context holds it NOW), has an alpha within eps_alpha of 1/255 (relative; in units of what float32 resolves at that pixel,
 * the forward's rule).  Returns the number of flagged Gaussians. */
int64_t ro_alpha_marginal(const ro_ctx *c, float eps_alpha, uint8_t *gauss_flag) {
    const int W = c->W, H = c->H, tiles = c->gx * c->gy;
    memset(gauss_flag, 0, (size_t)(c->P > 0 ? c->P : 1));
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < tiles; t++) {
        int tx = t % c->gx, ty = t / c->gx;
        int64_t s = c->tile_start[t];
        for (int py = ty * TILE; py < ty * TILE + TILE && py < H; py++)
            for (int px = tx * TILE; px < tx * TILE + TILE && px < W; px++) {
                size_t pid = (size_t)py * W + px;
                int64_t lim = s + c->n_contrib[pid];
                if (lim > c->tile_start[t + 1]) lim = c->tile_start[t + 1];
                for (int64_t k = s; k < lim; k++) {
                    int g = c->point_list[k];
                    float dx = c->xy[2 * g] - (float)px, dy = c->xy[2 * g + 1] - (float)py;
                    const float *co = c->conic_o + 4 * g;
                    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 1e-5f) continue;
                    float alpha = fminf(0.99f, co[3] * expf(power));
                    float tmag = 0.5f * (fabsf(co[0] * dx * dx) + fabsf(co[2] * dy * dy)) + fabsf(co[1] * dx * dy);
                    if (fabsf(alpha * 255.0f - 1.0f) / (1.0f + 0.02f * tmag) < eps_alpha) gauss_flag[g] = 1;   /* (benign race: all writers store 1) */
                }
            }
    }
    int64_t n = 0;
    for (int i = 0; i < c->P; i++) n += gauss_flag[i];
    return n;
}

/* The (Gaussian, pixel) pairs whose alpha lies within eps_alpha of 1/255 (same measure as ro_alpha_marginal), at list positions
 * the backward walks.  gid / pix receive up to max_n pairs (pix = py * W + px); returns the number found (may exceed max_n). */
int64_t ro_alpha_band_pairs(const ro_ctx *c, float eps_alpha, int64_t max_n, int *gid, int *pix) {
    const int W = c->W, H = c->H, tiles = c->gx * c->gy;
    int64_t n = 0;
    for (int t = 0; t < tiles; t++) {
        int tx = t % c->gx, ty = t / c->gx;
        int64_t s = c->tile_start[t];
        for (int py = ty * TILE; py < ty * TILE + TILE && py < H; py++)
            for (int px = tx * TILE; px < tx * TILE + TILE && px < W; px++) {
                size_t pid = (size_t)py * W + px;
                int64_t lim = s + c->n_contrib[pid];
                if (lim > c->tile_start[t + 1]) lim = c->tile_start[t + 1];
                for (int64_t k = s; k < lim; k++) {
                    int g = c->point_list[k];
                    float dx = c->xy[2 * g] - (float)px, dy = c->xy[2 * g + 1] - (float)py;
                    const float *co = c->conic_o + 4 * g;
                    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 1e-5f) continue;
                    float alpha = fminf(0.99f, co[3] * expf(power));
                    float tmag = 0.5f * (fabsf(co[0] * dx * dx) + fabsf(co[2] * dy * dy)) + fabsf(co[1] * dx * dy);
                    if (fabsf(alpha * 255.0f - 1.0f) / (1.0f + 0.02f * tmag) < eps_alpha) {
                        if (n < max_n) { gid[n] = g; pix[n] = (int)pid; }
                        n++;
                    }
                }
            }
    }
    return n;
}

/* Shared alpha decisions: for the listed (Gaussian, pixel) pairs ro_backward takes `ok` (1: the pair is blended with the
 * alpha this oracle computes, 0: skipped) instead of its own alpha >= 1/255 test -- the decisions of the implementation
 * under test at exactly the pairs where two float32 exps may disagree.  The pairs MUST be sorted by (pix, gid) (the order
 * ro_alpha_band_pairs produces per pixel is list order: the wrapper sorts).  n = 0 clears. */
void ro_set_alpha_overrides(ro_ctx *c, int64_t n, const int *gid, const int *pix, const uint8_t *ok) {
    free(c->ovr_pix); free(c->ovr_gid); free(c->ovr_ok); free(c->ovr_has);
    c->ovr_pix = c->ovr_gid = NULL; c->ovr_ok = c->ovr_has = NULL; c->ovr_n = 0;
    if (n <= 0) return;
    c->ovr_n = n;
    c->ovr_pix = (int *)malloc(4 * (size_t)n); c->ovr_gid = (int *)malloc(4 * (size_t)n); c->ovr_ok = (uint8_t *)malloc((size_t)n);
    c->ovr_has = (uint8_t *)calloc((size_t)c->W * c->H, 1);
    memcpy(c->ovr_pix, pix, 4 * (size_t)n); memcpy(c->ovr_gid, gid, 4 * (size_t)n); memcpy(c->ovr_ok, ok, (size_t)n);
    for (int64_t i = 0; i < n; i++) c->ovr_has[pix[i]] = 1;
}
/* -1: no override for the pair, else 0 / 1 */
static int alpha_override(const ro_ctx *c, int pid, int g) {
    int64_t lo = 0, hi = c->ovr_n;
    while (lo < hi) {
        int64_t mid = (lo + hi) / 2;
        if (c->ovr_pix[mid] < pid || (c->ovr_pix[mid] == pid && c->ovr_gid[mid] < g)) lo = mid + 1; else hi = mid;
    }
    return (lo < c->ovr_n && c->ovr_pix[lo] == pid && c->ovr_gid[lo] == g) ? (int)c->ovr_ok[lo] : -1;
}

/* Per-Gaussian sums over pixels are accumulated in DOUBLE: the terms are the float32 values a tile splatter
 * produces, but their sum is then independent of the summation order (the GPU sums them in a different order, and
 * a float32 running sum over 10^3..10^4 signed terms carries ~1e-3 relative noise of its own). */
static void atomic_addd(double *p, double v) {
#pragma omp atomic
    *p += v;
}

static double g_sum_noise_sigma = 0.0;
static int g_sum_noise_pattern = 0;
void ro_set_sum_noise(double sigma, int pattern) { g_sum_noise_sigma = sigma; g_sum_noise_pattern = pattern; }
static double noise_sign(int slot, int pattern) {
    uint32_t h = (uint32_t)slot * 2654435761u ^ ((uint32_t)pattern + 1u) * 0x9E3779B9u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return (h & 1u) ? 1.0 : -1.0;
}

/* pattern >= 1000: ONE of the 13 perturbed quantities of a Gaussian at a time (kind = pattern - 1000: the four dL/dconic sums,
 * three colour sums, two mean2D sums, the opacity sum, the three cov2D entries), sign +1 -- the chain behind the sums is
 * per Gaussian and first-order linear in these, so the sum over the 13 kinds of |effect| is the effect of the WORST sign
 * combination (tests/util.py: conditioning_noise); a handful of random patterns only samples it. */
static double noise_sign_k(int kind, int slot, int pattern) {
    if (pattern >= 1000) return pattern - 1000 == kind ? 1.0 : 0.0;
    return noise_sign(slot, pattern);
}

/* Backward.  dL_dpix (3,H,W).  All outputs must be zero-initialised by the caller; NULL where not applicable. */
void ro_backward(const ro_ctx *c, const float *means3D, const float *shs, const float *scales, const float *rots,
                 const float *dL_dpix, float *dL_dmeans3D, float *dL_dmeans2D /*P,3*/, float *dL_dsh,
                 float *dL_dcolors, float *dL_dopacity, float *dL_dscales, float *dL_drots, float *dL_dcov3D) {
    int P = c->P, W = c->W, H = c->H, tiles = c->gx * c->gy;
    double *dL_dconic = (double *)calloc((size_t)(P > 0 ? P : 1), 32);
    double *dL_drgb = (double *)calloc((size_t)(P > 0 ? P : 1), 24);
    double *dL_dm2 = (double *)calloc((size_t)(P > 0 ? P : 1), 16);
    double *dL_dop = (double *)calloc((size_t)(P > 0 ? P : 1), 8);
    double *dL_dinvd = (double *)calloc((size_t)(P > 0 ? P : 1), 8);       /* dL/d(1/depth) per Gaussian (ro_set_invdepth_grad) */
    const float *gdepth = g_dL_dinvdepth;
    /* conditioning probe only: the sums of the ABSOLUTE values of the same terms */
    const int probe = g_sum_noise_sigma != 0.0;
    double *ab_conic = (double *)calloc((size_t)(probe && P > 0 ? P : 1), 32);
    double *ab_rgb = (double *)calloc((size_t)(probe && P > 0 ? P : 1), 24);
    double *ab_m2 = (double *)calloc((size_t)(probe && P > 0 ? P : 1), 16);
    double *ab_op = (double *)calloc((size_t)(probe && P > 0 ? P : 1), 8);
    float *dL_dcov = dL_dcov3D ? dL_dcov3D : (float *)calloc((size_t)(P > 0 ? P : 1), 24);

#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < tiles; t++) {
        int tx = t % c->gx, ty = t / c->gx;
        int64_t s = c->tile_start[t], e = c->tile_start[t + 1];
        for (int py = ty * TILE; py < ty * TILE + TILE && py < H; py++)
            for (int px = tx * TILE; px < tx * TILE + TILE && px < W; px++) {
                size_t pid = (size_t)py * W + px;
                const float T_final = c->final_T[pid];
                float T = T_final;
                int last = c->n_contrib[pid];
                float accum[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0.f;
                float accum_d = 0.f, last_invd = 0.f;
                const float dLd = gdepth ? gdepth[pid] : 0.f;
                float dLp[3] = {dL_dpix[pid], dL_dpix[(size_t)H * W + pid], dL_dpix[(size_t)2 * H * W + pid]};
                float bg_dot = c->bg[0] * dLp[0] + c->bg[1] * dLp[1] + c->bg[2] * dLp[2];
                for (int64_t k = s + last - 1; k >= s; k--) {
                    int g = c->point_list[k];
                    float dx = c->xy[2 * g] - (float)px, dy = c->xy[2 * g + 1] - (float)py;
                    const float *co = c->conic_o + 4 * g;
                    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    float G = expf(power);
                    float alpha = fminf(0.99f, co[3] * G);
                    int forced = (c->ovr_n && c->ovr_has[pid]) ? alpha_override(c, (int)pid, g) : -1;
                    if (forced < 0 ? (alpha < 1.0f / 255.0f) : !forced) continue;
                    T = T / (1.f - alpha);
                    float dch = alpha * T;
                    float dL_dalpha = 0.f;
                    for (int ch = 0; ch < 3; ch++) {
                        float col = c->rgb[3 * g + ch];
                        accum[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum[ch];
                        last_color[ch] = col;
                        dL_dalpha += (col - accum[ch]) * dLp[ch];
                        atomic_addd(&dL_drgb[3 * g + ch], dch * dLp[ch]);
                        if (probe) atomic_addd(&ab_rgb[3 * g + ch], fabsf(dch * dLp[ch]));
                    }
                    if (gdepth) {
                        float invd = 1.f / c->depth[g];
                        accum_d = last_alpha * last_invd + (1.f - last_alpha) * accum_d;
                        last_invd = invd;
                        dL_dalpha += (invd - accum_d) * dLd;
                        atomic_addd(&dL_dinvd[g], dch * dLd);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                    float dL_dG = co[3] * dL_dalpha;           /* alpha clamp: gradient passes through */
                    float gdx = G * dx, gdy = G * dy;
                    float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    float dG_ddely = -gdy * co[2] - gdx * co[1];
                    atomic_addd(&dL_dm2[2 * g], dL_dG * dG_ddelx * 0.5f * W);
                    atomic_addd(&dL_dm2[2 * g + 1], dL_dG * dG_ddely * 0.5f * H);
                    atomic_addd(&dL_dconic[4 * g], -0.5f * gdx * dx * dL_dG);
                    atomic_addd(&dL_dconic[4 * g + 1], -0.5f * gdx * dy * dL_dG); /* HALF of dL/dB; doubled below */
                    atomic_addd(&dL_dconic[4 * g + 3], -0.5f * gdy * dy * dL_dG);
                    atomic_addd(&dL_dop[g], G * dL_dalpha);
                    if (probe) {
                        atomic_addd(&ab_m2[2 * g], fabsf(dL_dG * dG_ddelx * 0.5f * W));
                        atomic_addd(&ab_m2[2 * g + 1], fabsf(dL_dG * dG_ddely * 0.5f * H));
                        atomic_addd(&ab_conic[4 * g], fabsf(0.5f * gdx * dx * dL_dG));
                        atomic_addd(&ab_conic[4 * g + 1], fabsf(0.5f * gdx * dy * dL_dG));
                        atomic_addd(&ab_conic[4 * g + 3], fabsf(0.5f * gdy * dy * dL_dG));
                        atomic_addd(&ab_op[g], fabsf(G * dL_dalpha));
                    }
                }
            }
    }

    /* Conditioning probe (tests/util.py: conditioning_noise): every per-Gaussian pixel sum is moved by +- gamma x (the sum of
     * the absolute values of its terms), sign from a hash of (slot, pattern) -- the textbook bound on the error of a float32
     * sum taken in any order (gamma = depth x eps), which a GPU implementation and upstream's float atomics carry in these
     * sums while this oracle sums in double.  How far the OUTPUTS move under it measures how ill-conditioned the chain behind
     * the sums is for that Gaussian (a frame-filling splat: the terms of dL/dconic carry dx^2 ~ 1e4 and alternate in sign, and
     * the conic -> cov2D step cancels again).  gamma = 0: off. */
    if (probe) {
        for (int i = 0; i < P; i++) {
            for (int k = 0; k < 4; k++) dL_dconic[4 * i + k] += g_sum_noise_sigma * ab_conic[4 * i + k] * noise_sign_k(k, 4 * i + k, g_sum_noise_pattern);
            for (int k = 0; k < 3; k++) dL_drgb[3 * i + k] += g_sum_noise_sigma * ab_rgb[3 * i + k] * noise_sign_k(4 + k, 3 * i + k + 101, g_sum_noise_pattern);
            for (int k = 0; k < 2; k++) dL_dm2[2 * i + k] += g_sum_noise_sigma * ab_m2[2 * i + k] * noise_sign_k(7 + k, 2 * i + k + 202, g_sum_noise_pattern);
            dL_dop[i] += g_sum_noise_sigma * ab_op[i] * noise_sign_k(9, i + 303, g_sum_noise_pattern);
        }
    }
    free(ab_conic); free(ab_rgb); free(ab_m2); free(ab_op);
    for (int i = 0; i < P; i++) {
        dL_dmeans2D[3 * i] = (float)dL_dm2[2 * i]; dL_dmeans2D[3 * i + 1] = (float)dL_dm2[2 * i + 1];
        dL_dopacity[i] = (float)dL_dop[i] * c->aa[i];      /* dL_dop: w.r.t. opacity x h */
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (!(c->radii[i] > 0)) continue;
        const float *m = means3D + 3 * i;
        const float *c6 = c->cov3D + 6 * i;
        float dmean[3] = {0, 0, 0};
        /* ---- conic -> cov2D -> cov3D and mean (through T) ---- */
        {
            float t[3], T[6]; int clx, cly;
            ewa_T(c, m, t, T, &clx, &cly);
            float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
            float TS[6];
            for (int r = 0; r < 2; r++)
                for (int k = 0; k < 3; k++)
                    TS[3 * r + k] = T[3 * r] * S[k] + T[3 * r + 1] * S[3 + k] + T[3 * r + 2] * S[6 + k];
            float a = TS[0] * T[0] + TS[1] * T[1] + TS[2] * T[2] + 0.3f;
            float b = TS[0] * T[3] + TS[1] * T[4] + TS[2] * T[5];
            float cc = TS[3] * T[3] + TS[4] * T[4] + TS[5] * T[5] + 0.3f;
            float dcx = (float)dL_dconic[4 * i], dcy = (float)dL_dconic[4 * i + 1], dcz = (float)dL_dconic[4 * i + 3];
            if (g_sum_noise_sigma != 0.0) {
                /* conditioning probe, second part: the cov2D entries carry their own float32 rounding (2 ulp here), and the
                 * step behind them -- d conic / d cov2D, written with (denom - a c) = -b^2 as upstream writes it -- cancels
                 * catastrophically for a splat hundreds of pixels wide (a c ~ 1e8 against b^2): two float32 evaluations of
                 * this chain that round differently disagree by what this perturbation shows */
                a *= 1.0f + 2.4e-7f * (float)noise_sign_k(10, 7 * i + 1, g_sum_noise_pattern);
                b *= 1.0f + 2.4e-7f * (float)noise_sign_k(11, 7 * i + 2, g_sum_noise_pattern);
                cc *= 1.0f + 2.4e-7f * (float)noise_sign_k(12, 7 * i + 3, g_sum_noise_pattern);
            }
            float denom = a * cc - b * b;
            float d2inv = 1.0f / ((denom * denom) + 0.0000001f);
            float dL_da = 0, dL_db = 0, dL_dc = 0;
            if (d2inv != 0) {
                dL_da = d2inv * (-cc * cc * dcx + 2 * b * cc * dcy + (denom - a * cc) * dcz);
                dL_dc = d2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * cc) * dcx);
                dL_db = d2inv * 2 * (b * cc * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
                if (c->antialiasing) {
                    /* h = sqrt(max(floor, rho)), rho = (x y - z^2) / ((x + w)(y + w) - z^2) with x, y the UNdilated diagonal:
                     * d rho/dx = w (w y + y^2 + z^2) / D^2, d rho/dy = w (w x + x^2 + z^2) / D^2, d rho/dz = -2 w z (w + x + y) / D^2 */
                    float x = a - 0.3f, y = cc - 0.3f, z = b, w = 0.3f;
                    float rho = (x * y - z * z) / denom;
                    float h = sqrtf(fmaxf(0.000025f, rho));
                    float g_h = (float)dL_dop[i] * c->op_raw[i];
                    float g_rho = rho <= 0.000025f ? 0.f : g_h / (2.f * h);
                    float k = g_rho * w / (denom * denom);
                    dL_da += k * (w * y + y * y + z * z);
                    dL_dc += k * (w * x + x * x + z * z);
                    dL_db += -2.f * k * z * (w + x + y);
                }
                float *o = dL_dcov + 6 * i;
                /* symmetric 6-vector: off-diagonal entries receive the summed gradient */
                o[0] += T[0] * T[0] * dL_da + T[0] * T[3] * dL_db + T[3] * T[3] * dL_dc;
                o[3] += T[1] * T[1] * dL_da + T[1] * T[4] * dL_db + T[4] * T[4] * dL_dc;
                o[5] += T[2] * T[2] * dL_da + T[2] * T[5] * dL_db + T[5] * T[5] * dL_dc;
                o[1] += 2 * T[0] * T[1] * dL_da + (T[0] * T[4] + T[1] * T[3]) * dL_db + 2 * T[3] * T[4] * dL_dc;
                o[2] += 2 * T[0] * T[2] * dL_da + (T[0] * T[5] + T[2] * T[3]) * dL_db + 2 * T[3] * T[5] * dL_dc;
                o[4] += 2 * T[2] * T[1] * dL_da + (T[1] * T[5] + T[2] * T[4]) * dL_db + 2 * T[4] * T[5] * dL_dc;
            }
            /* dL/dT (2x3): cov2D = T S T^T */
            float dT[6];
            for (int k = 0; k < 3; k++) {
                dT[k] = 2 * TS[k] * dL_da + TS[3 + k] * dL_db;
                dT[3 + k] = 2 * TS[3 + k] * dL_dc + TS[k] * dL_db;
            }
            const float *v = c->view;
            /* T[r][k] = sum_j J[r][j] Wr[j][k],  Wr[j][k] = v[4k + j] */
            float dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
            for (int k = 0; k < 3; k++) {
                dJ00 += v[4 * k + 0] * dT[k];
                dJ02 += v[4 * k + 2] * dT[k];
                dJ11 += v[4 * k + 1] * dT[3 + k];
                dJ12 += v[4 * k + 2] * dT[3 + k];
            }
            float fx = c->W / (2.f * c->tanfovx), fy = c->H / (2.f * c->tanfovy);
            float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
            float dtx = (clx ? 0.f : 1.f) * -fx * tz2 * dJ02;
            float dty = (cly ? 0.f : 1.f) * -fy * tz2 * dJ12;
            float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t[0]) * tz3 * dJ02 + (2 * fy * t[1]) * tz3 * dJ12;
            /* back through the view rotation: t = Wr * mean + trans */
            dmean[0] += v[0] * dtx + v[1] * dty + v[2] * dtz;
            dmean[1] += v[4] * dtx + v[5] * dty + v[6] * dtz;
            dmean[2] += v[8] * dtx + v[9] * dty + v[10] * dtz;
        }
        /* ---- inverse depth -> mean3D: d(1/z)/dmean = -view_z / z^2 ---- */
        if (gdepth) {
            const float *v = c->view;
            float z = c->depth[i];
            float gz = -(float)dL_dinvd[i] / (z * z);
            dmean[0] += v[2] * gz; dmean[1] += v[6] * gz; dmean[2] += v[10] * gz;
        }
        /* ---- mean2D -> mean3D through the perspective projection ---- */
        {
            const float *p = c->proj;
            float mh[4];
            xform4x4(p, m, mh);
            float mw = 1.0f / (mh[3] + 0.0000001f);
            float mul1 = (p[0] * m[0] + p[4] * m[1] + p[8] * m[2] + p[12]) * mw * mw;
            float mul2 = (p[1] * m[0] + p[5] * m[1] + p[9] * m[2] + p[13]) * mw * mw;
            float gx = dL_dmeans2D[3 * i], gy = dL_dmeans2D[3 * i + 1];
            dmean[0] += (p[0] * mw - p[3] * mul1) * gx + (p[1] * mw - p[3] * mul2) * gy;
            dmean[1] += (p[4] * mw - p[7] * mul1) * gx + (p[5] * mw - p[7] * mul2) * gy;
            dmean[2] += (p[8] * mw - p[11] * mul1) * gx + (p[9] * mw - p[11] * mul2) * gy;
        }
        /* ---- colour ---- */
        if (c->has_sh) {
            const float *sh = shs + (size_t)3 * c->M * i;
            float *dsh = dL_dsh + (size_t)3 * c->M * i;
            float dir0[3] = {m[0] - c->campos[0], m[1] - c->campos[1], m[2] - c->campos[2]};
            float len2 = dir0[0] * dir0[0] + dir0[1] * dir0[1] + dir0[2] * dir0[2];
            float inv = 1.f / sqrtf(len2);
            float x = dir0[0] * inv, y = dir0[1] * inv, z = dir0[2] * inv;
            float dRGB[3];
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = c->clamped[3 * i + ch] ? 0.f : (float)dL_drgb[3 * i + ch];
            float ddir[3] = {0, 0, 0}; /* dL/d(unit dir) */
            for (int ch = 0; ch < 3; ch++) {
#define SHC(k) sh[3 * (k) + ch]
#define DSH(k) dsh[3 * (k) + ch]
                float g = dRGB[ch];
                float dx_ = 0, dy_ = 0, dz_ = 0;
                DSH(0) = SH_C0 * g;
                if (c->deg > 0) {
                    DSH(1) = -SH_C1 * y * g; DSH(2) = SH_C1 * z * g; DSH(3) = -SH_C1 * x * g;
                    dx_ = -SH_C1 * SHC(3); dy_ = -SH_C1 * SHC(1); dz_ = SH_C1 * SHC(2);
                    if (c->deg > 1) {
                        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        DSH(4) = SH_C2[0] * xy * g; DSH(5) = SH_C2[1] * yz * g;
                        DSH(6) = SH_C2[2] * (2.f * zz - xx - yy) * g;
                        DSH(7) = SH_C2[3] * xz * g; DSH(8) = SH_C2[4] * (xx - yy) * g;
                        dx_ += SH_C2[0] * y * SHC(4) + SH_C2[2] * 2.f * -x * SHC(6) + SH_C2[3] * z * SHC(7) +
                               SH_C2[4] * 2.f * x * SHC(8);
                        dy_ += SH_C2[0] * x * SHC(4) + SH_C2[1] * z * SHC(5) + SH_C2[2] * 2.f * -y * SHC(6) +
                               SH_C2[4] * 2.f * -y * SHC(8);
                        dz_ += SH_C2[1] * y * SHC(5) + SH_C2[2] * 2.f * 2.f * z * SHC(6) + SH_C2[3] * x * SHC(7);
                        if (c->deg > 2) {
                            DSH(9) = SH_C3[0] * y * (3.f * xx - yy) * g; DSH(10) = SH_C3[1] * xy * z * g;
                            DSH(11) = SH_C3[2] * y * (4.f * zz - xx - yy) * g;
                            DSH(12) = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                            DSH(13) = SH_C3[4] * x * (4.f * zz - xx - yy) * g;
                            DSH(14) = SH_C3[5] * z * (xx - yy) * g; DSH(15) = SH_C3[6] * x * (xx - 3.f * yy) * g;
                            dx_ += SH_C3[0] * SHC(9) * 3.f * 2.f * xy + SH_C3[1] * SHC(10) * yz +
                                   SH_C3[2] * SHC(11) * -2.f * xy + SH_C3[3] * SHC(12) * -3.f * 2.f * xz +
                                   SH_C3[4] * SHC(13) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SHC(14) * 2.f * xz +
                                   SH_C3[6] * SHC(15) * 3.f * (xx - yy);
                            dy_ += SH_C3[0] * SHC(9) * 3.f * (xx - yy) + SH_C3[1] * SHC(10) * xz +
                                   SH_C3[2] * SHC(11) * (-3.f * yy + 4.f * zz - xx) +
                                   SH_C3[3] * SHC(12) * -3.f * 2.f * yz + SH_C3[4] * SHC(13) * -2.f * xy +
                                   SH_C3[5] * SHC(14) * -2.f * yz + SH_C3[6] * SHC(15) * -3.f * 2.f * xy;
                            dz_ += SH_C3[1] * SHC(10) * xy + SH_C3[2] * SHC(11) * 4.f * 2.f * yz +
                                   SH_C3[3] * SHC(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SHC(13) * 4.f * 2.f * xz +
                                   SH_C3[5] * SHC(14) * (xx - yy);
                        }
                    }
                }
#undef SHC
#undef DSH
                ddir[0] += dx_ * g; ddir[1] += dy_ * g; ddir[2] += dz_ * g;
            }
            /* d(normalize(v))/dv applied to ddir */
            float inv3 = inv * inv * inv;
            float vx = dir0[0], vy = dir0[1], vz = dir0[2];
            dmean[0] += ((len2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * inv3;
            dmean[1] += (-vx * vy * ddir[0] + (len2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * inv3;
            dmean[2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (len2 - vz * vz) * ddir[2]) * inv3;
        } else if (dL_dcolors) {
            for (int ch = 0; ch < 3; ch++) dL_dcolors[3 * i + ch] = (float)dL_drgb[3 * i + ch];
        }
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * i + k] += dmean[k];
        /* ---- cov3D -> scale / rotation ---- */
        if (c->from_scale_rot && dL_dscales && dL_drots) {
            const float *q = rots + 4 * i, *s = scales + 3 * i, *g6 = dL_dcov + 6 * i;
            float r = q[0], x = q[1], y = q[2], z = q[3];
            float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                          2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                          2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
            float mod = c->scale_modifier;
            float sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
            /* G = dL/dSigma as a full symmetric matrix (off-diagonals halved) */
            float G[9] = {g6[0], 0.5f * g6[1], 0.5f * g6[2], 0.5f * g6[1], g6[3], 0.5f * g6[4],
                          0.5f * g6[2], 0.5f * g6[4], g6[5]};
            /* Sigma = L L^T, L = R diag(sc):  dL/dL = 2 G L */
            float L[9], dLm[9];
            for (int a_ = 0; a_ < 3; a_++)
                for (int b_ = 0; b_ < 3; b_++) L[3 * a_ + b_] = R[3 * a_ + b_] * sc[b_];
            for (int a_ = 0; a_ < 3; a_++)
                for (int b_ = 0; b_ < 3; b_++) {
                    float acc = 0;
                    for (int k = 0; k < 3; k++) acc += G[3 * a_ + k] * L[3 * k + b_];
                    dLm[3 * a_ + b_] = 2.f * acc;
                }
            float dR[9];
            for (int b_ = 0; b_ < 3; b_++) {
                float ds = 0;
                for (int a_ = 0; a_ < 3; a_++) {
                    ds += dLm[3 * a_ + b_] * R[3 * a_ + b_];
                    dR[3 * a_ + b_] = dLm[3 * a_ + b_] * sc[b_];
                }
                dL_dscales[3 * i + b_] = ds * mod;
            }
            float dr = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            float dx_ = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] -
                               2.f * x * dR[8]);
            float dy_ = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] -
                               2.f * y * dR[8]);
            float dz_ = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] +
                               x * dR[6] + y * dR[7]);
            dL_drots[4 * i] = dr; dL_drots[4 * i + 1] = dx_; dL_drots[4 * i + 2] = dy_; dL_drots[4 * i + 3] = dz_;
        }
    }
    free(dL_dconic); free(dL_drgb); free(dL_dm2); free(dL_dop); free(dL_dinvd);
    if (!dL_dcov3D) free(dL_dcov);
}

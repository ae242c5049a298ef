// d3ga_math.h -- per-element arithmetic of the deform-and-rasterize path.
//
// Pure functions on registers, shared by the gfx950 kernels (compiled by hipcc as __device__) and by the
// host-side arithmetic self-check in tests/hostcheck (compiled by g++, D3GA_HD empty) so that the formulas can
// be checked against the oracle on a box without a GPU.  No memory access, no torch, no HIP runtime here.
//
// Conventions (reference file:line):
//   * covariance 6-vector order xx,xy,xz,yy,yz,zz                 utils/general_utils.py:24-35
//   * quaternion (w,x,y,z)                                         utils/general_utils.py:58-79
//   * tet edge matrix columns (v3-v0, v2-v0, v1-v0)                lib/tet_mesh.py:88-94
//   * 4x4 matrices are the reference's transposed (row-vector) matrices, flattened row-major:
//     p' = [p,1] M  =>  p'.x = m[0]x + m[4]y + m[8]z + m[12]       lib/cameras.py:68-74
#pragma once
#include <math.h>
#include <stdint.h>

// The functions that decide INTEGERS -- radius, tile rectangle, depth order, and through the conic the alpha thresholds --
// are compiled without FMA contraction: the oracle (gcc -ffp-contract=off) and this code then perform the same correctly
// rounded operations in the same order and agree bit for bit.  (With hipcc's default contraction the two differed in the
// last ulp; a 300-seed fuzz campaign found one radius in 120 000 that ceil() then rounded differently.)
#if defined(__clang__)
#define D3GA_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define D3GA_NO_CONTRACT
#endif

#ifndef D3GA_HD
#ifdef __HIPCC__
#define D3GA_HD __host__ __device__ __forceinline__
#else
#define D3GA_HD static inline
#define __expf expf
#endif
#endif

namespace d3ga {

constexpr int kTile = 16;               // tile edge in pixels
constexpr float kNear = 0.2f;           // view-space z cull
constexpr float kDilate = 0.3f;         // screen-space low-pass added to the 2D covariance diagonal
constexpr float kAlphaMax = 0.99f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTmin = 0.0001f;

struct V3 { float x, y, z; };
D3GA_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
D3GA_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
D3GA_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
D3GA_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
D3GA_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// 3x3 row-major
struct M3 { float m[9]; };
D3GA_HD M3 matmul(const M3 &A, const M3 &B) {
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
D3GA_HD M3 matmul_nt(const M3 &A, const M3 &B) {
    D3GA_NO_CONTRACT  // A * B^T
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C.m[3 * i + j] = A.m[3 * i] * B.m[3 * j] + A.m[3 * i + 1] * B.m[3 * j + 1] + A.m[3 * i + 2] * B.m[3 * j + 2];
    return C;
}
D3GA_HD M3 matmul_tn(const M3 &A, const M3 &B) {  // A^T * B
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C.m[3 * i + j] = A.m[i] * B.m[j] + A.m[3 + i] * B.m[3 + j] + A.m[6 + i] * B.m[6 + j];
    return C;
}

// ---------------------------------------------------------------------------------------------------------
// rotation from a quaternion
// ---------------------------------------------------------------------------------------------------------
D3GA_HD M3 quat_to_rot(float w, float x, float y, float z) {
    D3GA_NO_CONTRACT
    M3 R;
    R.m[0] = 1.f - 2.f * (y * y + z * z); R.m[1] = 2.f * (x * y - w * z);       R.m[2] = 2.f * (x * z + w * y);
    R.m[3] = 2.f * (x * y + w * z);       R.m[4] = 1.f - 2.f * (x * x + z * z); R.m[5] = 2.f * (y * z - w * x);
    R.m[6] = 2.f * (x * z - w * y);       R.m[7] = 2.f * (y * z + w * x);       R.m[8] = 1.f - 2.f * (x * x + y * y);
    return R;
}
// dL/dq (w,x,y,z) from dL/dR for the (un-normalised) polynomial above
D3GA_HD void quat_to_rot_bwd(float w, float x, float y, float z, const M3 &dR, float dq[4]) {
    const float *d = dR.m;
    dq[0] = 2.f * (-z * d[1] + y * d[2] + z * d[3] - x * d[5] - y * d[6] + x * d[7]);
    dq[1] = 2.f * (y * d[1] + z * d[2] + y * d[3] - 2.f * x * d[4] - w * d[5] + z * d[6] + w * d[7] - 2.f * x * d[8]);
    dq[2] = 2.f * (-2.f * y * d[0] + x * d[1] + w * d[2] + x * d[3] + z * d[5] - w * d[6] + z * d[7] - 2.f * y * d[8]);
    dq[3] = 2.f * (-2.f * z * d[0] - w * d[1] + x * d[2] + w * d[3] - 2.f * z * d[4] + y * d[5] + x * d[6] + y * d[7]);
}

// ---------------------------------------------------------------------------------------------------------
// D1-D5 cage deformation of one Gaussian      (models/cage_net.py:213-230, lib/cage.py:339-342)
// ---------------------------------------------------------------------------------------------------------
struct DeformIn {
    V3 x0, x1, x2, x3;   // posed tet corners
    float bary[4];
    M3 G;                // canonical gradient inv(Dm)
    float s[3];          // activated scales
    float q[4];          // rotation (w,x,y,z), any norm > 0
};

D3GA_HD M3 tet_edges(V3 x0, V3 x1, V3 x2, V3 x3) {   // columns (x3-x0, x2-x0, x1-x0)
    V3 c0 = x3 - x0, c1 = x2 - x0, c2 = x1 - x0;
    M3 D;
    D.m[0] = c0.x; D.m[1] = c1.x; D.m[2] = c2.x;
    D.m[3] = c0.y; D.m[4] = c1.y; D.m[5] = c2.y;
    D.m[6] = c0.z; D.m[7] = c1.z; D.m[8] = c2.z;
    return D;
}

D3GA_HD void deform_fwd(const DeformIn &in, float mean[3], float cov6[6]) {
    mean[0] = in.bary[0] * in.x0.x + in.bary[1] * in.x1.x + in.bary[2] * in.x2.x + in.bary[3] * in.x3.x;
    mean[1] = in.bary[0] * in.x0.y + in.bary[1] * in.x1.y + in.bary[2] * in.x2.y + in.bary[3] * in.x3.y;
    mean[2] = in.bary[0] * in.x0.z + in.bary[1] * in.x1.z + in.bary[2] * in.x2.z + in.bary[3] * in.x3.z;
    const M3 J = matmul(tet_edges(in.x0, in.x1, in.x2, in.x3), in.G);
    const float qn = 1.0f / sqrtf(in.q[0] * in.q[0] + in.q[1] * in.q[1] + in.q[2] * in.q[2] + in.q[3] * in.q[3]);
    M3 L = quat_to_rot(in.q[0] * qn, in.q[1] * qn, in.q[2] * qn, in.q[3] * qn);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) L.m[3 * a + b] *= in.s[b];
    const M3 A = matmul(J, L);                       // cov = (J L)(J L)^T = J Sigma J^T
    const M3 C = matmul_nt(A, A);
    cov6[0] = C.m[0]; cov6[1] = C.m[1]; cov6[2] = C.m[2]; cov6[3] = C.m[4]; cov6[4] = C.m[5]; cov6[5] = C.m[8];
}

struct DeformGrad {
    V3 gx0, gx1, gx2, gx3;
    float gbary[4];
    float gs[3];
    float gq[4];
};

D3GA_HD void deform_bwd(const DeformIn &in, const float gmean[3], const float gcov6[6], DeformGrad &out) {
    const V3 gm = v3(gmean[0], gmean[1], gmean[2]);
    out.gbary[0] = dot(gm, in.x0); out.gbary[1] = dot(gm, in.x1);
    out.gbary[2] = dot(gm, in.x2); out.gbary[3] = dot(gm, in.x3);
    // recompute forward intermediates
    const M3 J = matmul(tet_edges(in.x0, in.x1, in.x2, in.x3), in.G);
    const float n2 = in.q[0] * in.q[0] + in.q[1] * in.q[1] + in.q[2] * in.q[2] + in.q[3] * in.q[3];
    const float qn = 1.0f / sqrtf(n2);
    const float qw = in.q[0] * qn, qx = in.q[1] * qn, qy = in.q[2] * qn, qz = in.q[3] * qn;
    const M3 R = quat_to_rot(qw, qx, qy, qz);
    M3 L = R;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) L.m[3 * a + b] *= in.s[b];
    const M3 A = matmul(J, L);
    // cov6 lists each off-diagonal once: dL/dA = (Gu + Gu^T) A with Gu upper-triangular
    M3 Gs;
    Gs.m[0] = 2.f * gcov6[0]; Gs.m[1] = gcov6[1];       Gs.m[2] = gcov6[2];
    Gs.m[3] = gcov6[1];       Gs.m[4] = 2.f * gcov6[3]; Gs.m[5] = gcov6[4];
    Gs.m[6] = gcov6[2];       Gs.m[7] = gcov6[4];       Gs.m[8] = 2.f * gcov6[5];
    const M3 dA = matmul(Gs, A);
    const M3 dJ = matmul_nt(dA, L);                  // A = J L
    const M3 dL = matmul_tn(J, dA);
    M3 dR;
    for (int b = 0; b < 3; ++b) {
        float acc = 0.f;
        for (int a = 0; a < 3; ++a) {
            acc += dL.m[3 * a + b] * R.m[3 * a + b];
            dR.m[3 * a + b] = dL.m[3 * a + b] * in.s[b];
        }
        out.gs[b] = acc;
    }
    float dqh[4];
    quat_to_rot_bwd(qw, qx, qy, qz, dR, dqh);
    const float proj = qw * dqh[0] + qx * dqh[1] + qy * dqh[2] + qz * dqh[3];   // through q / |q|
    out.gq[0] = (dqh[0] - qw * proj) * qn; out.gq[1] = (dqh[1] - qx * proj) * qn;
    out.gq[2] = (dqh[2] - qy * proj) * qn; out.gq[3] = (dqh[3] - qz * proj) * qn;
    const M3 dD = matmul_nt(dJ, in.G);               // J = D G
    const V3 g3 = v3(dD.m[0], dD.m[3], dD.m[6]);     // column 0 -> x3 - x0
    const V3 g2 = v3(dD.m[1], dD.m[4], dD.m[7]);     // column 1 -> x2 - x0
    const V3 g1 = v3(dD.m[2], dD.m[5], dD.m[8]);     // column 2 -> x1 - x0
    out.gx0 = in.bary[0] * gm - (g1 + g2 + g3);
    out.gx1 = in.bary[1] * gm + g1;
    out.gx2 = in.bary[2] * gm + g2;
    out.gx3 = in.bary[3] * gm + g3;
}

// D6 FEM energy of one tet (lib/cage.py:349-361)
D3GA_HD float det3(const M3 &F) {
    const float *f = F.m;
    return f[0] * (f[4] * f[8] - f[5] * f[7]) - f[1] * (f[3] * f[8] - f[5] * f[6]) + f[2] * (f[3] * f[7] - f[4] * f[6]);
}
D3GA_HD float fem_energy_fwd(V3 x0, V3 x1, V3 x2, V3 x3, const M3 &Dn_inv) {
    const M3 F = matmul(tet_edges(x0, x1, x2, x3), Dn_inv);
    float fro = 0.f;
    for (int i = 0; i < 9; ++i) fro += F.m[i] * F.m[i];
    const float d = det3(F) - 1.f;
    return 0.5f * d * d + 0.5f * (fro - 3.f);
}
D3GA_HD void fem_energy_bwd(V3 x0, V3 x1, V3 x2, V3 x3, const M3 &Dn_inv, float g, V3 gx[4]) {
    const M3 F = matmul(tet_edges(x0, x1, x2, x3), Dn_inv);
    const float *f = F.m;
    const float d = det3(F) - 1.f;
    M3 dF;   // cofactor matrix * d + F
    dF.m[0] = g * (d * (f[4] * f[8] - f[5] * f[7]) + f[0]);
    dF.m[1] = g * (d * (f[5] * f[6] - f[3] * f[8]) + f[1]);
    dF.m[2] = g * (d * (f[3] * f[7] - f[4] * f[6]) + f[2]);
    dF.m[3] = g * (d * (f[2] * f[7] - f[1] * f[8]) + f[3]);
    dF.m[4] = g * (d * (f[0] * f[8] - f[2] * f[6]) + f[4]);
    dF.m[5] = g * (d * (f[1] * f[6] - f[0] * f[7]) + f[5]);
    dF.m[6] = g * (d * (f[1] * f[5] - f[2] * f[4]) + f[6]);
    dF.m[7] = g * (d * (f[2] * f[3] - f[0] * f[5]) + f[7]);
    dF.m[8] = g * (d * (f[0] * f[4] - f[1] * f[3]) + f[8]);
    const M3 dD = matmul_nt(dF, Dn_inv);
    const V3 g3 = v3(dD.m[0], dD.m[3], dD.m[6]), g2 = v3(dD.m[1], dD.m[4], dD.m[7]), g1 = v3(dD.m[2], dD.m[5], dD.m[8]);
    gx[0] = -1.f * (g1 + g2 + g3); gx[1] = g1; gx[2] = g2; gx[3] = g3;
}

// ---------------------------------------------------------------------------------------------------------
// R1 per-Gaussian projection (3DGS preprocess)
// ---------------------------------------------------------------------------------------------------------
D3GA_HD V3 xform_point(const float *m, V3 p) {
    D3GA_NO_CONTRACT
    return v3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
              m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
D3GA_HD float xform_w(const float *m, V3 p) {
    D3GA_NO_CONTRACT return m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]; }

// 3D covariance from scale & rotation as the rasterizer defines it: quaternion NOT normalised
D3GA_HD void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float c6[6]) {
    D3GA_NO_CONTRACT
    M3 L = quat_to_rot(q[0], q[1], q[2], q[3]);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) L.m[3 * a + b] *= mod * s[b];
    const M3 C = matmul_nt(L, L);
    c6[0] = C.m[0]; c6[1] = C.m[1]; c6[2] = C.m[2]; c6[3] = C.m[4]; c6[4] = C.m[5]; c6[5] = C.m[8];
}
D3GA_HD void cov3d_from_scale_rot_bwd(const float s[3], float mod, const float q[4], const float g6[6], float gs[3],
                                      float gq[4]) {
    const M3 R = quat_to_rot(q[0], q[1], q[2], q[3]);
    M3 L = R;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) L.m[3 * a + b] *= mod * s[b];
    M3 Gs;
    Gs.m[0] = 2.f * g6[0]; Gs.m[1] = g6[1];       Gs.m[2] = g6[2];
    Gs.m[3] = g6[1];       Gs.m[4] = 2.f * g6[3]; Gs.m[5] = g6[4];
    Gs.m[6] = g6[2];       Gs.m[7] = g6[4];       Gs.m[8] = 2.f * g6[5];
    const M3 dL = matmul(Gs, L);
    M3 dR;
    for (int b = 0; b < 3; ++b) {
        float acc = 0.f;
        for (int a = 0; a < 3; ++a) {
            acc += dL.m[3 * a + b] * R.m[3 * a + b];
            dR.m[3 * a + b] = dL.m[3 * a + b] * mod * s[b];
        }
        gs[b] = acc * mod;
    }
    quat_to_rot_bwd(q[0], q[1], q[2], q[3], dR, gq);
}

// EWA projection matrix T = Jac * Wrot (2x3) at view-space point t; returns t with the clamped x,y
struct Ewa {
    float T[6];
    V3 t;
    bool clamp_x, clamp_y;
};
D3GA_HD Ewa ewa_matrix(const float *view, V3 mean, float fx, float fy, float tanfovx, float tanfovy) {
    D3GA_NO_CONTRACT
    Ewa e;
    V3 t = xform_point(view, mean);
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    e.clamp_x = (txtz < -limx) || (txtz > limx);
    e.clamp_y = (tytz < -limy) || (tytz > limy);
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    e.t = t;
    const float j00 = fx / t.z, j02 = -(fx * t.x) / (t.z * t.z);
    const float j11 = fy / t.z, j12 = -(fy * t.y) / (t.z * t.z);
    for (int k = 0; k < 3; ++k) {                // Wrot[j][k] = view[4k + j]
        e.T[k] = j00 * view[4 * k] + j02 * view[4 * k + 2];
        e.T[3 + k] = j11 * view[4 * k + 1] + j12 * view[4 * k + 2];
    }
    return e;
}
// cov2D = T S T^T -> (a, b, c) before dilation; TS returned for the backward
D3GA_HD void cov2d(const float T[6], const float c6[6], float TS[6], float &a, float &b, float &c) {
    D3GA_NO_CONTRACT
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    for (int r = 0; r < 2; ++r)
        for (int k = 0; k < 3; ++k) TS[3 * r + k] = T[3 * r] * S[k] + T[3 * r + 1] * S[3 + k] + T[3 * r + 2] * S[6 + k];
    a = TS[0] * T[0] + TS[1] * T[1] + TS[2] * T[2];
    b = TS[0] * T[3] + TS[1] * T[4] + TS[2] * T[5];
    c = TS[3] * T[3] + TS[4] * T[4] + TS[5] * T[5];
}

struct Splat {
    bool visible;
    float depth;
    float px, py;          // pixel-space centre
    float conic[3];
    int radius;
    int rect[4];           // tile rectangle [minx, miny, maxx, maxy)
    float aa;              // antialiasing: sqrt(max(2.5e-5, det(cov2D) / det(cov2D + 0.3 I))), the factor on the opacity (1: off)
};

D3GA_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

D3GA_HD void tile_rect(float px, float py, float radius, int gx, int gy, int rect[4]) {
    D3GA_NO_CONTRACT
    rect[0] = clampi((int)((px - radius) / kTile), 0, gx);
    rect[1] = clampi((int)((py - radius) / kTile), 0, gy);
    rect[2] = clampi((int)((px + radius + kTile - 1) / kTile), 0, gx);
    rect[3] = clampi((int)((py + radius + kTile - 1) / kTile), 0, gy);
}

constexpr float kAaFloor = 0.000025f;   // [UPSTREAM-RECALL] branch dr_aa: h_convolution_scaling = sqrt(max(0.000025, det_cov / det_cov_plus_h_cov))
D3GA_HD Splat project_gaussian(V3 mean, const float c6[6], const float *view, const float *proj, int W, int H,
                               float tanfovx, float tanfovy, bool antialiasing = false) {
    D3GA_NO_CONTRACT
    Splat s;
    s.aa = 1.0f;
    s.visible = false; s.radius = 0; s.depth = 0.f; s.px = s.py = 0.f;
    s.conic[0] = s.conic[1] = s.conic[2] = 0.f;
    s.rect[0] = s.rect[1] = s.rect[2] = s.rect[3] = 0;
    const V3 pv = xform_point(view, mean);
    if (!(pv.z > kNear)) return s;        // z <= 0.2 is culled; so is a NaN position (it would poison the depth keys)
    const V3 ph = xform_point(proj, mean);
    const float pw = 1.0f / (xform_w(proj, mean) + 0.0000001f);
    const float fx = W / (2.0f * tanfovx), fy = H / (2.0f * tanfovy);
    const Ewa e = ewa_matrix(view, mean, fx, fy, tanfovx, tanfovy);
    float TS[6], a, b, c;
    cov2d(e.T, c6, TS, a, b, c);
    const float det0 = a * c - b * b;     // before the dilation: the antialiasing factor compensates the energy the dilation adds
    a += kDilate; c += kDilate;
    const float det = a * c - b * b;
    if (det == 0.0f) return s;
    if (antialiasing) s.aa = sqrtf(fmaxf(kAaFloor, det0 / det));
    const float inv = 1.0f / det;
    const float mid = 0.5f * (a + c);
    const float root = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float radius = ceilf(3.0f * sqrtf(fmaxf(mid + root, mid - root)));
    if (!(radius == radius) || !(ph.x * pw == ph.x * pw) || !(ph.y * pw == ph.y * pw)) return s;   // NaN covariance / projection: culled
    const float px = ((ph.x * pw + 1.0f) * W - 1.0f) * 0.5f;
    const float py = ((ph.y * pw + 1.0f) * H - 1.0f) * 0.5f;
    tile_rect(px, py, radius, (W + kTile - 1) / kTile, (H + kTile - 1) / kTile, s.rect);
    if ((s.rect[2] - s.rect[0]) * (s.rect[3] - s.rect[1]) == 0) return s;
    s.visible = true;
    s.depth = pv.z;
    s.px = px; s.py = py;
    s.conic[0] = c * inv; s.conic[1] = -b * inv; s.conic[2] = a * inv;
    s.radius = (int)radius;
    return s;
}

// ---------------------------------------------------------------------------------------------------------
// spherical harmonics colour (constants: utils/sh_utils.py:7-24)
// ---------------------------------------------------------------------------------------------------------
constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;
constexpr float kC2_0 = 1.0925484305920792f, kC2_1 = -1.0925484305920792f, kC2_2 = 0.31539156525252005f,
                kC2_3 = -1.0925484305920792f, kC2_4 = 0.5462742152960396f;
constexpr float kC3_0 = -0.5900435899266435f, kC3_1 = 2.890611442640554f, kC3_2 = -0.4570457994644658f,
                kC3_3 = 0.3731763325901154f, kC3_4 = -0.4570457994644658f, kC3_5 = 1.445305721320277f,
                kC3_6 = -0.5900435899266435f;

// basis[k], k < (deg+1)^2, for unit direction (x,y,z)
D3GA_HD void sh_basis(int deg, float x, float y, float z, float B[16]) {
    // no contraction: the basis values must not depend on what the surrounding code shares with them (round 5: the forward that
    // also leaves d(colour)/d(direction) reuses xx, yy, ... -- with contraction its colours differed from the inference
    // forward's in the last ulp; an inference render and a training render of the same inputs give the same image bit for bit)
    D3GA_NO_CONTRACT
    B[0] = kC0;
    if (deg > 0) {
        B[1] = -kC1 * y; B[2] = kC1 * z; B[3] = -kC1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = kC2_0 * xy; B[5] = kC2_1 * yz; B[6] = kC2_2 * (2.f * zz - xx - yy);
            B[7] = kC2_3 * xz; B[8] = kC2_4 * (xx - yy);
            if (deg > 2) {
                B[9] = kC3_0 * y * (3.f * xx - yy); B[10] = kC3_1 * xy * z;
                B[11] = kC3_2 * y * (4.f * zz - xx - yy); B[12] = kC3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                B[13] = kC3_4 * x * (4.f * zz - xx - yy); B[14] = kC3_5 * z * (xx - yy);
                B[15] = kC3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}
// d basis / d(x,y,z)
D3GA_HD void sh_basis_grad(int deg, float x, float y, float z, float Bx[16], float By[16], float Bz[16]) {
    Bx[0] = By[0] = Bz[0] = 0.f;
    if (deg > 0) {
        Bx[1] = 0.f;  By[1] = -kC1; Bz[1] = 0.f;
        Bx[2] = 0.f;  By[2] = 0.f;  Bz[2] = kC1;
        Bx[3] = -kC1; By[3] = 0.f;  Bz[3] = 0.f;
        if (deg > 1) {
            Bx[4] = kC2_0 * y;        By[4] = kC2_0 * x;        Bz[4] = 0.f;
            Bx[5] = 0.f;              By[5] = kC2_1 * z;        Bz[5] = kC2_1 * y;
            Bx[6] = -2.f * kC2_2 * x; By[6] = -2.f * kC2_2 * y; Bz[6] = 4.f * kC2_2 * z;
            Bx[7] = kC2_3 * z;        By[7] = 0.f;              Bz[7] = kC2_3 * x;
            Bx[8] = 2.f * kC2_4 * x;  By[8] = -2.f * kC2_4 * y; Bz[8] = 0.f;
            if (deg > 2) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                Bx[9] = kC3_0 * 6.f * xy;                  By[9] = kC3_0 * 3.f * (xx - yy);            Bz[9] = 0.f;
                Bx[10] = kC3_1 * yz;                       By[10] = kC3_1 * xz;                        Bz[10] = kC3_1 * xy;
                Bx[11] = kC3_2 * -2.f * xy;                By[11] = kC3_2 * (4.f * zz - xx - 3.f * yy); Bz[11] = kC3_2 * 8.f * yz;
                Bx[12] = kC3_3 * -6.f * xz;                By[12] = kC3_3 * -6.f * yz;                 Bz[12] = kC3_3 * 3.f * (2.f * zz - xx - yy);
                Bx[13] = kC3_4 * (4.f * zz - 3.f * xx - yy); By[13] = kC3_4 * -2.f * xy;               Bz[13] = kC3_4 * 8.f * xz;
                Bx[14] = kC3_5 * 2.f * xz;                 By[14] = kC3_5 * -2.f * yz;                 Bz[14] = kC3_5 * (xx - yy);
                Bx[15] = kC3_6 * 3.f * (xx - yy);          By[15] = kC3_6 * -6.f * xy;                 Bz[15] = 0.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// R6 per-Gaussian backward pieces
// ---------------------------------------------------------------------------------------------------------
// conic gradient (dA, dB/2, dC) -> dL/dcov3D (6, accumulated with =) and dL/dmean (view-space chain, through T)
// antialiasing: g_op_scaled = dL/d(opacity x aa) and the activated opacity as stored (opacity x aa); the factor's own
// gradient dL/daa = g_op_scaled x opacity flows into the covariance through det(cov2D) / det(cov2D + 0.3 I); *aa_out = aa
D3GA_HD void cov2d_bwd(V3 mean, const float c6[6], const float *view, int W, int H, float tanfovx, float tanfovy,
                       float dA, float dBh, float dC, float g6[6], float gmean[3], bool antialiasing = false,
                       float g_op_scaled = 0.f, float op_scaled = 0.f, float *aa_out = nullptr) {
    const float fx = W / (2.0f * tanfovx), fy = H / (2.0f * tanfovy);
    const Ewa e = ewa_matrix(view, mean, fx, fy, tanfovx, tanfovy);
    const float *T = e.T;
    float TS[6], a, b, c;
    cov2d(T, c6, TS, a, b, c);
    const float a0 = a, c0 = c;
    a += kDilate; c += kDilate;
    const float denom = a * c - b * b;
    const float d2 = 1.0f / (denom * denom + 0.0000001f);
    // conic = (c, -b, a)/denom ; dBh is half of dL/dB
    float ga = d2 * (-c * c * dA + 2.f * b * c * dBh + (denom - a * c) * dC);
    float gc = d2 * (-a * a * dC + 2.f * a * b * dBh + (denom - a * c) * dA);
    float gb = d2 * 2.f * (b * c * dA - (denom + 2.f * b * b) * dBh + a * b * dC);
    if (aa_out) *aa_out = 1.0f;
    if (antialiasing) {
        // rho = (a0 c0 - b^2) / ((a0 + w)(c0 + w) - b^2), aa = sqrt(max(floor, rho));  with D = the dilated determinant:
        //   d rho / d a0 = w (w c0 + c0^2 + b^2) / D^2,  d rho / d c0 = w (w a0 + a0^2 + b^2) / D^2,  d rho / d b = -2 w b (w + a0 + c0) / D^2
        const float rho = (a0 * c0 - b * b) / denom;
        const float aa = sqrtf(fmaxf(kAaFloor, rho));
        if (aa_out) *aa_out = aa;
        const float g_aa = g_op_scaled * (op_scaled / aa);            // dL/daa = dL/d(opacity aa) x opacity
        const float g_rho = rho <= kAaFloor ? 0.f : g_aa / (2.f * aa);
        const float k = g_rho * kDilate / (denom * denom);
        ga += k * (kDilate * c0 + c0 * c0 + b * b);
        gc += k * (kDilate * a0 + a0 * a0 + b * b);
        gb += -2.f * k * b * (kDilate + a0 + c0);
    }
    g6[0] = T[0] * T[0] * ga + T[0] * T[3] * gb + T[3] * T[3] * gc;
    g6[3] = T[1] * T[1] * ga + T[1] * T[4] * gb + T[4] * T[4] * gc;
    g6[5] = T[2] * T[2] * ga + T[2] * T[5] * gb + T[5] * T[5] * gc;
    g6[1] = 2.f * T[0] * T[1] * ga + (T[0] * T[4] + T[1] * T[3]) * gb + 2.f * T[3] * T[4] * gc;
    g6[2] = 2.f * T[0] * T[2] * ga + (T[0] * T[5] + T[2] * T[3]) * gb + 2.f * T[3] * T[5] * gc;
    g6[4] = 2.f * T[1] * T[2] * ga + (T[1] * T[5] + T[2] * T[4]) * gb + 2.f * T[4] * T[5] * gc;
    float dT[6];
    for (int k = 0; k < 3; ++k) {
        dT[k] = 2.f * TS[k] * ga + TS[3 + k] * gb;
        dT[3 + k] = 2.f * TS[3 + k] * gc + TS[k] * gb;
    }
    float dj00 = 0.f, dj02 = 0.f, dj11 = 0.f, dj12 = 0.f;
    for (int k = 0; k < 3; ++k) {
        dj00 += view[4 * k] * dT[k];
        dj02 += view[4 * k + 2] * dT[k];
        dj11 += view[4 * k + 1] * dT[3 + k];
        dj12 += view[4 * k + 2] * dT[3 + k];
    }
    const float iz = 1.f / e.t.z, iz2 = iz * iz, iz3 = iz2 * iz;
    const float dtx = e.clamp_x ? 0.f : -fx * iz2 * dj02;
    const float dty = e.clamp_y ? 0.f : -fy * iz2 * dj12;
    const float dtz = -fx * iz2 * dj00 - fy * iz2 * dj11 + 2.f * fx * e.t.x * iz3 * dj02 + 2.f * fy * e.t.y * iz3 * dj12;
    gmean[0] = view[0] * dtx + view[1] * dty + view[2] * dtz;
    gmean[1] = view[4] * dtx + view[5] * dty + view[6] * dtz;
    gmean[2] = view[8] * dtx + view[9] * dty + view[10] * dtz;
}

// screen-space mean gradient (in the compositing kernel's NDC-scaled units) -> dL/dmean (added)
D3GA_HD void project_bwd(V3 m, const float *p, float gx, float gy, float gmean[3]) {
    const float mw = 1.0f / (xform_w(p, m) + 0.0000001f);
    const float mul1 = (p[0] * m.x + p[4] * m.y + p[8] * m.z + p[12]) * mw * mw;
    const float mul2 = (p[1] * m.x + p[5] * m.y + p[9] * m.z + p[13]) * mw * mw;
    gmean[0] += (p[0] * mw - p[3] * mul1) * gx + (p[1] * mw - p[3] * mul2) * gy;
    gmean[1] += (p[4] * mw - p[7] * mul1) * gx + (p[5] * mw - p[7] * mul2) * gy;
    gmean[2] += (p[8] * mw - p[11] * mul1) * gx + (p[9] * mw - p[11] * mul2) * gy;
}

// gradient of normalize(v) applied to g
D3GA_HD V3 normalize_bwd(V3 v, V3 g) {
    const float l2 = dot(v, v);
    const float inv3 = 1.0f / (l2 * sqrtf(l2));
    return v3(((l2 - v.x * v.x) * g.x - v.y * v.x * g.y - v.z * v.x * g.z) * inv3,
              (-v.x * v.y * g.x + (l2 - v.y * v.y) * g.y - v.z * v.y * g.z) * inv3,
              (-v.x * v.z * g.x - v.y * v.z * g.y + (l2 - v.z * v.z) * g.z) * inv3);
}

// ---------------------------------------------------------------------------------------------------------
// R4/R5 one Gaussian against one pixel
// ---------------------------------------------------------------------------------------------------------
// returns false if the Gaussian does not touch the pixel; otherwise alpha (clamped) and G = exp(power)
D3GA_HD bool splat_alpha(float dx, float dy, float ca, float cb, float cc, float opacity, float &alpha, float &G) {
    const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
    if (power > 0.0f) return false;
    G = __expf(power);
    alpha = fminf(kAlphaMax, opacity * G);
    return alpha >= kAlphaMin;
}

}  // namespace d3ga

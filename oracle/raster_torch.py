"""Oracle: dense, autograd-differentiable restatement of the 3DGS tile rasterizer (CPU, torch).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.         *** PARITY UNPINNED ***

The reference calls the un-vendored CUDA package diff_gaussian_rasterization (graphdeco-inria,
branch dr_aa, SHA unpinned: /root/reference/.gitmodules:9-12) at renderer.py:79-141 and ships no
test or golden vector for it.  This file restates the published algorithm (Kerbl et al., "3D Gaussian
Splatting", SIGGRAPH 2023, sec. 4-6 and App. A-C) with the constants listed in SURVEY.md sec. 8a R1-R6:

  R1 preprocess: cull z_view <= 0.2; p_proj = P p / (w + 1e-7); EWA cov2D = (J W) Sigma (J W)^T with the
     view-space x/z, y/z clamped to +-1.3 tanfov; +0.3 on the diagonal (antialiasing off); conic; radius =
     ceil(3 sqrt(lambda_max)), lambda from mid +- sqrt(max(0.1, mid^2 - det)); pixel = ((ndc+1) S - 1)/2;
     16x16 tile rectangle; colour = precomputed or clamp(SH(deg, normalize(p - campos)) + 0.5, >= 0).
  R2/R3 ordering: per tile by (depth, Gaussian index) ascending.
  R4 compositing front-to-back: power = -1/2 (A dx^2 + C dy^2) - B dx dy; skip power > 0;
     alpha = min(0.99, o exp(power)); skip alpha < 1/255; stop BEFORE blending once T (1-alpha) < 1e-4;
     out = C + T bg.
  R5/R6 backward = derivative of the above, with these hand-written deviations restated explicitly:
     (i) the 0.99 alpha clamp passes the gradient straight through; (ii) when the x/z (y/z) clamp is
     active the clamped value is treated as a constant.  Everything else is plain autograd.

Dense O(pixels x Gaussians): use for small scenes only.  Works in float32 or float64.
"""
import math
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)
TILE = 16


def eval_sh(deg, sh, dirs):
    """sh (P,M,3), dirs (P,3) unit -> (P,3) before the +0.5/clamp.  Basis constants: utils/sh_utils.py:7-24."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
               + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
               + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
               + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
               + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res


def _sym6_to_mat(c):
    xx, xy, xz, yy, yz, zz = c.unbind(dim=1)
    return torch.stack([torch.stack([xx, xy, xz], 1), torch.stack([xy, yy, yz], 1),
                        torch.stack([xz, yz, zz], 1)], dim=1)


def _quat_rot_nonorm(q):
    """3DGS rasterizer's computeCov3D uses the quaternion as given (no normalisation), order wxyz."""
    r, x, y, z = q.unbind(dim=1)
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], 1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], 1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)], dim=1)


def preprocess(means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H,
               cov3D_precomp=None, scales=None, rotations=None, scale_modifier=1.0,
               shs=None, colors_precomp=None, sh_degree=0, antialiasing=False):
    """Per-Gaussian stage R1.  viewmatrix/projmatrix are the reference's TRANSPOSED 4x4 (row-vector) matrices.
    antialiasing (branch dr_aa, [UPSTREAM-RECALL]): opacity x sqrt(max(0.000025, det(cov2D) / det(cov2D + 0.3 I)))."""
    dt = means3D.dtype
    P = means3D.shape[0]
    V = viewmatrix.to(dt)
    PM = projmatrix.to(dt)
    hom = torch.cat([means3D, torch.ones(P, 1, dtype=dt)], dim=1)
    p_view = hom @ V[:, :3]                                   # (P,3)
    p_hom = hom @ PM                                          # (P,4)
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    ndc = p_hom[:, :3] * p_w[:, None]
    depth = p_view[:, 2]
    in_front = depth > 0.2

    if cov3D_precomp is not None:
        Sigma = _sym6_to_mat(cov3D_precomp)
    else:
        R = _quat_rot_nonorm(rotations)
        M = R * (scale_modifier * scales)[:, None, :]
        Sigma = M @ M.transpose(1, 2)

    focal_x = W / (2.0 * tanfovx)
    focal_y = H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tz).detach(), p_view[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tz).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    Jm = torch.stack([torch.stack([focal_x / tz, zero, -(focal_x * tx) / (tz * tz)], 1),
                      torch.stack([zero, focal_y / tz, -(focal_y * ty) / (tz * tz)], 1)], dim=1)   # (P,2,3)
    Wr = V[:3, :3].transpose(0, 1)                                        # world->view rotation
    Tm = Jm @ Wr                                                          # (P,2,3)
    cov2 = Tm @ Sigma @ Tm.transpose(1, 2)
    det0 = cov2[:, 0, 0] * cov2[:, 1, 1] - cov2[:, 0, 1] * cov2[:, 0, 1]
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    det_ok = det != 0
    det_safe = torch.where(det_ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], dim=1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def trunc(v):  # C cast (int) truncation toward zero
        return torch.trunc(v.detach()).to(torch.int64)
    rminx = trunc((px - radius) / TILE).clamp(0, gx)
    rminy = trunc((py - radius) / TILE).clamp(0, gy)
    rmaxx = trunc((px + radius + TILE - 1) / TILE).clamp(0, gx)
    rmaxy = trunc((py + radius + TILE - 1) / TILE).clamp(0, gy)
    touched = (rmaxx - rminx) * (rmaxy - rminy)
    valid = in_front & det_ok & (touched > 0)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos.to(dt)[None]
        d = d / torch.sqrt((d * d).sum(1, keepdim=True))
        rgb = torch.clamp(eval_sh(sh_degree, shs, d) + 0.5, min=0.0)

    opac = opacities.reshape(-1)
    if antialiasing:
        opac = opac * torch.sqrt(torch.clamp(det0 / det_safe, min=0.000025))
    return dict(xy=torch.stack([px, py], 1), depth=depth, conic=conic, opacity=opac, rgb=rgb,
                radii=torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32),
                rect=(rminx, rminy, rmaxx, rmaxy), valid=valid, tiles_touched=torch.where(valid, touched, 0 * touched))


def composite(pre, bg, W, H, pixel_chunk=4096, return_aux=False):
    """Stages R2-R4, dense.  Returns color (3,H,W) [, final_T (H,W), n_contrib (H,W), invdepth (H,W)]."""
    dt = pre["xy"].dtype
    valid = pre["valid"]
    idx = torch.nonzero(valid).reshape(-1)
    depth = pre["depth"][idx].detach()
    # order by (depth, index): stable sort of an index-ordered list
    order = torch.sort(depth.to(torch.float32), stable=True)[1]
    idx = idx[order]
    N = idx.numel()
    xy = pre["xy"][idx]
    conic = pre["conic"][idx]
    opac = pre["opacity"][idx]
    rgb = pre["rgb"][idx]
    invd = 1.0 / pre["depth"][idx]           # differentiable: branch dr_aa back-propagates a loss on the inverse-depth image
    rminx, rminy, rmaxx, rmaxy = [r[idx] for r in pre["rect"]]

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    out_c, out_T, out_n, out_d = [], [], [], []
    for s in range(0, W * H, pixel_chunk):
        px, py = xs[s:s + pixel_chunk], ys[s:s + pixel_chunk]
        n = px.numel()
        if N == 0:
            out_c.append(bg.to(dt)[None].expand(n, 3))
            out_T.append(torch.ones(n, dtype=dt))
            out_n.append(torch.zeros(n, dtype=torch.int64))
            out_d.append(torch.zeros(n, dtype=dt))
            continue
        tx, ty = (px // TILE)[:, None], (py // TILE)[:, None]
        member = (tx >= rminx[None]) & (tx < rmaxx[None]) & (ty >= rminy[None]) & (ty < rmaxy[None])
        dx = xy[None, :, 0] - px[:, None].to(dt)
        dy = xy[None, :, 1] - py[:, None].to(dt)
        power = -0.5 * (conic[None, :, 0] * dx * dx + conic[None, :, 2] * dy * dy) - conic[None, :, 1] * dx * dy
        a_raw = opac[None] * torch.exp(power)
        alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()      # clamp, gradient straight through
        use = member & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
        a = torch.where(use, alpha, torch.zeros_like(alpha))
        T_incl = torch.cumprod(1.0 - a, dim=1)
        keep = T_incl.detach() >= 0.0001                                      # stop before blending
        a = torch.where(keep, a, torch.zeros_like(a))
        T_incl = torch.cumprod(1.0 - a, dim=1)
        T_excl = torch.cat([torch.ones(n, 1, dtype=dt), T_incl[:, :-1]], dim=1)
        w = a * T_excl
        col = w @ rgb + T_incl[:, -1:] * bg.to(dt)[None]
        out_c.append(col)
        out_T.append(T_incl[:, -1])
        contrib = (use & keep)
        # n_contrib = 1-based position IN THE TILE LIST of the last contributor
        pos_in_tile = torch.cumsum(member.to(torch.int64), dim=1)
        last = torch.where(contrib, pos_in_tile, torch.zeros_like(pos_in_tile)).max(dim=1)[0]
        out_n.append(last)
        out_d.append(w @ invd)
    color = torch.cat(out_c, 0).reshape(H, W, 3).permute(2, 0, 1).contiguous()
    if not return_aux:
        return color
    return (color, torch.cat(out_T).reshape(H, W), torch.cat(out_n).reshape(H, W), torch.cat(out_d).reshape(H, W))


def rasterize(means3D, opacities, bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H,
              cov3D_precomp=None, scales=None, rotations=None, scale_modifier=1.0,
              shs=None, colors_precomp=None, sh_degree=0, pixel_chunk=4096, return_aux=False, antialiasing=False):
    pre = preprocess(means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H,
                     cov3D_precomp, scales, rotations, scale_modifier, shs, colors_precomp, sh_degree, antialiasing)
    out = composite(pre, bg, W, H, pixel_chunk, return_aux)
    if return_aux:
        return out + (pre,)
    return out, pre["radii"]

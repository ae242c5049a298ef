"""ctypes front-end of the C oracle rasterizer (oracle/raster_c/raster_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED (no reference rasterizer source).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "..", "_build", "libraster_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "raster_oracle.c")
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", _SO, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.ro_forward.restype = ctypes.c_void_p
        L.ro_num_rendered.restype = ctypes.c_int64
        L.ro_num_rendered.argtypes = [ctypes.c_void_p]
        L.ro_free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def set_threads(n):
    lib().ro_set_threads(ctypes.c_int(int(n)))


def max_threads():
    return int(lib().ro_max_threads())


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Context:
    def __init__(self, handle, keep):
        self.handle = handle
        self.keep = keep

    def __del__(self):
        try:
            if self.handle and _lib is not None:
                _lib.ro_free(ctypes.c_void_p(self.handle))
        except Exception:
            pass
        self.handle = None


def forward(means3D, opacities, bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H,
            cov3D_precomp=None, scales=None, rotations=None, scale_modifier=1.0,
            shs=None, colors_precomp=None, sh_degree=0, antialiasing=False):
    """numpy in / numpy out.  Returns (color (3,H,W), radii (P,), invdepth (H,W), ctx).
    antialiasing: branch dr_aa's opacity compensation of the 0.3 px dilation [UPSTREAM-RECALL] (ro_set_antialiasing)."""
    L = lib()
    L.ro_set_antialiasing(ctypes.c_int(1 if antialiasing else 0))
    means3D = _f(means3D)
    P = means3D.shape[0]
    opacities, bg = _f(opacities).reshape(-1), _f(bg)
    viewmatrix, projmatrix, campos = _f(viewmatrix).reshape(-1), _f(projmatrix).reshape(-1), _f(campos)
    cov3D_precomp, scales, rotations, shs, colors_precomp = map(_f, (cov3D_precomp, scales, rotations, shs,
                                                                    colors_precomp))
    M = shs.shape[1] if shs is not None else 0
    color = np.zeros((3, H, W), np.float32)
    radii = np.zeros(P, np.int32)
    invd = np.zeros((H, W), np.float32)
    h = L.ro_forward(ctypes.c_int(P), ctypes.c_int(M), ctypes.c_int(sh_degree), ctypes.c_int(W), ctypes.c_int(H),
                     _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales), _p(rotations),
                     ctypes.c_float(scale_modifier), _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), _p(campos),
                     ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), _p(bg), _p(color), _p(radii), _p(invd))
    L.ro_set_antialiasing(ctypes.c_int(0))
    ctx = Context(h, dict(means3D=means3D, shs=shs, scales=scales, rotations=rotations, P=P, M=M, W=W, H=H,
                          has_sh=shs is not None, from_sr=cov3D_precomp is None))
    return color, radii, invd, ctx


def num_rendered(ctx):
    return int(lib().ro_num_rendered(ctypes.c_void_p(ctx.handle)))


def tile_lists(ctx):
    k = ctx.keep
    tiles = ((k["W"] + 15) // 16) * ((k["H"] + 15) // 16)
    D = num_rendered(ctx)
    start = np.zeros(tiles + 1, np.int64)
    pl = np.zeros(max(D, 1), np.int32)
    lib().ro_get_tile_list(ctypes.c_void_p(ctx.handle), _p(start), _p(pl))
    return start, pl[:D]


def geom(ctx):
    k = ctx.keep
    P, W, H = k["P"], k["W"], k["H"]
    depth = np.zeros(P, np.float32); xy = np.zeros((P, 2), np.float32); co = np.zeros((P, 4), np.float32)
    rgb = np.zeros((P, 3), np.float32); nc = np.zeros((H, W), np.int32); fT = np.zeros((H, W), np.float32)
    lib().ro_get_geom(ctypes.c_void_p(ctx.handle), _p(depth), _p(xy), _p(co), _p(rgb), _p(nc), _p(fT))
    return dict(depth=depth, xy=xy, conic_o=co, rgb=rgb, n_contrib=nc, final_T=fT)


def margins(ctx, eps_alpha=1e-5, eps_T=1e-3):
    """(pixel mask (H,W) bool, Gaussian mask (P,) bool): where a discontinuous decision of the algorithm (alpha < 1/255,
    T(1-alpha) < 1e-4) sits within float32 implementation noise of its threshold (ro_marginal in raster_oracle.c).
    eps_alpha: two f32 evaluations of alpha differ by the rounding of the quadratic form (its terms cancel, |term| up to
    ~20 => ~1e-6 absolute in the exponent) plus the exp itself; eps_T: test_T is a product of up to a few hundred
    (1 - alpha) factors."""
    k = ctx.keep
    pix = np.zeros((k["H"], k["W"]), np.uint8)
    gs = np.zeros(max(k["P"], 1), np.uint8)
    L = lib()
    L.ro_marginal.restype = ctypes.c_int64
    L.ro_marginal(ctypes.c_void_p(ctx.handle), ctypes.c_float(eps_alpha), ctypes.c_float(eps_T), _p(pix), _p(gs))
    return pix.astype(bool), gs[:k["P"]].astype(bool)


def set_termination(ctx, final_T, n_contrib):
    """Shared decisions: replace the context's per-pixel termination (final_T (H,W) f32, n_contrib (H,W) i32) with the ones
    another implementation's forward produced; `backward` then walks exactly those entries (ro_set_termination)."""
    k = ctx.keep
    fT = np.ascontiguousarray(final_T, dtype=np.float32).reshape(k["H"], k["W"])
    nc = np.ascontiguousarray(n_contrib, dtype=np.int32).reshape(k["H"], k["W"])
    lib().ro_set_termination(ctypes.c_void_p(ctx.handle), _p(fT), _p(nc))


def alpha_marginal(ctx, eps_alpha=1e-5):
    """(P,) bool: Gaussians with an alpha within eps_alpha (relative) of 1/255 on some pixel, at a list position the backward
    walks there -- the only discontinuous decision left once the termination is shared (ro_alpha_marginal)."""
    k = ctx.keep
    gs = np.zeros(max(k["P"], 1), np.uint8)
    L = lib()
    L.ro_alpha_marginal.restype = ctypes.c_int64
    L.ro_alpha_marginal(ctypes.c_void_p(ctx.handle), ctypes.c_float(eps_alpha), _p(gs))
    return gs[:k["P"]].astype(bool)


def alpha_band_pairs(ctx, eps_alpha=1e-4):
    """(gid (n,) i32, pix (n,) i32 = py * W + px): every (Gaussian, pixel) pair the backward walks whose alpha lies within
    eps_alpha of 1/255 (ro_alpha_band_pairs), sorted by (pix, gid)."""
    L = lib()
    L.ro_alpha_band_pairs.restype = ctypes.c_int64
    cap = 1 << 16
    while True:
        gid, pix = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n = int(L.ro_alpha_band_pairs(ctypes.c_void_p(ctx.handle), ctypes.c_float(eps_alpha), ctypes.c_int64(cap), _p(gid), _p(pix)))
        if n <= cap:
            break
        cap = n
    gid, pix = gid[:n], pix[:n]
    order = np.lexsort((gid, pix))
    return np.ascontiguousarray(gid[order]), np.ascontiguousarray(pix[order])


def set_alpha_overrides(ctx, gid, pix, ok):
    """Shared alpha decisions: `backward` takes ok[i] (blended or not) for pair (gid[i], pix[i]) instead of its own
    alpha >= 1/255 test (ro_set_alpha_overrides; pairs sorted by (pix, gid) as alpha_band_pairs returns them)."""
    gid = np.ascontiguousarray(gid, np.int32); pix = np.ascontiguousarray(pix, np.int32); ok = np.ascontiguousarray(ok, np.uint8)
    assert len(gid) == len(pix) == len(ok)
    if len(gid) > 1:
        key = pix.astype(np.int64) * (1 << 31) + gid
        assert (np.diff(key) > 0).all(), "pairs must be unique and sorted by (pix, gid)"
    L = lib()
    L.ro_set_alpha_overrides.restype = None
    L.ro_set_alpha_overrides(ctypes.c_void_p(ctx.handle), ctypes.c_int64(len(gid)), _p(gid), _p(pix), _p(ok))


def backward(ctx, dL_dpix, sum_noise=None, dL_dinvdepth=None):
    """Returns dict of numpy grads: means3D, means2D(P,3), shs|colors, opacities(P,1), scales, rotations, cov3D.
    dL_dinvdepth (H,W): gradient of the inverse-depth image (branch dr_aa; ro_set_invdepth_grad), added to dL_dpix's.
    sum_noise=(gamma, pattern): conditioning probe -- every per-Gaussian pixel sum is moved by +- gamma x (sum of the absolute
    values of its terms) before the chain behind it runs (ro_set_sum_noise); the difference to the plain result bounds what the
    error of a float32 sum (gamma ~ depth x eps) does to each output element."""
    L = lib()
    L.ro_set_sum_noise.argtypes = [ctypes.c_double, ctypes.c_int]
    L.ro_set_sum_noise.restype = None
    if sum_noise is not None:
        L.ro_set_sum_noise(float(sum_noise[0]), int(sum_noise[1]))
        try:
            return backward(ctx, dL_dpix, dL_dinvdepth=dL_dinvdepth)
        finally:
            L.ro_set_sum_noise(0.0, 0)
    k = ctx.keep
    P, M = k["P"], k["M"]
    dL_dpix = _f(dL_dpix)
    g = dict(means3D=np.zeros((P, 3), np.float32), means2D=np.zeros((P, 3), np.float32),
             opacities=np.zeros((P, 1), np.float32), cov3D=np.zeros((P, 6), np.float32))
    g["shs"] = np.zeros((P, M, 3), np.float32) if k["has_sh"] else None
    g["colors"] = None if k["has_sh"] else np.zeros((P, 3), np.float32)
    g["scales"] = np.zeros((P, 3), np.float32) if k["from_sr"] else None
    g["rotations"] = np.zeros((P, 4), np.float32) if k["from_sr"] else None
    gd = None if dL_dinvdepth is None else np.ascontiguousarray(dL_dinvdepth, dtype=np.float32).reshape(k["H"], k["W"])
    lib().ro_set_invdepth_grad.restype = None
    lib().ro_set_invdepth_grad(_p(gd))
    try:
        lib().ro_backward(ctypes.c_void_p(ctx.handle), _p(k["means3D"]), _p(k["shs"]), _p(k["scales"]), _p(k["rotations"]),
                          _p(dL_dpix), _p(g["means3D"]), _p(g["means2D"]), _p(g["shs"]), _p(g["colors"]),
                          _p(g["opacities"]), _p(g["scales"]), _p(g["rotations"]), _p(g["cov3D"]))
    finally:
        lib().ro_set_invdepth_grad(None)
    return g

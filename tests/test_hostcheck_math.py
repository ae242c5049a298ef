"""The product's per-element arithmetic headers (d3ga_math.h, raster_pre_body.h -- the very code the gfx950
kernels execute per Gaussian), compiled for the host, against the oracle.  Runs without a GPU."""
import ctypes

import numpy as np
import torch

from conftest import ptr
from oracle import deform as od
from oracle import raster_c as rc
from oracle import raster_torch as rt
from util import rel_err, scene_inputs

from d3ga_amd._lib import RasterParams


def _np(t):
    return np.ascontiguousarray(t.detach().numpy())


def test_deform_forward_backward_vs_golden(hostcheck, golden):
    for name in ("deform_case0.npz", "deform_case1.npz"):
        g = golden(name)
        P, V = g["tetra_id"].shape[0], g["tetpoints"].shape[0]
        tets, tid = g["tetras"].astype(np.int32), g["tetra_id"].astype(np.int32)
        tp, barys = g["tetpoints"], g["canon_barys"]
        cg = np.ascontiguousarray(g["canonical_gradient"])
        scales, rots = g["scales"], g["rotations"]
        means, cov6 = np.zeros((P, 3), np.float32), np.zeros((P, 6), np.float32)
        hostcheck.hc_deform_fwd(P, ptr(tp), ptr(tets), ptr(tid), ptr(barys), ptr(cg), ptr(scales), ptr(rots), ptr(means),
                                ptr(cov6))
        np.testing.assert_allclose(means, g["means3D"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(cov6, g["cov3D_precomp"], rtol=5e-4, atol=1e-10)
        # backward w.r.t. (tetpoints, barys, activated scales, normalised rotation) vs oracle autograd
        t = lambda a: torch.from_numpy(a).double().requires_grad_(True)
        tp_t, b_t, s_t, r_t = t(tp), t(barys), t(scales), t(rots)
        m, c = od.cage_deform(tp_t, torch.from_numpy(tets), torch.from_numpy(tid), b_t, torch.from_numpy(cg).double(),
                              s_t, r_t)
        gm, gc = g["up_grad_means"], g["up_grad_cov"]
        ((m * torch.from_numpy(gm).double()).sum() + (c * torch.from_numpy(gc).double()).sum()).backward()
        g_tp, g_b = np.zeros((V, 3), np.float32), np.zeros((P, 4), np.float32)
        g_s, g_r = np.zeros((P, 3), np.float32), np.zeros((P, 4), np.float32)
        hostcheck.hc_deform_bwd(P, V, ptr(tp), ptr(tets), ptr(tid), ptr(barys), ptr(cg), ptr(scales), ptr(rots), ptr(gm),
                                ptr(gc), ptr(g_tp), ptr(g_b), ptr(g_s), ptr(g_r))
        assert rel_err(g_tp, _np(tp_t.grad)) < 1e-4
        assert rel_err(g_b, _np(b_t.grad)) < 1e-5
        assert rel_err(g_s, _np(s_t.grad)) < 1e-4
        assert rel_err(g_r, _np(r_t.grad)) < 1e-4
        # the reference's own gradient w.r.t. tetpoints and barys (captured by tools/gen_golden.py)
        assert rel_err(g_tp, g["grad_tetpoints"]) < 1e-3
        assert rel_err(g_b, g["grad_barys"]) < 1e-3


def test_fem_energy_vs_oracle(hostcheck, golden):
    g = golden("deform_case0.npz")
    T, V = g["tetras"].shape[0], g["tetpoints"].shape[0]
    tets, tp, Dn = g["tetras"].astype(np.int32), g["tetpoints"], np.ascontiguousarray(g["Dn_inv"])
    e = np.zeros(T, np.float32)
    hostcheck.hc_fem_fwd(T, ptr(tp), ptr(tets), ptr(Dn), ptr(e))
    tp_t = torch.from_numpy(tp).double().requires_grad_(True)
    ref = od.fem_energy(tp_t, torch.from_numpy(tets), torch.from_numpy(Dn).double())
    np.testing.assert_allclose(e, _np(ref), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(e.mean(), g["fm_energy"][0], rtol=1e-4)
    w = np.random.default_rng(0).normal(size=T).astype(np.float32)
    (ref * torch.from_numpy(w).double()).sum().backward()
    gt = np.zeros((V, 3), np.float32)
    hostcheck.hc_fem_bwd(T, V, ptr(tp), ptr(tets), ptr(Dn), ptr(w), ptr(gt))
    assert rel_err(gt, _np(tp_t.grad)) < 1e-4


def _prm(inp, M, deg, mod=1.0, antialiasing=0):
    return RasterParams(P=inp["means3D"].shape[0], M=M, sh_degree=deg, W=inp["W"], H=inp["H"],
                        tanfovx=inp["cam"]["tanfovx"], tanfovy=inp["cam"]["tanfovy"], scale_modifier=mod,
                        antialiasing=antialiasing, prefiltered=0, debug=0)


def _run_pre(hostcheck, inp, prm, shs=None, colors=None, cov=None, scales=None, rots=None):
    P = prm.P
    o = dict(depth=np.zeros(P, np.float32), xy=np.zeros((P, 2), np.float32), conic_o=np.zeros((P, 4), np.float32),
             rgb=np.zeros((P, 3), np.float32), radii=np.zeros(P, np.int32), rect=np.zeros((P, 4), np.int32),
             clamped=np.zeros(P, np.uint8), cov3D=np.zeros((P, 6), np.float32))
    hostcheck.hc_preprocess(ctypes.byref(prm), ptr(_np(inp["means3D"])), ptr(shs), ptr(colors), ptr(_np(inp["opacities"])),
                            ptr(scales), ptr(rots), ptr(cov), ptr(_np(inp["view"])), ptr(_np(inp["proj"])),
                            ptr(_np(inp["campos"])), ptr(o["depth"]), ptr(o["xy"]), ptr(o["conic_o"]), ptr(o["rgb"]),
                            ptr(o["radii"]), ptr(o["rect"]), ptr(o["clamped"]), ptr(o["cov3D"]))
    return o


def test_preprocess_forward_vs_oracles(hostcheck):
    inp = scene_inputs("T1", scale_mult=3.0)
    prm = _prm(inp, 16, 3)
    shs, cov = _np(inp["shs"]), _np(inp["cov6"])
    o = _run_pre(hostcheck, inp, prm, shs=shs, cov=cov)
    cam = inp["cam"]
    _, radii, _, ctx = rc.forward(_np(inp["means3D"]), _np(inp["opacities"]), np.ones(3, np.float32),
                                  cam["world_view_transform"], cam["full_proj_transform"], cam["camera_center"],
                                  cam["tanfovx"], cam["tanfovy"], inp["W"], inp["H"], cov3D_precomp=cov, shs=shs,
                                  sh_degree=3)
    g = rc.geom(ctx)
    np.testing.assert_array_equal(o["radii"], radii)
    vis = radii > 0
    assert vis.sum() > 100
    np.testing.assert_allclose(o["depth"][vis], g["depth"][vis], rtol=1e-6)
    np.testing.assert_allclose(o["xy"][vis], g["xy"][vis], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(o["conic_o"][vis], g["conic_o"][vis], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(o["rgb"][vis], g["rgb"][vis], rtol=1e-5, atol=1e-6)
    # tile rectangles vs the dense torch oracle
    pre = rt.preprocess(inp["means3D"], inp["opacities"], inp["view"], inp["proj"], inp["campos"], cam["tanfovx"],
                        cam["tanfovy"], inp["W"], inp["H"], cov3D_precomp=inp["cov6"], shs=inp["shs"], sh_degree=3)
    rect = np.stack([r.numpy() for r in pre["rect"]], 1)
    np.testing.assert_array_equal(o["rect"][vis], rect[vis])


def _bwd_case(hostcheck, inp, prm, use_sh, from_sr, seed, invdepth=False, use_dcol=False):
    """Feed the same accumulated screen-space gradients to the product's per-Gaussian backward and to an
    autograd evaluation of the oracle's preprocess stage."""
    P = prm.P
    rng = np.random.default_rng(seed)
    shs = _np(inp["shs"]) if use_sh else None
    colors = None if use_sh else _np(inp["rgb"])
    q = torch.nn.functional.normalize(inp["scene"]["rotation"]) * 1.1       # deliberately not unit norm
    scales, rots = (_np(inp["scales"]), _np(q)) if from_sr else (None, None)
    cov = None if from_sr else _np(inp["cov6"])
    o = _run_pre(hostcheck, inp, prm, shs=shs, colors=colors, cov=cov, scales=scales, rots=rots)
    vis = o["radii"] > 0
    acc = np.zeros((P, 16), np.float32)       # D3GA_ACC_STRIDE
    acc[:, [0, 1, 3, 4, 5, 6, 7, 8, 9]] = rng.normal(size=(P, 9)).astype(np.float32)
    if invdepth:
        acc[:, 10] = rng.normal(size=P).astype(np.float32)
    acc[~vis] = 0
    outs = dict(m3=np.zeros((P, 3), np.float32), m2=np.zeros((P, 3), np.float32), op=np.zeros((P, 1), np.float32),
                sh=np.zeros((P, 16, 3), np.float32) if use_sh else None,
                col=None if use_sh else np.zeros((P, 3), np.float32),
                cov=None if from_sr else np.zeros((P, 6), np.float32),
                sc=np.zeros((P, 3), np.float32) if from_sr else None, ro=np.zeros((P, 4), np.float32) if from_sr else None)
    hostcheck.hc_preprocess_bwd(ctypes.byref(prm), ptr(_np(inp["means3D"])), ptr(shs), ptr(scales), ptr(rots),
                                ptr(_np(inp["view"])), ptr(_np(inp["proj"])), ptr(_np(inp["campos"])), ptr(o["radii"]),
                                ptr(o["cov3D"]), ptr(o["clamped"]), ptr(acc), ptr(outs["m3"]), ptr(outs["m2"]),
                                ptr(outs["op"]), ptr(outs["sh"]), ptr(outs["col"]), ptr(outs["cov"]), ptr(outs["sc"]),
                                ptr(outs["ro"]), ptr(np.ascontiguousarray(o["conic_o"][:, 3])), int(use_dcol))
    if use_sh and not use_dcol:       # round 5: the same case with d(colour)/d(direction) from the forward instead of the coefficients
        _bwd_case(hostcheck, inp, prm, use_sh, from_sr, seed, invdepth, use_dcol=True)
    # oracle: autograd through preprocess with a linear functional reproducing `acc`
    dd = torch.float64
    m = inp["means3D"].to(dd).requires_grad_(True)
    kw = {}
    if from_sr:
        s_t = torch.from_numpy(scales).to(dd).requires_grad_(True)
        r_t = torch.from_numpy(rots).to(dd).requires_grad_(True)
        kw.update(scales=s_t, rotations=r_t, scale_modifier=prm.scale_modifier)
    else:
        c_t = inp["cov6"].to(dd).requires_grad_(True)
        kw.update(cov3D_precomp=c_t)
    if use_sh:
        sh_t = inp["shs"].to(dd).requires_grad_(True)
        kw.update(shs=sh_t, sh_degree=prm.sh_degree)
    else:
        col_t = inp["rgb"].to(dd).requires_grad_(True)
        kw.update(colors_precomp=col_t)
    op_t = inp["opacities"].to(dd).requires_grad_(True)
    pre = rt.preprocess(m, op_t, inp["view"], inp["proj"], inp["campos"], prm.tanfovx, prm.tanfovy,
                        prm.W, prm.H, antialiasing=bool(prm.antialiasing), **kw)
    a = torch.from_numpy(acc).to(dd)
    # mean2D gradient is expressed in NDC-scaled units: d(pixel)/d(ndc) = 0.5*W  => pixel-space grad = a / (0.5 W)
    L = (pre["xy"][:, 0] * a[:, 0] / (0.5 * prm.W)).sum() + (pre["xy"][:, 1] * a[:, 1] / (0.5 * prm.H)).sum()
    L = L + (pre["conic"][:, 0] * a[:, 3]).sum() + (pre["conic"][:, 1] * 2.0 * a[:, 4]).sum() + (pre["conic"][:, 2] * a[:, 5]).sum()
    L = L + (pre["rgb"] * a[:, 7:10]).sum() + (pre["opacity"] * a[:, 6]).sum()
    if invdepth:
        visible = torch.from_numpy(vis)
        L = L + ((1.0 / pre["depth"][visible]) * a[visible, 10]).sum()
    L.backward()
    assert rel_err(outs["m3"], _np(m.grad)) < 1e-3
    if use_sh:
        assert rel_err(outs["sh"], _np(sh_t.grad)) < 1e-4
    else:
        assert rel_err(outs["col"], _np(col_t.grad)) < 1e-5
    if from_sr:
        assert rel_err(outs["sc"], _np(s_t.grad)) < 1e-3
        assert rel_err(outs["ro"], _np(r_t.grad)) < 1e-3
    else:
        assert rel_err(outs["cov"], _np(c_t.grad)) < 1e-3
    np.testing.assert_array_equal(outs["m2"][:, :2], acc[:, :2])
    if prm.antialiasing:
        assert rel_err(outs["op"][:, 0], _np(op_t.grad).reshape(-1)) < 1e-5
    else:
        np.testing.assert_array_equal(outs["op"][:, 0], acc[:, 6])


def test_preprocess_backward_sh_precomputed_cov(hostcheck):
    inp = scene_inputs("T1", scale_mult=3.0)
    _bwd_case(hostcheck, inp, _prm(inp, 16, 3), use_sh=True, from_sr=False, seed=1)
    _bwd_case(hostcheck, inp, _prm(inp, 16, 1), use_sh=True, from_sr=False, seed=2)


def test_preprocess_backward_colors_scale_rot(hostcheck):
    inp = scene_inputs("T1", scale_mult=3.0)
    _bwd_case(hostcheck, inp, _prm(inp, 0, 0, mod=1.3), use_sh=False, from_sr=True, seed=3)


def test_preprocess_backward_antialiasing_and_inverse_depth(hostcheck):
    """Branch dr_aa: the opacity seen by compositing is opacity x sqrt(max(2.5e-5, det / det_dilated)) -- its chain into
    cov3D / mean / opacity -- and acc[10] = dL/d(1/z) chains into the mean.  Small Gaussians (scale_mult 0.3) put the
    factor well below one; large ones leave it near one."""
    for mult, seed in ((0.3, 5), (3.0, 6)):
        inp = scene_inputs("T1", scale_mult=mult)
        _bwd_case(hostcheck, inp, _prm(inp, 16, 2, antialiasing=1), use_sh=True, from_sr=False, seed=seed, invdepth=True)
        _bwd_case(hostcheck, inp, _prm(inp, 0, 0, mod=1.2, antialiasing=1), use_sh=False, from_sr=True, seed=seed + 10,
                  invdepth=True)
    # forward: the stored opacity is the product
    inp = scene_inputs("T1", scale_mult=0.3)
    prm = _prm(inp, 16, 3, antialiasing=1)
    o = _run_pre(hostcheck, inp, prm, shs=_np(inp["shs"]), cov=_np(inp["cov6"]))
    pre = rt.preprocess(inp["means3D"].double(), inp["opacities"].double(), inp["view"], inp["proj"], inp["campos"],
                        prm.tanfovx, prm.tanfovy, prm.W, prm.H, cov3D_precomp=inp["cov6"].double(), shs=inp["shs"].double(),
                        sh_degree=3, antialiasing=True)
    vis = o["radii"] > 0
    ratio = o["conic_o"][vis, 3] / _np(inp["opacities"]).reshape(-1)[vis]
    assert ratio.min() < 0.5 and ratio.max() <= 1.0
    np.testing.assert_allclose(o["conic_o"][vis, 3], _np(pre["opacity"])[vis], rtol=2e-4)


def test_preprocess_backward_with_clamped_sh_and_frustum_edge(hostcheck):
    """Dark SH colours (clamp mask active) and a camera so close that x/z, y/z leave the 1.3*tanfov guard band."""
    inp = scene_inputs("T1", scale_mult=3.0, azimuth=1.2)
    inp["shs"] = inp["shs"].clone()
    inp["shs"][::2, 0, :] = -2.5           # SH_C0 * (-2.5) + 0.5 < 0  -> clamped
    prm = _prm(inp, 16, 2)
    prm.tanfovx *= 0.25                    # narrow guard band: many Gaussians are clamped in x
    _bwd_case(hostcheck, inp, prm, use_sh=True, from_sr=False, seed=4)

"""The two rasterizer oracles against each other and against known answers (parity for this stage is UNPINNED:
the reference ships no rasterizer source, test or golden vector -- see oracle/__init__.py)."""
import math

import numpy as np
import torch

from oracle import raster_c as rc
from oracle import raster_torch as rt
from util import rel_err, scene_inputs


def _np(t):
    return np.ascontiguousarray(t.detach().numpy())


def _both(inp, dt, deg, use_sh=True, from_sr=False, mod=1.0, seed=1):
    cam = inp["cam"]
    W, H = inp["W"], inp["H"]
    bg = torch.tensor([1.0, 0.5, 0.2])
    gpix = torch.randn(3, H, W, generator=torch.Generator().manual_seed(seed))
    leaf = lambda t: t.to(dt).clone().requires_grad_(True)
    m, o = leaf(inp["means3D"]), leaf(inp["opacities"])
    kw, ckw, leaves = {}, {}, {"means3D": m, "opacities": o}
    if use_sh:
        sh = leaf(inp["shs"]); kw.update(shs=sh, sh_degree=deg); ckw.update(shs=_np(inp["shs"]), sh_degree=deg)
        leaves["shs"] = sh
    else:
        col = leaf(inp["rgb"]); kw.update(colors_precomp=col); ckw.update(colors_precomp=_np(inp["rgb"]))
        leaves["colors"] = col
    if from_sr:
        q = torch.nn.functional.normalize(inp["scene"]["rotation"]) * 0.9
        s, r = leaf(inp["scales"]), leaf(q)
        kw.update(scales=s, rotations=r, scale_modifier=mod)
        ckw.update(scales=_np(inp["scales"]), rotations=_np(q), scale_modifier=mod)
        leaves.update(scales=s, rotations=r)
    else:
        c6 = leaf(inp["cov6"]); kw.update(cov3D_precomp=c6); ckw.update(cov3D_precomp=_np(inp["cov6"]))
        leaves["cov3D"] = c6
    col_t, fT, nc, invd, pre = rt.rasterize(m, o, bg, inp["view"], inp["proj"], inp["campos"], cam["tanfovx"],
                                            cam["tanfovy"], W, H, return_aux=True, **kw)
    (col_t * gpix.to(dt)).sum().backward()
    ccol, radii, cinv, ctx = rc.forward(_np(inp["means3D"]), _np(inp["opacities"]), _np(bg), cam["world_view_transform"],
                                        cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"], cam["tanfovy"],
                                        W, H, **ckw)
    g = rc.backward(ctx, _np(gpix))
    geom = rc.geom(ctx)
    assert np.array_equal(radii, _np(pre["radii"]))
    assert rc.num_rendered(ctx) == int(pre["tiles_touched"].sum())
    assert np.abs(ccol - _np(col_t)).max() < 2e-6
    assert np.abs(cinv - _np(invd)).max() < 2e-5
    assert (geom["n_contrib"] == _np(nc)).mean() > 0.9995
    assert np.abs(geom["final_T"] - _np(fT)).max() < 2e-6
    for k, leaf_t in leaves.items():
        assert rel_err(g[k], _np(leaf_t.grad)) < 2e-5, k
    return g


def test_c_oracle_matches_autograd_oracle_sh_cov():
    inp = scene_inputs("T0", scale_mult=3.0)
    _both(inp, torch.float32, 3)
    _both(inp, torch.float64, 3)
    _both(inp, torch.float64, 1, seed=2)


def test_c_oracle_matches_autograd_oracle_colors_scale_rot():
    inp = scene_inputs("T0", scale_mult=3.0, azimuth=-0.7)
    _both(inp, torch.float64, 0, use_sh=False, from_sr=True, mod=1.4)


def test_c_oracle_matches_autograd_oracle_clamped_sh():
    inp = scene_inputs("T0", scale_mult=3.0)
    inp["shs"] = inp["shs"].clone()
    inp["shs"][::3, 0, :] = -2.5
    _both(inp, torch.float64, 2)


def test_single_gaussian_known_answer():
    """One isotropic Gaussian on the optical axis: pixel (x,y) gets alpha = o*exp(-r^2/(2 s^2)) with
    s^2 = (f sigma/z)^2 + 0.3, colour = alpha*c + (1-alpha)*bg."""
    W = H = 33
    fov = 0.6
    f = W / (2 * math.tan(fov / 2))
    z, sigma, o = 4.0, 0.08, 0.7
    view = torch.eye(4)
    proj = torch.from_numpy(__import__("oracle.camera", fromlist=["projection"]).projection(0.01, 100.0, fov, fov).T.copy())
    means = torch.tensor([[0.0, 0.0, z]])
    cov = torch.tensor([[sigma ** 2, 0, 0, sigma ** 2, 0, sigma ** 2]])
    color = torch.tensor([[0.2, 0.6, 0.9]])
    bg = torch.tensor([0.1, 0.1, 0.1])
    img, radii = rt.rasterize(means.double(), torch.tensor([[o]]).double(), bg, view, proj, torch.zeros(3), math.tan(fov / 2),
                              math.tan(fov / 2), W, H, cov3D_precomp=cov.double(), colors_precomp=color.double())
    s2 = (f * sigma / z) ** 2 + 0.3
    assert int(radii[0]) == math.ceil(3 * math.sqrt(s2))
    cx = cy = (W - 1) / 2            # ndc 0 -> pixel ((0+1)W-1)/2
    for (x, y) in ((16, 16), (18, 16), (16, 20), (10, 12)):
        r2 = (x - cx) ** 2 + (y - cy) ** 2
        a = min(0.99, o * math.exp(-0.5 * r2 / s2))
        if a < 1 / 255:
            a = 0.0
        expect = a * color[0].double() + (1 - a) * bg.double()
        np.testing.assert_allclose(img[:, y, x].numpy(), expect.numpy(), atol=1e-9)


def test_front_to_back_order_and_termination():
    """Depth order decides who is in front; blending stops BEFORE the splat that would take T below 1e-4."""
    W = H = 16
    fov = 0.5
    view = torch.eye(4)
    from oracle.camera import projection
    proj = torch.from_numpy(projection(0.01, 100.0, fov, fov).T.copy())
    means = torch.tensor([[0.0, 0.0, 5.0], [0.0, 0.0, 3.0], [0.0, 0.0, 4.0]]).double()
    cov = torch.tensor([[1.0, 0, 0, 1.0, 0, 1.0]]).repeat(3, 1).double()
    col = torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]).double()
    op = torch.tensor([[1.0], [1.0], [0.9]]).double()      # red z=5, green z=3, blue z=4
    img, _ = rt.rasterize(means, op, torch.zeros(3), view, proj, torch.zeros(3), math.tan(fov / 2), math.tan(fov / 2), W, H,
                          cov3D_precomp=cov, colors_precomp=col)
    px = img[:, 8, 8]
    # order: green (alpha .99), blue (alpha .9) -> T = 1e-3; red (alpha .99) would give T = 1e-5 < 1e-4: not blended
    f = W / (2 * math.tan(fov / 2))
    a_blue = 0.9 * math.exp(-0.5 * (0.5 ** 2 + 0.5 ** 2) / ((f * 1.0 / 4.0) ** 2 + 0.3))   # centre is at pixel 7.5
    np.testing.assert_allclose(px.numpy(), [0.0, 0.99, a_blue * 0.01], atol=1e-9)
    c, _, _, ctx = rc.forward(means.numpy(), op.numpy(), np.zeros(3), view.numpy(), proj.numpy(), np.zeros(3),
                              math.tan(fov / 2), math.tan(fov / 2), W, H, cov3D_precomp=cov.numpy(), colors_precomp=col.numpy())
    np.testing.assert_allclose(c[:, 8, 8], [0.0, 0.99, a_blue * 0.01], atol=1e-6)
    assert rc.geom(ctx)["n_contrib"][8, 8] == 2


def test_decision_margins_flag_threshold_pixels_and_their_gaussians():
    """oracle.raster_c.margins: a pixel is marginal when an evaluated alpha sits within eps (scaled by 1 + 0.02 x the magnitude of
    the quadratic form's terms: its float32 rounding is the relative error of alpha) of 1/255 (or a test_T within eps of 1e-4); a Gaussian is marginal when it can contribute to such a pixel.  Checked against a direct evaluation."""
    W = H = 33
    fov = 0.6
    f = W / (2 * math.tan(fov / 2))
    z, sigma, o = 4.0, 0.08, 0.7
    from oracle.camera import projection
    view = np.eye(4, dtype=np.float32)
    proj = projection(0.01, 100.0, fov, fov).T.copy().astype(np.float32)
    means = np.array([[0.0, 0.0, z], [5.0, 5.0, z]], np.float32)             # the second one is far outside the image
    cov = np.array([[sigma ** 2, 0, 0, sigma ** 2, 0, sigma ** 2]] * 2, np.float32)
    _, radii, _, ctx = rc.forward(means, np.array([[o], [o]], np.float32), np.zeros(3, np.float32), view, proj, np.zeros(3, np.float32),
                                  math.tan(fov / 2), math.tan(fov / 2), W, H, cov3D_precomp=cov, colors_precomp=np.ones((2, 3), np.float32))
    s2 = (f * sigma / z) ** 2 + 0.3
    ys, xs = np.mgrid[0:H, 0:W]
    alpha = np.minimum(0.99, o * np.exp(-0.5 * ((xs - 16.0) ** 2 + (ys - 16.0) ** 2) / s2))
    tmag = 0.5 * ((xs - 16.0) ** 2 + (ys - 16.0) ** 2) / s2            # magnitude of the quadratic form's terms (B = 0 here)
    rel = np.abs(alpha * 255.0 - 1.0) / (1.0 + 0.02 * tmag)            # distance in units of what float32 resolves there
    inside = np.hypot(xs - 16.0, ys - 16.0) <= radii[0] + 16          # pixels whose tile holds the splat at all
    for eps in (0.5, 0.05):
        pix, gs = rc.margins(ctx, eps_alpha=eps, eps_T=0.0)
        want = (rel < eps)
        # tiles the splat does not touch evaluate nothing there: compare where the oracle evaluated it
        t_touched = np.zeros((H, W), bool)
        r = int(radii[0])
        t_touched[max(0, (16 - r) // 16 * 16):min(H, ((16 + r) // 16 + 1) * 16), max(0, (16 - r) // 16 * 16):min(W, ((16 + r) // 16 + 1) * 16)] = True
        assert np.array_equal(pix & t_touched, want & t_touched), eps
        assert bool(gs[0]) == bool(want[t_touched].any()) and not gs[1]
    pix, gs = rc.margins(ctx, eps_alpha=1e-9, eps_T=0.0)
    assert not pix.any() and not gs.any()
    assert inside.any()


def test_shared_termination_and_alpha_marginal_hooks():
    """The hooks of the shared-decision parity test (tests/test_gpu_parity.py): handing the oracle its OWN termination changes
    nothing; truncating one pixel's n_contrib removes exactly the entries behind it from that pixel's backward; the
    alpha-marginal flags grow with eps, vanish at eps = 0 and are a subset of `margins()`' Gaussian flags."""
    inp = scene_inputs("T1", scale_mult=2.0)
    cam, W, H = inp["cam"], inp["W"], inp["H"]
    bg = np.array([1.0, 0.5, 0.2], np.float32)
    gpix = np.random.default_rng(3).standard_normal((3, H, W)).astype(np.float32)
    _, _, _, ctx = rc.forward(_np(inp["means3D"]), _np(inp["opacities"]), bg, cam["world_view_transform"],
                              cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"], cam["tanfovy"], W, H,
                              cov3D_precomp=_np(inp["cov6"]), shs=_np(inp["shs"]), sh_degree=3)
    g0 = rc.backward(ctx, gpix)
    own = rc.geom(ctx)
    rc.set_termination(ctx, own["final_T"], own["n_contrib"])
    g1 = rc.backward(ctx, gpix)
    for k in g0:
        if g0[k] is not None:
            assert np.array_equal(g0[k], g1[k]), k
    # one deep pixel gives up everything: only Gaussians of its tile may change, and something does change
    py, px = np.unravel_index(int(own["n_contrib"].argmax()), own["n_contrib"].shape)
    nc = own["n_contrib"].copy()
    nc[py, px] = 0
    rc.set_termination(ctx, own["final_T"], nc)
    g2 = rc.backward(ctx, gpix)
    changed = np.flatnonzero(np.abs(g2["opacities"] - g0["opacities"]).reshape(-1) > 0)
    start, plist = rc.tile_lists(ctx)
    t = (py // 16) * ((W + 15) // 16) + px // 16
    assert len(changed) > 0 and set(changed.tolist()) <= set(plist[start[t]:start[t + 1]].tolist())
    rc.set_termination(ctx, own["final_T"], own["n_contrib"])
    none, few, many = rc.alpha_marginal(ctx, 0.0), rc.alpha_marginal(ctx, 1e-3), rc.alpha_marginal(ctx, 1e-1)
    assert not none.any() and many.sum() > few.sum() > 0 and not (few & ~many).any()
    _, gm = rc.margins(ctx, 1e-3, 0.0)
    assert not (few & ~gm).any()


def test_alpha_override_hooks():
    """`alpha_band_pairs` / `set_alpha_overrides` (shared alpha decisions of the GPU parity test): overriding the band pairs
    with the oracle's OWN decisions changes nothing; flipping ONE decision changes that pair's Gaussian and only Gaussians of
    the same tile; clearing restores the plain backward."""
    inp = scene_inputs("T1", scale_mult=2.0)
    cam, W, H = inp["cam"], inp["W"], inp["H"]
    bg = np.array([1.0, 0.5, 0.2], np.float32)
    gpix = np.random.default_rng(5).standard_normal((3, H, W)).astype(np.float32)
    _, _, _, ctx = rc.forward(_np(inp["means3D"]), _np(inp["opacities"]), bg, cam["world_view_transform"],
                              cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"], cam["tanfovy"], W, H,
                              cov3D_precomp=_np(inp["cov6"]), shs=_np(inp["shs"]), sh_degree=3)
    g0 = rc.backward(ctx, gpix)
    gid, pix = rc.alpha_band_pairs(ctx, 5e-2)
    assert len(gid) > 10 and (np.diff(pix.astype(np.int64) * (1 << 31) + gid) > 0).all()
    geom = rc.geom(ctx)
    dx, dy = geom["xy"][gid, 0] - (pix % W).astype(np.float32), geom["xy"][gid, 1] - (pix // W).astype(np.float32)
    co = geom["conic_o"][gid]
    power = np.float32(-0.5) * (co[:, 0] * dx * dx + co[:, 2] * dy * dy) - co[:, 1] * dx * dy
    alpha = np.minimum(np.float32(0.99), co[:, 3] * np.exp(power))
    sure = np.abs(alpha * 255.0 - 1.0) > 1e-4            # (numpy's exp vs libm's expf: stay off the exact threshold)
    gid, pix, alpha, power = gid[sure], pix[sure], alpha[sure], power[sure]
    own = ((power <= 0) & (alpha >= 1.0 / 255.0)).astype(np.uint8)
    rc.set_alpha_overrides(ctx, gid, pix, own)
    g1 = rc.backward(ctx, gpix)
    for k in g0:
        if g0[k] is not None:
            assert np.array_equal(g0[k], g1[k]), k
    i = int(np.flatnonzero(own == 1)[0])
    flipped = own.copy(); flipped[i] = 0
    rc.set_alpha_overrides(ctx, gid, pix, flipped)
    g2 = rc.backward(ctx, gpix)
    changed = np.flatnonzero(np.abs(g2["opacities"] - g0["opacities"]).reshape(-1) > 0)
    start, plist = rc.tile_lists(ctx)
    t = (int(pix[i]) // W // 16) * ((W + 15) // 16) + (int(pix[i]) % W) // 16
    assert gid[i] in changed and set(changed.tolist()) <= set(plist[start[t]:start[t + 1]].tolist())
    rc.set_alpha_overrides(ctx, gid[:0], pix[:0], own[:0])
    g3 = rc.backward(ctx, gpix)
    assert np.array_equal(g0["means3D"], g3["means3D"])


def test_antialiasing_and_invdepth_gradient_c_oracle_matches_autograd_oracle():
    """Branch dr_aa's two extras ([UPSTREAM-RECALL], D3GA itself passes antialiasing=False and drops the depth image): the
    opacity compensation h = sqrt(max(2.5e-5, det(cov2D) / det(cov2D + 0.3 I))) and a loss on the inverse-depth image.  The
    hand-derived C backward (d h / d cov2D, the fourth channel of the compositing backward, d(1/z)/dmean) against autograd."""
    inp = scene_inputs("T0", scale_mult=2.0)
    cam, W, H = inp["cam"], inp["W"], inp["H"]
    bg = torch.tensor([0.3, 0.6, 0.1])
    gen = torch.Generator().manual_seed(5)
    gpix, gd = torch.randn(3, H, W, generator=gen), torch.randn(H, W, generator=gen)
    dt = torch.float64
    leaf = lambda t: t.to(dt).clone().requires_grad_(True)
    m, o, c6, sh = leaf(inp["means3D"]), leaf(inp["opacities"]), leaf(inp["cov6"]), leaf(inp["shs"])
    col, fT, nc, invd, pre = rt.rasterize(m, o, bg, inp["view"], inp["proj"], inp["campos"], cam["tanfovx"], cam["tanfovy"], W, H,
                                          cov3D_precomp=c6, shs=sh, sh_degree=3, return_aux=True, antialiasing=True)
    ((col * gpix.to(dt)).sum() + (invd * gd.to(dt)).sum()).backward()
    ccol, radii, cinv, ctx = rc.forward(_np(inp["means3D"]), _np(inp["opacities"]), _np(bg), cam["world_view_transform"],
                                        cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"], cam["tanfovy"], W, H,
                                        cov3D_precomp=_np(inp["cov6"]), shs=_np(inp["shs"]), sh_degree=3, antialiasing=True)
    g = rc.backward(ctx, _np(gpix), dL_dinvdepth=_np(gd))
    assert np.abs(ccol - _np(col)).max() < 1e-5 and np.abs(cinv - _np(invd)).max() < 1e-5
    # the factor does something here: the same scene without it renders differently
    ccol0, _, _, ctx0 = rc.forward(_np(inp["means3D"]), _np(inp["opacities"]), _np(bg), cam["world_view_transform"],
                                   cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"], cam["tanfovy"], W, H,
                                   cov3D_precomp=_np(inp["cov6"]), shs=_np(inp["shs"]), sh_degree=3)
    assert np.abs(ccol0 - ccol).max() > 1e-3
    for k, t in (("means3D", m), ("opacities", o), ("cov3D", c6), ("shs", sh)):
        assert rel_err(g[k], _np(t.grad)) < 5e-5, (k, rel_err(g[k], _np(t.grad)))
    # and each extra on its own: inverse depth without antialiasing, antialiasing without a depth loss
    for aa, use_d in ((False, True), (True, False)):
        for t in (m, o, c6, sh):
            t.grad = None
        col, fT, nc, invd, pre = rt.rasterize(m, o, bg, inp["view"], inp["proj"], inp["campos"], cam["tanfovx"], cam["tanfovy"], W, H,
                                              cov3D_precomp=c6, shs=sh, sh_degree=3, return_aux=True, antialiasing=aa)
        ((col * gpix.to(dt)).sum() + ((invd * gd.to(dt)).sum() if use_d else 0.0)).backward()
        _, _, _, ctx = rc.forward(_np(inp["means3D"]), _np(inp["opacities"]), _np(bg), cam["world_view_transform"],
                                  cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"], cam["tanfovy"], W, H,
                                  cov3D_precomp=_np(inp["cov6"]), shs=_np(inp["shs"]), sh_degree=3, antialiasing=aa)
        g = rc.backward(ctx, _np(gpix), dL_dinvdepth=_np(gd) if use_d else None)
        for k, t in (("means3D", m), ("opacities", o), ("cov3D", c6), ("shs", sh)):
            assert rel_err(g[k], _np(t.grad)) < 5e-5, (aa, use_d, k, rel_err(g[k], _np(t.grad)))

"""View-batched rasterization (d3ga_amd/raster_views.py; include/d3ga.h: d3ga_raster_params::n_views): k cameras of one set of
Gaussians in one grid per stage.  Claim under test: every view's image IS the single-view render of that camera and the gradients
are the sum over the k single-view backwards -- the single-view operator is the one the oracle tests pin (tests/test_gpu_parity.py),
and one case here goes to the C oracle directly."""
import math
import os

import numpy as np
import pytest
import torch

from util import Parity, scene_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _order_bar(from_sr):
    """Bar of the same-arithmetic comparisons below (max |a - b| / max |a| per leaf): both sides form the same sums, with the float
    atomics of the compositing backward landing in another order -- as they do from one run of the SAME operator to the next.
    tools/diag_spread.py measures that spread on these scenes (single-view operator against itself, 30 runs): <= 4e-7 on every
    leaf when the covariance is given; with (scales, rotations) -- splats without the cage's deformation gradient, some of them
    needles whose covariance Jacobian cancels heavily -- 5.4e-5 on means3D, 5.5e-5 on scales, 6.4e-5 on rotations.  The 2e-5 of
    the other same-arithmetic tests would be below the operator's own run-to-run spread there (it failed one run in three)."""
    return 2e-4 if from_sr else 2e-5


def _batches(inp, k, fov_jitter=False):
    from d3ga_amd import synthetic as syn
    out = []
    for v in range(k):
        b = syn.make_batch(inp["W"], inp["H"], azimuth=0.4 + 2 * math.pi * v / max(k, 3), camera_id=v,
                           fill=0.85 * (1.0 + (0.1 * v if fov_jitter else 0.0)))
        out.append(b)
    return out


def _single_view(inp, batch, leaves, bg, use_sh, from_sr, target=None, gpix=None):
    """One view through the single-view operator -> (image, loss | None); gradients accumulate into the leaves."""
    from d3ga_amd import rasterizer as R
    from d3ga_amd.cameras import batch_to_camera
    cam = batch_to_camera(batch, device=DEV)
    s = R.GaussianRasterizationSettings(
        image_height=inp["H"], image_width=inp["W"], tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3 if use_sh else 0,
        campos=cam.camera_center, prefiltered=False, debug=False, antialiasing=False)
    args = (leaves["means3D"], None, leaves["shs"] if use_sh else None, None if use_sh else leaves["rgb"], leaves["opacities"],
            leaves["scales"] if from_sr else None, leaves["rots"] if from_sr else None, None if from_sr else leaves["cov6"], s)
    if target is not None:
        img, _, _, loss = R.rasterize_gaussians_l1(*args, target, want_invdepth=False)
        return img, loss
    img = R.rasterize_gaussians(*args, want_invdepth=False)[0]
    return img, None


def _leaves(inp, use_sh, from_sr):
    d = {"means3D": inp["means3D"], "opacities": inp["opacities"]}
    d.update({"shs": inp["shs"]} if use_sh else {"rgb": inp["rgb"]})
    d.update({"scales": inp["scales"], "rots": inp["scene"]["rotation"]} if from_sr else {"cov6": inp["cov6"]})
    return {k: v.to(DEV).clone().contiguous().requires_grad_(True) for k, v in d.items()}


def _views(inp, batches, leaves, bg, use_sh, from_sr, targets=None):
    from d3ga_amd.raster_views import CameraBatch, rasterize_gaussians_views
    cams = CameraBatch(len(batches), inp["W"], inp["H"], device=DEV).set(batches)
    return rasterize_gaussians_views(
        leaves["means3D"], leaves["shs"] if use_sh else None, None if use_sh else leaves["rgb"], leaves["opacities"],
        leaves["scales"] if from_sr else None, leaves["rots"] if from_sr else None, None if from_sr else leaves["cov6"], cams, bg,
        sh_degree=3 if use_sh else 0, l1_targets=targets)


@pytest.mark.parametrize("name,scale_mult,k,use_sh,from_sr", [("T1", 3.0, 3, True, False), ("C1", 1.0, 4, True, False),
                                                               ("T1", 3.0, 2, False, True), ("T1", 8.0, 5, False, False),
                                                               # the looped per-Gaussian kernels' group splits with SH colours: 2, 3 + 2,
                                                               # 4 + 3, and 4 + 3 + 2 forward with a SECOND backward group (more than 8 views)
                                                               ("T1", 3.0, 2, True, False), ("T1", 3.0, 5, True, True),
                                                               ("T1", 2.0, 7, True, False), ("T1", 2.0, 9, True, False)])
def test_batched_views_equal_the_single_view_renders(name, scale_mult, k, use_sh, from_sr):
    """Images: bit-identical per view (the same kernels on the same per-view records; only the tile numbering differs).
    Gradients: the sum over the views, formed in another order (float atomics across tiles; the SH block rebuilt from the k
    per-view factors in one pass) -- 2e-5 of the largest element, the bar of the other same-arithmetic comparisons."""
    inp = scene_inputs(name, scale_mult=scale_mult)
    batches = _batches(inp, k, fov_jitter=True)
    bg = torch.tensor([0.3, 0.6, 0.1], device=DEV)
    g = torch.Generator().manual_seed(3)
    gpix = torch.randn(k, 3, inp["H"], inp["W"], generator=g).to(DEV)
    ref = _leaves(inp, use_sh, from_sr)
    imgs = []
    for v, b in enumerate(batches):
        img, _ = _single_view(inp, b, ref, bg, use_sh, from_sr)
        (img * gpix[v]).sum().backward()
        imgs.append(img.detach())
    mine = _leaves(inp, use_sh, from_sr)
    colors, radii = _views(inp, batches, mine, bg, use_sh, from_sr)
    (colors * gpix).sum().backward()
    torch.cuda.synchronize()
    assert colors.shape == (k, 3, inp["H"], inp["W"]) and radii.shape == (k, inp["means3D"].shape[0])
    for v in range(k):
        assert torch.equal(colors[v], imgs[v]), (v, float((colors[v] - imgs[v]).abs().max()))
        assert float((colors[v].detach() - bg.view(3, 1, 1)).abs().max()) > 0.05        # something was rendered in every view
    for key in ref:
        a, b = ref[key].grad, mine[key].grad
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= _order_bar(from_sr) * scale, (key, float((a - b).abs().max()) / scale)


def test_batched_views_with_the_fused_l1_loss():
    """loss = mean |colors - targets| over all k images = the mean over the views of the single-view fused L1 losses; its gradient
    is formed inside the batched compositing backward."""
    inp = scene_inputs("C1")
    k = 3
    batches = _batches(inp, k)
    bg = torch.ones(3, device=DEV)
    g = torch.Generator().manual_seed(5)
    targets = torch.rand(k, 3, inp["H"], inp["W"], generator=g).to(DEV)
    ref = _leaves(inp, True, False)
    tot = 0.0
    for v, b in enumerate(batches):
        _, loss = _single_view(inp, b, ref, bg, True, False, target=targets[v].contiguous())
        (loss / k).backward()
        tot += float(loss) / k
    mine = _leaves(inp, True, False)
    colors, _, loss = _views(inp, batches, mine, bg, True, False, targets=targets)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - tot) <= 2e-6 * abs(tot), (float(loss), tot)
    for key in ref:
        a, b = ref[key].grad, mine[key].grad
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale, (key, float((a - b).abs().max()) / scale)


def test_batched_views_against_the_oracle():
    """Two views of T1 against the C oracle directly, under the bars of the single-view parity tests (strict image bar on the
    non-marginal pixels; element-wise gradient bar on the SUM of the two views' oracle gradients, the incoming gradient zeroed
    on the oracle's marginal pixels on both sides)."""
    from oracle import raster_c as rc
    from oracle import camera as oc
    inp = scene_inputs("T1", scale_mult=3.0)
    k = 2
    batches = _batches(inp, k)
    bgc = torch.tensor([0.2, 0.5, 0.9])
    npy = lambda t: np.ascontiguousarray(t.detach().cpu().numpy())
    g = torch.Generator().manual_seed(9)
    gpix = torch.randn(k, 3, inp["H"], inp["W"], generator=g)
    octx, oimg = [], []
    for v, b in enumerate(batches):
        cam = oc.camera(b["R"], b["T"], b["FoVx"], b["FoVy"])
        img, _, _, ctx = rc.forward(npy(inp["means3D"]), npy(inp["opacities"]), npy(bgc), cam["world_view_transform"],
                                    cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"], cam["tanfovy"], inp["W"], inp["H"],
                                    cov3D_precomp=npy(inp["cov6"]), shs=npy(inp["shs"]), sh_degree=3)
        octx.append(ctx); oimg.append(img)
    pars = [Parity(c) for c in octx]
    for v in range(k):
        pars[v].mask(gpix[v])
    og = None
    for v in range(k):
        gv = rc.backward(octx[v], npy(gpix[v]))
        og = gv if og is None else {key: (None if og[key] is None else og[key] + gv[key]) for key in og}
    mine = _leaves(inp, True, False)
    colors, _ = _views(inp, batches, mine, bgc.to(DEV), True, False)
    (colors * gpix.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    for v in range(k):
        ok, mx, mxm, nm = pars[v].image(npy(colors[v]), oimg[v])
        assert ok, (v, mx, mxm, nm)
    par = pars[0]                                            # (both views were masked: every Gaussian is held to the strict bar)
    for key, okey in (("means3D", "means3D"), ("cov6", "cov3D"), ("opacities", "opacities"), ("shs", "shs")):
        ok, excess, _, _ = par.grads(npy(mine[key].grad), og[okey])
        assert ok, (key, excess)


def test_batched_views_refusals_and_sizes():
    import ctypes
    from d3ga_amd import _lib
    L = _lib.lib()
    s1, s4 = (ctypes.c_int64 * 3)(), (ctypes.c_int64 * 3)()
    assert L.d3ga_raster_scratch_bytes_views(1000, 640, 360, 1, 5000, 0, s1) == 0
    assert L.d3ga_raster_scratch_bytes_views(1000, 640, 360, 4, 20000, 0, s4) == 0
    assert s4[0] > 3 * s1[0] and s4[2] > 3 * s1[2]
    assert L.d3ga_raster_scratch_bytes_views(1000, 640, 16 * 70000, 1, 5000, 0, s1) == -2        # tile rows are 16-bit fields
    assert L.d3ga_raster_scratch_bytes_views(1000, 640, 360, -1, 5000, 0, s1) == -2
    prm = _lib.RasterParams(P=16, M=0, sh_degree=0, W=64, H=64, tanfovx=1.0, tanfovy=1.0, scale_modifier=1.0, n_views=2)
    buf = torch.zeros(1 << 22, dtype=torch.uint8, device=DEV)
    p = ctypes.c_void_p(buf.data_ptr())
    assert L.d3ga_raster_recolor(ctypes.byref(prm), p, None, p, p, p, ctypes.c_void_p(buf.data_ptr() + (1 << 21)), None) == -3      # D3GA_E_CONFIG


@pytest.mark.parametrize("use_sh,from_sr,k", [(True, False, 3), (False, True, 3), (True, True, 5), (True, False, 9)])
def test_batch_of_frames_with_per_view_geometry(use_sh, from_sr, k):
    """per_view_geometry: the reference's batch holds FRAMES (train.py:218-221) -- the avatar deformed per pose, (k,P,.) geometry, shared
    appearance.  Every image equals the single-view render of (frame v's Gaussians, camera v); the geometry gradients come back per
    frame, the appearance gradients summed over the frames."""
    inp = scene_inputs("T1", scale_mult=3.0)
    batches = _batches(inp, k)
    bg = torch.tensor([0.9, 0.8, 0.7], device=DEV)
    g = torch.Generator().manual_seed(13)
    gpix = torch.randn(k, 3, inp["H"], inp["W"], generator=g).to(DEV)
    P = inp["means3D"].shape[0]
    shift = 0.02 * torch.randn(k, P, 3, generator=g)
    base = _leaves(inp, use_sh, from_sr)
    geo = ("means3D",) + (("scales", "rots") if from_sr else ("cov6",))
    per = {n: torch.stack([base[n].detach().cpu() * (1.0 + (0.05 * v if n != "means3D" else 0.0)) + (shift[v] if n == "means3D" else 0.0)
                           for v in range(k)]).to(DEV) for n in geo}
    # reference: k single-view renders, each with its own geometry leaves, the appearance leaves shared
    shared = {n: t for n, t in _leaves(inp, use_sh, from_sr).items() if n not in geo}
    ref_geo, imgs = [], []
    for v, b in enumerate(batches):
        lv = dict(shared, **{n: per[n][v].clone().requires_grad_(True) for n in geo})
        img, _ = _single_view(inp, b, lv, bg, use_sh, from_sr)
        (img * gpix[v]).sum().backward()
        imgs.append(img.detach())
        ref_geo.append({n: lv[n].grad for n in geo})
    mine_shared = {n: t for n, t in _leaves(inp, use_sh, from_sr).items() if n not in geo}
    mine_geo = {n: per[n].clone().requires_grad_(True) for n in geo}
    colors, _ = _views(inp, batches, dict(mine_shared, **mine_geo), bg, use_sh, from_sr)
    (colors * gpix).sum().backward()
    torch.cuda.synchronize()
    for v in range(k):
        assert torch.equal(colors[v], imgs[v]), v
        for n in geo:
            a, b = ref_geo[v][n], mine_geo[n].grad[v]
            scale = float(a.abs().max())
            assert scale > 0 and float((a - b).abs().max()) <= _order_bar(from_sr) * scale, (v, n, float((a - b).abs().max()) / scale)
    for n in shared:
        a, b = shared[n].grad, mine_shared[n].grad
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale, (n, float((a - b).abs().max()) / scale)


def test_batched_pair_render_equals_the_single_view_pairs():
    """The reference's RGB + silhouette pair (models/trainer.py:102-110) for a batch: colors2 (P,3) shared by the views, blended with the
    same alphas into a second (k,3,H,W) image; both images' gradients reach the geometry and the opacities."""
    from d3ga_amd import rasterizer as R
    from d3ga_amd.cameras import batch_to_camera
    from d3ga_amd.raster_views import CameraBatch, rasterize_gaussians_views
    inp = scene_inputs("C1")
    k = 3
    batches = _batches(inp, k)
    bg, bg2 = torch.ones(3, device=DEV), torch.zeros(3, device=DEV)
    g = torch.Generator().manual_seed(21)
    P = inp["means3D"].shape[0]
    sil = torch.rand(P, 3, generator=g).to(DEV)
    gp, gp2 = torch.randn(k, 3, inp["H"], inp["W"], generator=g).to(DEV), torch.randn(k, 3, inp["H"], inp["W"], generator=g).to(DEV)
    ref = _leaves(inp, True, False)
    imgs, imgs2 = [], []
    for v, b in enumerate(batches):
        cam = batch_to_camera(b, device=DEV)
        s = R.GaussianRasterizationSettings(
            image_height=inp["H"], image_width=inp["W"], tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0,
            viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center,
            prefiltered=False, debug=False, antialiasing=False)
        img, _, _, img2 = R.rasterize_gaussians_pair(ref["means3D"], None, ref["shs"], None, ref["opacities"], None, None, ref["cov6"], s,
                                                     sil, bg2, want_invdepth=False)
        ((img * gp[v]).sum() + (img2 * gp2[v]).sum()).backward()
        imgs.append(img.detach()); imgs2.append(img2.detach())
    mine = _leaves(inp, True, False)
    cams = CameraBatch(k, inp["W"], inp["H"], device=DEV).set(batches)
    colors, _, colors2 = rasterize_gaussians_views(mine["means3D"], mine["shs"], None, mine["opacities"], None, None, mine["cov6"], cams, bg,
                                                   sh_degree=3, colors2=sil, bg2=bg2)
    ((colors * gp).sum() + (colors2 * gp2).sum()).backward()
    torch.cuda.synchronize()
    for v in range(k):
        assert torch.equal(colors[v], imgs[v]) and torch.equal(colors2[v], imgs2[v]), v
    for n in ref:
        a, b = ref[n].grad, mine[n].grad
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale, (n, float((a - b).abs().max()) / scale)


def test_render_views_over_a_batch_of_frames_equals_render_per_frame():
    """renderer.render_views with a LIST of packages (one per frame of the batch: own means / covariances, shared appearance tensors)
    and the silhouette pair against renderer.render_pair per frame: identical images, the same summed gradients on the shared leaves."""
    from d3ga_amd.renderer import render_pair, render_views
    inp = scene_inputs("T1", scale_mult=3.0)
    k = 3
    batches = _batches(inp, k)
    bg, bg0 = torch.ones(3, device=DEV), torch.zeros(3, device=DEV)
    P = inp["means3D"].shape[0]
    g = torch.Generator().manual_seed(31)
    sil = torch.ones(P, 3, device=DEV)
    gp, gp2 = torch.randn(k, 3, inp["H"], inp["W"], generator=g).to(DEV), torch.randn(k, 3, inp["H"], inp["W"], generator=g).to(DEV)
    shift = (0.02 * torch.randn(k, P, 3, generator=g)).to(DEV)

    def leaves():
        return {n: inp[n].to(DEV).clone().requires_grad_(True) for n in ("means3D", "cov6", "opacities", "shs")}

    def packages(L):          # "poses": the shared means moved per frame (a differentiable function of the shared leaf, like the deform)
        return [{"means3D": L["means3D"] + shift[v], "cov3D_precomp": L["cov6"] * (1.0 + 0.1 * v), "opacities": L["opacities"], "shs": L["shs"],
                 "rgb": None, "sh_degree": 3} for v in range(k)]
    ref = leaves()
    imgs = []
    for v, pk in enumerate(packages(ref)):
        both = render_pair(batches[v], pk, bg, sil, bg0)
        ((both["render"] * gp[v]).sum() + (both["render2"] * gp2[v]).sum()).backward()
        imgs.append((both["render"].detach(), both["render2"].detach()))
    mine = leaves()
    out = render_views(batches, packages(mine), bg, colors2=sil, bg_color2=bg0)
    ((out["render"] * gp).sum() + (out["render2"] * gp2).sum()).backward()
    torch.cuda.synchronize()
    for v in range(k):
        assert torch.equal(out["render"][v], imgs[v][0]) and torch.equal(out["render2"][v], imgs[v][1]), v
    for n in ref:
        a, b = ref[n].grad, mine[n].grad
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale, (n, float((a - b).abs().max()) / scale)


def test_batched_views_inference_single_view_batch_and_empty_scene():
    """The corners of the batched operator: (1) under torch.no_grad the forward_only path (an image scratch without block lists)
    renders the same images; (2) a batch of ONE view is the single-view render; (3) an empty scene gives k background images and
    no gradients to compute."""
    from d3ga_amd.raster_views import CameraBatch, rasterize_gaussians_views
    inp = scene_inputs("T1", scale_mult=3.0)
    k = 3
    batches = _batches(inp, k)
    bg = torch.tensor([0.25, 0.5, 0.75], device=DEV)
    L = _leaves(inp, True, False)
    colors, radii = _views(inp, batches, L, bg, True, False)
    with torch.no_grad():
        c2, r2 = _views(inp, batches, L, bg, True, False)
    assert torch.equal(colors, c2) and torch.equal(radii, r2) and not c2.requires_grad
    one, _ = _views(inp, batches[:1], L, bg, True, False)
    img, _ = _single_view(inp, batches[0], L, bg, True, False)
    assert torch.equal(one[0], img)
    cams = CameraBatch(k, inp["W"], inp["H"], device=DEV).set(batches)
    z = lambda *s: torch.zeros(*s, device=DEV, requires_grad=True)
    e_col, e_rad = rasterize_gaussians_views(z(0, 3), z(0, 16, 3), None, z(0, 1), None, None, z(0, 6), cams, bg, sh_degree=3)
    assert e_col.shape == (k, 3, inp["H"], inp["W"]) and e_rad.shape == (k, 0)
    assert torch.equal(e_col, bg.view(1, 3, 1, 1).expand_as(e_col))
    e_col.sum().backward()                                  # (nothing to differentiate: must not raise)


def _fuzz_case(seed):
    """-> (tag, from_sr, per_view, geo, k, run_sequential, run_batched): one random case of test_views_fuzz; each run_* renders from fresh
    leaves and returns (images, gradients: leaf name -> tensor, per-view geometry as (k,P,.))."""
    from d3ga_amd import rasterizer as R
    from d3ga_amd.cameras import batch_to_camera
    from d3ga_amd.raster_views import CameraBatch, rasterize_gaussians_views
    rng = np.random.default_rng(1000 + seed)
    name = ("T0", "T1")[int(rng.integers(2))]
    inp = scene_inputs(name, scale_mult=float(rng.uniform(1.0, 6.0)), seed=int(rng.integers(1, 50)))
    P0 = inp["means3D"].shape[0]
    P = int(rng.integers(max(P0 // 3, 70), P0 + 1))
    k = int(rng.integers(2, 11))
    use_sh, from_sr, per_view = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(3) == 0)
    deg = int(rng.integers(0, 4)) if use_sh else 0
    M = int(rng.choice([m for m in (1, 4, 9, 12, 16) if m >= (deg + 1) ** 2])) if use_sh else 0
    g = torch.Generator().manual_seed(seed)
    base = {"means3D": inp["means3D"][:P].clone(), "opacities": inp["opacities"][:P].clone()}
    base["means3D"][torch.rand(P, generator=g) < 0.05] *= -3.0           # some land behind a camera / outside the frustum
    if use_sh:
        base["shs"] = inp["shs"][:P, :M].clone()
    else:
        base["rgb"] = inp["rgb"][:P].clone()
    if from_sr:
        base["scales"], base["rots"] = inp["scales"][:P].clone(), inp["scene"]["rotation"][:P].clone()
    else:
        base["cov6"] = inp["cov6"][:P].clone()
    geo = ("means3D",) + (("scales", "rots") if from_sr else ("cov6",))
    batches = _batches(inp, k, fov_jitter=bool(rng.integers(2)))
    bg = torch.rand(3, generator=g).to(DEV)
    gpix = torch.randn(k, 3, inp["H"], inp["W"], generator=g).to(DEV)
    leaf = lambda t: t.to(DEV).contiguous().requires_grad_(True)
    # what is rendered: the images alone | with the fused L1 loss against targets | the RGB + silhouette pair (constant second colours)
    mode = ("plain", "l1", "pair")[int(rng.integers(3))]
    aa = bool(rng.integers(4) == 0)                                      # branch dr_aa's opacity compensation (preprocess fwd / bwd)
    targets = torch.rand(k, 3, inp["H"], inp["W"], generator=g).to(DEV)
    colors2, bg2 = torch.rand(P, 3, generator=g).to(DEV), torch.rand(3, generator=g).to(DEV)
    gpix2 = torch.randn(k, 3, inp["H"], inp["W"], generator=g).to(DEV)

    def frame_geo(n, v):          # frame v's own geometry (a batch of frames) or the shared one
        t = base[n]
        if not per_view:
            return t
        return t * (1.0 + 0.04 * v) if n != "means3D" else t + 0.01 * v

    def single(v, lv):
        cam = batch_to_camera(batches[v], device=DEV)
        st = R.GaussianRasterizationSettings(
            image_height=inp["H"], image_width=inp["W"], tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0,
            viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center,
            prefiltered=False, debug=False, antialiasing=aa)
        args = (lv["means3D"], None, lv.get("shs"), lv.get("rgb"), lv["opacities"], lv.get("scales"), lv.get("rots"), lv.get("cov6"), st)
        if mode == "l1":
            out = R.rasterize_gaussians_l1(*args, targets[v].contiguous(), want_invdepth=False)
            return out[0], out[3]
        if mode == "pair":
            out = R.rasterize_gaussians_pair(*args, colors2, bg2, want_invdepth=False)
            return out[0], out[3]
        return R.rasterize_gaussians(*args, want_invdepth=False)[0], None

    def run_sequential():
        shared = {n: leaf(t) for n, t in base.items() if not (per_view and n in geo)}
        per, imgs = [], []
        for v in range(k):
            lv = dict(shared)
            if per_view:
                lv.update({n: leaf(frame_geo(n, v)) for n in geo})
            img, extra = single(v, lv)
            if mode == "l1":
                (extra / k).backward()                                  # the batch's loss is the mean over its k images
            elif mode == "pair":
                ((img * gpix[v]).sum() + (extra * gpix2[v]).sum()).backward()
            else:
                (img * gpix[v]).sum().backward()
            imgs.append(img.detach() if extra is None else (img.detach(), extra.detach()))
            per.append({n: lv[n].grad for n in geo} if per_view else None)
        grads = {n: t.grad for n, t in shared.items()}
        if per_view:
            grads.update({n: torch.stack([per[v][n] for v in range(k)]) for n in geo})
        torch.cuda.synchronize()
        return imgs, grads

    def run_batched():
        mine = {n: leaf(t) for n, t in base.items() if not (per_view and n in geo)}
        if per_view:
            mine.update({n: leaf(torch.stack([frame_geo(n, v) for v in range(k)])) for n in geo})
        cams = CameraBatch(k, inp["W"], inp["H"], device=DEV).set(batches)
        out = rasterize_gaussians_views(mine["means3D"], mine.get("shs"), mine.get("rgb"), mine["opacities"], mine.get("scales"),
                                        mine.get("rots"), mine.get("cov6"), cams, bg, sh_degree=deg, antialiasing=aa,
                                        l1_targets=targets if mode == "l1" else None, colors2=colors2 if mode == "pair" else None,
                                        bg2=bg2 if mode == "pair" else None)
        colors = out[0]
        if mode == "l1":
            out[2].backward()
            imgs = [(colors[v].detach(), out[2].detach()) for v in range(k)]
        elif mode == "pair":
            ((colors * gpix).sum() + (out[2] * gpix2).sum()).backward()
            imgs = [(colors[v].detach(), out[2][v].detach()) for v in range(k)]
        else:
            (colors * gpix).sum().backward()
            imgs = [colors[v].detach() for v in range(k)]
        torch.cuda.synchronize()
        return imgs, {n: t.grad for n, t in mine.items()}

    return (seed, name, P, k, use_sh, deg, M, from_sr, per_view, aa, mode), from_sr, per_view, geo, k, run_sequential, run_batched


@pytest.mark.parametrize("seed", range(int(os.environ.get("D3GA_VIEWS_FUZZ_N", "8"))))
def test_views_fuzz(seed):
    """Random batches through the view-batched operator against the single-view operator, view by view (a campaign:
    D3GA_VIEWS_FUZZ_N=400): k = 2..10 cameras (every group split of the looped per-Gaussian kernels, a second backward group beyond
    eight), a ragged number of Gaussians (the last wavefront partly filled: the generic SH staging), SH colours with active degree
    0..3 at 16 coefficients per Gaussian or fewer (the stride of the rows; 3 M not a multiple of four: the unstaged path) or
    precomputed colours, covariance given or from (scales, rotations), shared geometry or a batch of frames, cameras with their own
    field of view, some Gaussians moved behind the cameras, antialiasing on in a quarter of the cases; the images alone, with the fused L1 loss, or as the RGB + silhouette
    pair.  Images bit-identical (the loss to 3e-6).  Gradients: both sides form the same sums in
    another order, so the bar is 8x what the SEQUENTIAL side differs by from itself -- measured in the test, five more sequential runs
    -- or _order_bar where that is larger: random scenes with large splats from (scales, rotations) reach 4e-4 of the largest
    element between two runs of the same operator (tools/diag_fuzz_views.py: needles whose covariance Jacobian cancels), while a
    wiring error -- a view's gradient dropped, doubled or landed in another view's rows -- is O(1 / k)."""
    tag, from_sr, per_view, geo, k, run_sequential, run_batched = _fuzz_case(seed)
    imgs, ref = run_sequential()
    colors, mine = run_batched()
    mode = tag[-1]
    for v in range(k):
        if mode == "plain":
            assert torch.equal(colors[v], imgs[v]), (tag, v, float((colors[v] - imgs[v]).abs().max()))
        else:
            assert torch.equal(colors[v][0], imgs[v][0]), (tag, v)
            if mode == "pair":
                assert torch.equal(colors[v][1], imgs[v][1]), (tag, v, "second image")
    if mode == "l1":                                                    # the batch's loss = the mean of the views' losses
        want = float(sum(float(im[1]) for im in imgs)) / k
        assert abs(float(colors[0][1]) - want) <= 3e-6 * abs(want), (tag, float(colors[0][1]), want)
    again = [run_sequential()[1] for _ in range(5)]
    bar = _order_bar(from_sr)
    for n in ref:
        views = range(k) if (per_view and n in geo) else (None,)
        for v in views:
            pick = (lambda t: t) if v is None else (lambda t: t[v])
            a, b = pick(ref[n]), pick(mine[n])
            scale = float(a.abs().max())
            if scale == 0.0:
                assert float(b.abs().max()) == 0.0, (tag, n, v)
                continue
            own = max(float((a - pick(r[n])).abs().max()) for r in again)
            allowed = max(bar * scale, 8.0 * own)             # (the spread is heavy-tailed: one needle decides the maximum)
            assert allowed <= 2e-2 * scale, (tag, n, v, own / scale)                 # (the operator itself: never that loose)
            assert float((a - b).abs().max()) <= allowed, (tag, n, v, float((a - b).abs().max()) / scale, own / scale)


def test_batched_capacity_overflow_is_detected_and_retried():
    """A cold high-water mark and footprints far beyond 4 P duplicates per view: the batched forward overflows its first capacity,
    reads the count back and runs again -- images and gradients as the single-view operator's; with a static, too small capacity
    the overflow is flagged (lists truncated), nothing faults."""
    from d3ga_amd import rasterizer as R
    inp = scene_inputs("T1", scale_mult=8.0)
    k = 3
    batches = _batches(inp, k)
    bg = torch.tensor([0.2, 0.5, 0.8], device=DEV)
    gpix = torch.randn(k, 3, inp["H"], inp["W"], generator=torch.Generator().manual_seed(4)).to(DEV)
    ref = _leaves(inp, False, False)
    imgs = []
    for v, b in enumerate(batches):
        img, _ = _single_view(inp, b, ref, bg, False, False)
        (img * gpix[v]).sum().backward()
        imgs.append(img.detach())
    R._hwm.clear()                                           # the batched call starts from k (4 P + 1024)
    mine = _leaves(inp, False, False)
    colors, _ = _views(inp, batches, mine, bg, False, False)
    cnt = R.last_counters()
    assert cnt["D"] > k * (4 * inp["means3D"].shape[0] + 1024), "scene too small to exercise the retry"
    assert not cnt["overflow"]
    (colors * gpix).sum().backward()
    torch.cuda.synchronize()
    for v in range(k):
        assert torch.equal(colors[v], imgs[v]), v
    for key in ref:
        a, b = ref[key].grad, mine[key].grad
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale, (key, float((a - b).abs().max()) / scale)
    R.set_capacity_policy("static", 3000)
    try:
        with torch.no_grad():
            _views(inp, batches, {n: t.detach() for n, t in mine.items()}, bg, False, False)
        torch.cuda.synchronize()
        assert R.last_counters()["overflow"]
    finally:
        R.set_capacity_policy("auto")


def test_batched_nan_and_inf_inputs_are_contained():
    """The containment of tests/test_gpu_parity.py::test_nan_and_inf_inputs_are_contained for a batch of views: a few Gaussians with
    NaN / Inf positions, covariances or opacities are culled in every view (the looped per-Gaussian kernels walk them view by view):
    finite images, bit-identical to the single-view renders of the same poisoned inputs, finite gradients for everybody else."""
    inp = scene_inputs("T1", scale_mult=3.0)
    k = 4
    batches = _batches(inp, k, fov_jitter=True)
    bg = torch.tensor([0.2, 0.3, 0.4], device=DEV)
    bad = torch.tensor([3, 50, 117, 400, 801], device=DEV)
    for t, (i, col, val) in zip(("means3D", "means3D", "cov6", "cov6", "opacities"),
                                ((0, None, "nan"), (1, 2, "inf"), (2, None, "nan"), (3, 0, "inf"), (4, None, "nan"))):
        x = inp[t]
        if col is None:
            x[int(bad[i])] = float(val)
        else:
            x[int(bad[i]), col] = float(val)
    ref = _leaves(inp, True, False)
    imgs = []
    for b in batches:
        img, _ = _single_view(inp, b, ref, bg, True, False)
        img.sum().backward()
        imgs.append(img.detach())
    mine = _leaves(inp, True, False)
    colors, radii = _views(inp, batches, mine, bg, True, False)
    colors.sum().backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(colors).all())
    good = torch.ones(inp["means3D"].shape[0], dtype=torch.bool, device=DEV)
    good[bad] = False
    for v in range(k):
        assert torch.equal(colors[v], imgs[v]), v
        assert int(radii[v, bad[0]]) == 0 and int(radii[v, bad[2]]) == 0
    for key in ("means3D", "cov6", "opacities", "shs"):
        a, b = ref[key].grad[good], mine[key].grad[good]
        assert bool(torch.isfinite(b).all()), key
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale, (key, float((a - b).abs().max()) / scale)


def test_batched_views_at_full_c3_size():
    """BASELINE configs[2] through the batched operator: 500k Gaussians, 1920 x 1080, SH degree 3, k = 3 cameras in one grid per stage
    (1.5 M records, 24 480 tiles, ~5 M duplicates: the list-sort classes and the heavy-tile split of the backward at their real
    sizes) -- every image bit-identical to the single-view render of that camera, the gradients the sum over the views."""
    inp = scene_inputs("C3")
    k = 3
    batches = _batches(inp, k, fov_jitter=True)
    bg = torch.tensor([1.0, 1.0, 1.0], device=DEV)
    g = torch.Generator().manual_seed(8)
    ref = _leaves(inp, True, False)
    imgs, seeds = [], []
    for v, b in enumerate(batches):
        gp = torch.randn(3, inp["H"], inp["W"], generator=g).to(DEV)
        img, _ = _single_view(inp, b, ref, bg, True, False)
        (img * gp).sum().backward()
        imgs.append(img.detach())
        seeds.append(gp)
    mine = _leaves(inp, True, False)
    colors, radii = _views(inp, batches, mine, bg, True, False)
    (colors * torch.stack(seeds)).sum().backward()
    torch.cuda.synchronize()
    for v in range(k):
        assert torch.equal(colors[v], imgs[v]), (v, float((colors[v] - imgs[v]).abs().max()))
        assert int((radii[v] > 0).sum()) > 400_000
    for key in ref:
        a, b = ref[key].grad, mine[key].grad
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale, (key, float((a - b).abs().max()) / scale)


@pytest.mark.parametrize("persistent_acc", [False, True])
def test_captured_batched_step_replays_with_other_cameras(persistent_acc):
    """The view-batched training step (LBS + cage deform once, k cameras in one grid per stage, mean L1 over the k images, the whole
    backward -- bench.py: batched_views) as ONE hipGraph: every replay, with the cameras of its CameraBatch rewritten in place and
    new targets copied into the static buffer, leaves the loss and the gradients of the same step run eagerly; with the
    accumulator kept between backwards (self-clearing) and allocated per backward."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from d3ga_amd import rasterizer as R
    from d3ga_amd import synthetic as syn
    from d3ga_amd.graph import CapturedStep
    from d3ga_amd.raster_views import CameraBatch
    from d3ga_amd.renderer import render_views
    dev = torch.device(DEV)
    frame = bench.Frame("T1", dev, view_index=0)
    k, W, H = 3, frame.wl.width, frame.wl.height
    cams_of = lambda r: [syn.make_batch(W, H, azimuth=0.3 + 0.7 * r + 2 * math.pi * v / 8, camera_id=v, fill=0.8 + 0.05 * v) for v in range(k)]
    cams = CameraBatch(k, W, H, device=dev).set(cams_of(0))
    targets = torch.rand(k, 3, H, W, generator=torch.Generator().manual_seed(0)).to(dev)
    params = list(frame.params.values())
    one = torch.ones((), device=dev)

    def step():
        loss = render_views(None, frame.upstream(), frame.bg, targets=targets, cameras=cams)["l1"]
        loss.backward(one)
        return loss

    def zero():
        for p in params:
            p.grad = None
    R.set_accumulator_policy("persistent" if persistent_acc else "fresh")
    try:
        zero(); step()
        R.set_capacity_policy("static", int(R.last_counters()["D"] * 2.0) + 4096)
        zero()
        graph = CapturedStep(step, params=params, check_every=1)
        for r in range(1, 5):
            cams.set(cams_of(r))
            targets.copy_(torch.rand(k, 3, H, W, generator=torch.Generator().manual_seed(r)).to(dev))
            loss_g = graph.replay()
            torch.cuda.synchronize()
            static = [p.grad for p in params]                        # the graph's own gradient tensors: every replay writes THESE
            got, lg = [g.clone() for g in static], float(loss_g.detach())
            zero()
            le = float(step())
            torch.cuda.synchronize()
            assert abs(lg - le) <= 1e-6 * abs(le), (r, lg, le)
            for p, g in zip(params, got):
                scale = float(p.grad.abs().max())
                assert scale > 0 and float((p.grad - g).abs().max()) <= 1e-5 * scale, (r, float((p.grad - g).abs().max()) / scale)
            for p, g in zip(params, static):        # (give them back: the eager step left tensors of its own in .grad)
                p.grad = g
        assert graph.check_overflow()["D"] > 0
    finally:
        R.set_capacity_policy("auto")
        R.set_accumulator_policy("fresh")

"""BASELINE configs[3] on the single-GPU leg: 100 optimisation steps of the reference's training step at the actor02-shaped
size (C4: 135k Gaussians, 747x1022) through the product path.  Needs a real MI355X.

The step is models/trainer.py:91-140 + 186-192 with train.py:190-203's losses: the avatar package from the cage deform with the
DeformationField / CanonicalField networks in front (models/cage_net.py:197-230), TWO renders (RGB over the background, then the
silhouette pass with a constant colour on black), (1 - lambda) L1 + lambda (1 - SSIM) on the RGB image + L1 on the silhouette
+ the scale regulariser, `clip_grad_norm_(parameters, 2.5)`, Adam.  Asserted: the loss falls by >= 20 %, every loss and
gradient stays finite, the binning capacity never overflows, and a CapturedStep of the same step captured AFTER the 100 in-place
optimizer updates (split-weight cache, parameter versions) replays to the eager step's loss and gradients."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
sys.path.insert(0, ROOT)


def _reachable_targets(frame):
    """Targets an optimiser can reach: the render (and the coverage) of a perturbed copy of the avatar (tools/soak.py)."""
    from d3ga_amd.cage_deform import cage_deform, lbs_cage
    from d3ga_amd.renderer import render
    with torch.no_grad():
        p = frame.params
        g = torch.Generator(device=DEV).manual_seed(5)
        tp = lbs_cage(frame.canon, p["delta_node"] + 0.01 * torch.randn(p["delta_node"].shape, device=DEV, generator=g), frame.joint_mats,
                      frame.skin_idx, frame.skin_w)
        m, c = cage_deform(tp, frame.tetras, frame.tetra_id, frame.barys0, frame.canon_grad, p["scaling"] + 0.1, p["rotation"],
                           scale_activation="exp")
        pkg = {"means3D": m, "cov3D_precomp": c, "opacities": torch.sigmoid(p["opacity"] + 0.5), "shs": p["features"] * 0.8, "rgb": None,
               "sh_degree": frame.sh_degree}
        frame.target = render(frame.batch, pkg, frame.bg)["render"].clone()
        frame.sil_rgb = torch.ones(m.shape[0], 3, device=DEV)
        frame.bg0 = torch.zeros_like(frame.bg)
        frame.sil_target = render(frame.batch, pkg, frame.bg0, colors_precomp=frame.sil_rgb)["render"].clone()


def test_c4_hundred_training_steps_then_a_captured_step():
    import bench
    from d3ga_amd import rasterizer as R
    from d3ga_amd.graph import CapturedStep
    torch.manual_seed(0)
    frame = bench.Frame("C4", torch.device(DEV), 0)
    _reachable_targets(frame)
    step = lambda: frame.train_step(with_fields=True, pair=False, scale_weight=175.0)       # two render() calls per step
    step()                                                                                  # creates the field networks
    # (with the networks in front, the free per-vertex / per-Gaussian offsets of the network-less variant are unused)
    params = [q for q in list(frame.params.values()) + frame.field_params if q.grad is not None]
    opt = torch.optim.Adam(params, lr=1e-3)
    losses, dmax = [], 0
    for it in range(100):
        opt.zero_grad(set_to_none=True)
        loss = step()
        gn = torch.nn.utils.clip_grad_norm_(params, 2.5, foreach=True)                      # models/trainer.py:188
        opt.step()
        if it % 10 == 0 or it == 99:
            cnt = R.last_counters()                                                         # synchronises
            dmax = max(dmax, cnt["D"])
            assert not cnt["overflow"], cnt
            assert torch.isfinite(loss) and torch.isfinite(gn), (it, float(loss), float(gn))
            assert all(torch.isfinite(q).all() for q in params), it
            losses.append(float(loss.detach()))
    print("C4 soak: loss", [round(v, 5) for v in losses], "D max", dmax)
    assert losses[-1] <= 0.8 * losses[0], losses

    # the same step, captured after the optimizer's in-place updates, against the eager step at the trained parameters
    # (no autograd graph of an eager step may be alive across a capture: d3ga_amd/graph.py)
    del loss, gn
    R.set_capacity_policy("static", int(1.25 * dmax) + 4096)
    try:
        opt.zero_grad(set_to_none=True)
        l_eager = float(step().detach())
        g_eager = [q.grad.clone() for q in params]
        opt.zero_grad(set_to_none=True)
        cap = CapturedStep(step, params=params, check_every=1)
        for _ in range(3):
            l_cap = cap.replay()
        info = cap.check_overflow()
        assert info is not None and info["D"] <= info["capacity"] and info["D"] > 0
        assert abs(float(l_cap.detach()) - l_eager) <= 1e-5 * abs(l_eager) + 1e-7
        for q, ge in zip(params, g_eager):
            ref = ge.abs().max() + 1e-30
            assert float((q.grad - ge).abs().max() / ref) < 2e-3          # float atomics: the summation order differs between launches
    finally:
        R.set_capacity_policy("auto")


def test_captured_step_reports_a_capacity_overflow():
    """A step captured with too small a binning capacity must say so (round 2: silently truncated tile lists)."""
    from d3ga_amd import rasterizer as R
    from d3ga_amd.graph import CapacityOverflowError, CapturedStep
    from d3ga_amd.renderer import render
    from util import scene_inputs
    inp = scene_inputs("T1", scale_mult=3.0)
    means = inp["means3D"].to(DEV).requires_grad_(True)
    pkg = {"means3D": means, "cov3D_precomp": inp["cov6"].to(DEV), "opacities": inp["opacities"].to(DEV), "shs": None,
           "rgb": inp["rgb"].to(DEV), "sh_degree": 0}
    bg = torch.ones(3, device=DEV)

    def step():
        img = render(inp["batch"], pkg, bg)["render"]
        img.sum().backward()
        return img.detach()
    step()
    d = R.last_counters()["D"]
    assert d > 2000
    R.set_capacity_policy("static", d // 2)
    try:
        means.grad = None
        cap = CapturedStep(step, params=(means,), check_every=2)
        cap.replay()                                               # (capturing runs nothing: the counters exist after a replay)
        with pytest.raises(CapacityOverflowError, match="exceed the binning capacity"):
            cap.check_overflow()
        with pytest.raises(CapacityOverflowError):                 # and without being asked: within 2 x check_every replays
            for _ in range(6):
                cap.replay()
                torch.cuda.synchronize()
    finally:
        R.set_capacity_policy("auto")
    # with room to spare nothing is raised
    R.set_capacity_policy("static", 2 * d)
    try:
        means.grad = None
        cap = CapturedStep(step, params=(means,), check_every=1)
        for _ in range(4):
            cap.replay()
        assert cap.check_overflow()["D"] == d
    finally:
        R.set_capacity_policy("auto")

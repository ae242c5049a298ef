"""Shared helpers for the parity tests: seeded scenes, oracle evaluation, comparison metrics."""
import numpy as np
import torch

from d3ga_amd import synthetic as syn
from oracle import camera as oc
from oracle import deform as od


def rel_err(a, b):
    """max |a-b| / max |b|  (the gradient-parity metric: <= 1e-3 per BASELINE.md)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def scene_inputs(name="T0", seed=17, azimuth=0.4, scale_mult=1.0, cx=None, cy=None, width=None, height=None):
    """CPU tensors for one frame: oracle-deformed Gaussians + camera of a synthetic workload."""
    sc = syn.make_scene(name, seed=seed)
    wl = sc["workload"]
    tp = od.lbs_cage(sc["canon_points"], sc["delta_node"], sc["joint_mats"], sc["skin_idx"], sc["skin_w"])
    cg = od.canonical_gradient(sc["canon_points"], sc["tetras"].long(), sc["tetra_id"].long())
    scales = torch.exp(sc["scaling"]) * scale_mult
    means, cov6 = od.cage_deform(tp, sc["tetras"], sc["tetra_id"], sc["barys"], cg, scales, sc["rotation"])
    batch = syn.make_batch(width or wl.width, height or wl.height, azimuth=azimuth, cx=cx, cy=cy)
    cam = oc.camera(batch["R"], batch["T"], batch["FoVx"], batch["FoVy"])
    return dict(
        scene=sc, batch=batch, cam=cam, W=batch["width"], H=batch["height"],
        tetpoints=tp, canon_grad=cg, scales=scales,
        means3D=means.contiguous(), cov6=cov6.contiguous(), opacities=torch.sigmoid(sc["opacity_logit"]),
        shs=torch.cat([sc["features_dc"], sc["features_rest"]], 1).contiguous(), rgb=sc["rgb"],
        view=torch.from_numpy(cam["world_view_transform"]), proj=torch.from_numpy(cam["full_proj_transform"]),
        campos=torch.from_numpy(cam["camera_center"]),
    )


def image_close(img, ref, atol=1e-4, outlier_frac=1e-4, outlier_atol=8e-3):
    """RGB parity: max-abs <= atol, except for a vanishing fraction of pixels where a 1-ulp difference in exp()
    flips an `alpha < 1/255` or `T < 1e-4` decision (each flip moves a pixel by at most alpha*T <= 1/255 * T)."""
    d = np.abs(np.asarray(img, np.float64) - np.asarray(ref, np.float64))
    bad = d > atol
    frac = bad.mean()
    return bool(frac <= outlier_frac and d.max() <= outlier_atol), float(d.max()), float(frac)


def grad_close(a, b, rtol=1e-3, outlier_frac=2e-5, outlier_rtol=5e-2):
    """Gradient parity at FULL size: per-Gaussian max |a-b| <= rtol * max|b| for all but a vanishing fraction of
    Gaussians.  A pixel whose `alpha < 1/255` / `T < 1e-4` decision flips (1-ulp exp difference, see image_close)
    moves the gradient of the few Gaussians on that pixel by one pixel's worth; with ~2e8 (pixel, Gaussian) pairs per
    frame a handful of such flips is unavoidable between any two exp implementations."""
    a = np.asarray(a, np.float64).reshape(len(b), -1)
    b = np.asarray(b, np.float64).reshape(len(b), -1)
    scale = np.abs(b).max() + 1e-30
    d = np.abs(a - b).max(1) / scale
    frac = float((d > rtol).mean())
    return bool(frac <= outlier_frac and d.max() <= outlier_rtol), float(d.max()), frac

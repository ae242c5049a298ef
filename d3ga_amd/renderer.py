"""Drop-in for the reference's renderer.py: `render(batch, pkg, bg_color, colors_precomp=None, measure_time=False,
solid_bg=True, fast=False, detach=[]) -> {"render": (3,H',W')}` (renderer.py:69-145), on the MI355X rasterizer.
"""
import torch

from .cameras import batch_to_camera
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

bg_colors = {"white": (1.0, 1.0, 1.0), "black": (0.0, 0.0, 0.0)}


_zeros = {}


def _zero_leaf(like):
    key = (like.device, tuple(like.shape))
    z = _zeros.get(key)
    if z is None:
        if len(_zeros) > 16:
            _zeros.clear()
        z = _zeros[key] = torch.zeros(like.shape, dtype=torch.float32, device=like.device)
    return z.detach().requires_grad_(True)


def paste(img, crop):
    """Undo the symmetric-FoV padding of lib/batch.py:186-198 (renderer.py:36-47)."""
    left_w, right_w, top_h, bottom_h, W, H = crop[0], crop[1], crop[2], crop[3], int(crop[4]), int(crop[5])
    # identity crops (centred principal point) are skipped: a no-op slice still costs a zero-fill + copy of the whole
    # image in its backward
    if W < img.shape[2]:
        img = img[:, :, :W] if left_w > right_w else img[:, :, -W:]
    if H < img.shape[1]:
        img = img[:, :H, :] if top_h > bottom_h else img[:, -H:, :]
    return img


def render(batch, pkg, bg_color, colors_precomp=None, measure_time=False, solid_bg=True, fast=False, detach=[],
           grad_sync=None):
    means3D = pkg["means3D"]
    cam = batch_to_camera(batch, device=means3D.device)
    crop = batch["crop"]

    settings = GaussianRasterizationSettings(
        image_height=int(batch["height"]),
        image_width=int(batch["width"]),
        tanfovx=cam.tanfovx,
        tanfovy=cam.tanfovy,
        bg=bg_color,
        scale_modifier=1.0,
        viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform,
        sh_degree=pkg["sh_degree"] if "sh_degree" in pkg else 0,
        campos=cam.camera_center,
        prefiltered=False,
        debug=False,
        antialiasing=False,
    )

    cov3D_precomp = pkg.get("cov3D_precomp")
    scales = pkg.get("scales")
    rotations = pkg.get("rotations")
    opacities = pkg["opacities"]
    shs = pkg["shs"]

    if len(detach) > 0:
        if "position" in detach:
            means3D = means3D.detach()
        if "covariance" in detach:
            cov3D_precomp = cov3D_precomp.detach()
        if "opacity" in detach:
            opacities = opacities.detach()

    if colors_precomp is None:
        colors_precomp = pkg["rgb"]
        if shs is not None:
            colors_precomp = None
    else:
        shs = None

    # screen-space points: a zero tensor whose .grad receives dL/d(mean2D) (renderer.py:122-128).  A fresh leaf over a
    # cached block of zeros: no fill kernel per render (the rasterizer never reads or writes its values)
    means2D = _zero_leaf(means3D)
    try:
        means2D.retain_grad()
    except Exception:
        pass

    rasterizer = GaussianRasterizer(raster_settings=settings)
    if grad_sync is not None:                       # extension over upstream's constructor: set only when asked for
        rasterizer.grad_sync = grad_sync
    if measure_time:
        torch.cuda.synchronize()
    rendered = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp,
                          opacities=opacities, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)[0]
    if measure_time:
        torch.cuda.synchronize()
    return {"render": paste(rendered, crop)}

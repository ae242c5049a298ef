"""Drop-in for the reference's renderer.py: `render(batch, pkg, bg_color, colors_precomp=None, measure_time=False,
solid_bg=True, fast=False, detach=[]) -> {"render": (3,H',W')}` (renderer.py:69-145), on the MI355X rasterizer.
"""
import torch

from .cameras import batch_to_camera
from .rasterizer import GaussianRasterizationSettings, rasterize_gaussians, rasterize_gaussians_l1, rasterize_gaussians_pair

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


def render_pair(batch, pkg, bg_color, colors2, bg_color2, grad_sync=None):
    """The two renders of the reference's training step (models/trainer.py:102-110: `render(frame, pkg, bg)` and
    `render(frame, pkg, colors_precomp=pkg["silhouette_rgb"], bg_color=zeros)`) from ONE pass over the same geometry:
    -> {"render": (3,H',W'), "render2": (3,H',W')}.  Same images and the same summed gradients as the two calls (colors2
    is treated as constant, as the reference's silhouette colours are); use the two calls when `detach` differs."""
    out = render(batch, pkg, bg_color, grad_sync=grad_sync, _pair=(colors2, bg_color2))
    return out


def render_l1(batch, pkg, bg_color, target, grad_sync=None):
    """`render(batch, pkg, bg_color)` and `l1_loss(render, target)` (utils/loss_utils.py:29, train.py:190) from one operator:
    -> {"render": (3,H',W'), "l1": scalar}.  Same image, same loss, same gradients as the two calls; the loss gradient is
    formed inside the compositing backward instead of travelling through a (3,H,W) gradient image.  `target`: a tensor of
    the render's shape or a `graph.TensorSlot`.  With an off-centre crop (lib/batch.py:186-198: the loss lives on the cropped
    window, not on the raster) the two calls are made instead."""
    crop = batch["crop"]
    if int(crop[4]) != int(batch["width"]) or int(crop[5]) != int(batch["height"]):
        from .losses import l1_loss
        img = render(batch, pkg, bg_color, grad_sync=grad_sync)["render"]
        return {"render": img, "l1": l1_loss(img, target)}
    return render(batch, pkg, bg_color, grad_sync=grad_sync, _l1=target)


def render_views(batches, pkg, bg_color, targets=None, cameras=None, colors2=None, bg_color2=None, grad_sync=None):
    """k views in one pass (extension; the reference renders one camera per call and averages the losses of a batch of frames,
    train.py:218-221): -> {"render": (k,3,H,W)}; with targets (k,3,H,W) also "l1" = the mean over the views of `l1_loss(render,
    target)` (its gradient formed inside the compositing backward); with colors2 (P,3) + bg_color2 also "render2" (k,3,H,W), the
    reference's silhouette pass (models/trainer.py:102-110) from the same pass.
    pkg: one package seen from k CAMERAS -- or a LIST of k packages, one per FRAME of the batch (the avatar deformed per pose:
    their means3D and covariances are stacked to (k,P,.)); appearance (opacities, shs | rgb) is taken from the first package and
    must be the same tensors in all of them.  Every image equals `render(batch_v, pkg_v, bg_color)["render"]`, the gradients equal
    the sum over the k calls (d3ga_amd/raster_views.py).  The views share the raster size and must not be cropped
    (lib/batch.py:186-198: centred principal point).  cameras: a `raster_views.CameraBatch` to reuse (a captured step keeps one and
    calls `cameras.set(batches)` before every replay); batches may then be None."""
    from .raster_views import CameraBatch, rasterize_gaussians_views
    frames = pkg if isinstance(pkg, (list, tuple)) else None
    if frames is not None:
        first = frames[0]
        for f in frames[1:]:
            for key in ("opacities", "opacity_logits", "shs", "rgb"):
                if f.get(key) is not first.get(key):
                    raise ValueError(f"render_views: the frames of a batch share their appearance; `{key}` differs between the packages")
        stack = lambda key: None if first.get(key) is None else torch.stack([f[key] for f in frames])
        pkg = dict(first, means3D=stack("means3D"), cov3D_precomp=stack("cov3D_precomp"), scales=stack("scales"), rotations=stack("rotations"))
    means3D = pkg["means3D"]
    if cameras is None:
        for b in batches:
            c = b["crop"]
            if int(c[4]) != int(b["width"]) or int(c[5]) != int(b["height"]):
                raise ValueError("render_views: a view with an off-centre crop -- use render() per view")
        cameras = CameraBatch(len(batches), int(batches[0]["width"]), int(batches[0]["height"]), device=means3D.device).set(batches)
    opacities, act = pkg.get("opacities"), None
    if opacities is None and pkg.get("opacity_logits") is not None:
        opacities, act = pkg["opacity_logits"], "sigmoid"
    shs = pkg["shs"]
    out = rasterize_gaussians_views(means3D, shs, None if shs is not None else pkg["rgb"], opacities, pkg.get("scales"),
                                    pkg.get("rotations"), pkg.get("cov3D_precomp"), cameras, bg_color,
                                    sh_degree=pkg["sh_degree"] if "sh_degree" in pkg else 0, opacity_activation=act, l1_targets=targets,
                                    colors2=colors2, bg2=bg_color2, grad_sync=grad_sync)
    if colors2 is not None:
        return {"render": out[0], "render2": out[2]}
    return {"render": out[0], "l1": out[2]} if targets is not None else {"render": out[0]}


def render(batch, pkg, bg_color, colors_precomp=None, measure_time=False, solid_bg=True, fast=False, detach=[],
           grad_sync=None, _pair=None, _l1=None):
    means3D = pkg["means3D"]
    # a cameras.CameraSlot in the batch: the camera is read from its static device buffer (graph-capturable step that
    # follows the trainer's camera-per-step, d3ga_amd/graph.py)
    cam = batch.get("camera_slot") or batch_to_camera(batch, device=means3D.device)
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

    # what the rasterizer is handed (renderer.py:95-120): geometry as packaged; `detach` names inputs whose gradient is cut
    # ("position", "covariance", "opacity": the silhouette pass of models/trainer.py:104-110); the colour is the explicit
    # `colors_precomp` if given, else the package's SH coefficients, else its RGB
    geo = {k: pkg.get(k) for k in ("cov3D_precomp", "scales", "rotations", "opacities")}
    # extension: a package may carry the raw `opacity_logits` instead of activated `opacities` (models/cage_net.py:247
    # applies sigmoid in Python): the activation then runs inside the per-Gaussian kernels, forward and backward
    act = None
    if geo["opacities"] is None and pkg.get("opacity_logits") is not None:
        geo["opacities"], act = pkg["opacity_logits"], "sigmoid"
    cut = {"position": "means3D", "covariance": "cov3D_precomp", "opacity": "opacities"}
    geo["means3D"] = means3D
    for name in detach:
        if name in cut:
            geo[cut[name]] = geo[cut[name]].detach()
    means3D, cov3D_precomp, scales, rotations, opacities = (geo[k] for k in ("means3D", "cov3D_precomp", "scales", "rotations", "opacities"))
    shs = pkg["shs"] if colors_precomp is None else None
    if colors_precomp is None and shs is None:
        colors_precomp = pkg["rgb"]

    # screen-space points: a zero tensor whose .grad receives dL/d(mean2D) (renderer.py:122-128).  A fresh leaf over a
    # cached block of zeros: no fill kernel per render (the rasterizer never reads or writes its values)
    means2D = _zero_leaf(means3D)
    try:
        means2D.retain_grad()
    except Exception:
        pass

    if _pair is not None:
        img, _radii, _invd, img2 = rasterize_gaussians_pair(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                                            cov3D_precomp, settings, _pair[0], _pair[1], grad_sync, act,
                                                            want_invdepth=False)
        return {"render": paste(img, crop), "render2": paste(img2, crop)}
    if _l1 is not None:
        img, _radii, _invd, loss = rasterize_gaussians_l1(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                                          cov3D_precomp, settings, _l1, grad_sync, act, want_invdepth=False)
        return {"render": img, "l1": loss}
    # the operator behind upstream's `GaussianRasterizer(raster_settings)(...)` module call (renderer.py:130-141), without
    # building an nn.Module per render: its constructor and attribute writes cost ~25 us of host time per call, a sixth of an
    # eager render's enqueue time (tools/prof_host.py); the module class stays available for callers that use it themselves
    if measure_time:
        torch.cuda.synchronize()
    rendered = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings,
                                   grad_sync, act, want_invdepth=False)[0]     # only [0] of the rasterizer's outputs is used here (renderer.py:141)
    if measure_time:
        torch.cuda.synchronize()
    return {"render": paste(rendered, crop)}

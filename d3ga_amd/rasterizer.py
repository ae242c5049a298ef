"""Differentiable tile rasterizer with the operator surface of `diff_gaussian_rasterization` (branch dr_aa).

Mirrors what renderer.py:13-16,79-93,130-141 of the reference imports and calls:
    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
                                  projmatrix, sh_degree, campos, prefiltered, debug, antialiasing)
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                                        rotations=None, cov3D_precomp=None) -> (color (3,H,W), radii (P,), invdepth (1,H,W))
backed by the gfx950 kernels of libd3ga_hip.so (include/d3ga.h).  GPU tensors only, no CPU fallback.

Scratch ("geomBuffer / binningBuffer / imgBuffer") is allocated from torch's caching allocator per call and kept
alive in the autograd context.  The duplicate capacity of the binning lists is a per-device high-water mark; after
the forward has been enqueued the 16-byte counter block is read back (the only host sync, the analogue of upstream
reading `num_rendered`) and the call is repeated with a larger buffer in the rare case of overflow.  With
``set_capacity_policy("static", n)`` nothing is read back (graph-capturable; caller checks `last_counters()`).
"""
import ctypes
import os
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib
from ._lib import RasterParams, check, dptr, require_cuda, stream_handle


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool = False


# ------------------------------------------------------------------------------------------------------------
# capacity policy for the binning lists
# ------------------------------------------------------------------------------------------------------------
_policy = {"mode": "auto", "static": 0}
_hwm = {}            # device index -> high-water mark of D
_last = {}           # device index -> (binning tensor, capacity) of the most recent forward (for last_counters)
_last_img = {}       # device index -> (img scratch tensor, W, H) of the most recent forward (for last_termination)
_capture_log = None  # while graph.CapturedStep / CapturedCutStep capture: list of (binning tensor, capacity) of EVERY forward issued


def set_capacity_policy(mode, capacity=0):
    """"auto": read the duplicate count back after every forward and retry on overflow (default).
    "static": use `capacity` duplicates, never synchronise; check `last_counters()["overflow"]` yourself."""
    if mode not in ("auto", "static"):
        raise ValueError(mode)
    _policy["mode"] = mode
    _policy["static"] = int(capacity)


# ------------------------------------------------------------------------------------------------------------
# screen-space gradient accumulator of the backward
# ------------------------------------------------------------------------------------------------------------
_acc_policy = {"persistent": False}
_acc_cache = {}      # (device index, P) -> zeroed (P, ACC_STRIDE) buffer kept between backwards


def set_accumulator_policy(mode):
    """"fresh" (default): every backward allocates its accumulator and clears it (a 64 B x P fill kernel).
    "persistent": ONE accumulator per (device, P) is kept between backwards; the per-Gaussian backward kernel leaves it all
    zero again (d3ga_raster_params.acc_self_clearing), so no clear is ever launched.  Valid when all rasterizer backwards
    of the process on that device are ordered on one stream (a training loop, or captured steps replayed on it) -- two
    backwards in flight on different streams would share the buffer."""
    if mode not in ("fresh", "persistent"):
        raise ValueError(mode)
    _acc_policy["persistent"] = mode == "persistent"
    if mode == "fresh":
        _acc_cache.clear()


# D3GA_L1_VALUE=separate: the L1 value from its own pass over the finished image (rounds 1-3; kept for A/B runs)
_l1_policy = {"fused_value": os.environ.get("D3GA_L1_VALUE", "fused") != "separate"}


def l1_mean_forward(image, target, cell, out, dev):
    """mean |image - target| into `out` (device scalar): one partial sum per workgroup, then one workgroup adds them (index
    order: reproducible)."""
    L = _lib.lib()
    ws = torch.empty(_lib.LOSS_PARTIALS, dtype=torch.float32, device=dev)
    if cell is None:
        check(L.d3ga_l1_mean_fwd_ws(image.numel(), dptr(image), dptr(target), dptr(out), dptr(ws), stream_handle()), "d3ga_l1_mean_fwd_ws")
    else:
        check(L.d3ga_l1_mean_fwd_ws_cell(image.numel(), dptr(image), dptr(cell), dptr(out), dptr(ws), stream_handle()),
              "d3ga_l1_mean_fwd_ws_cell")


def _accumulator(P, dev):
    """-> (buffer, self_clearing)"""
    if not _acc_policy["persistent"] or P == 0:
        return torch.empty((P, _lib.ACC_STRIDE), dtype=torch.float32, device=dev), False
    key = (dev.index, P)
    buf = _acc_cache.get(key)
    if buf is None:
        if len(_acc_cache) > 8:
            _acc_cache.clear()
        buf = _acc_cache[key] = torch.zeros((P, _lib.ACC_STRIDE), dtype=torch.float32, device=dev)
    return buf, True


def last_counters(device=None):
    """dict(D, overflow, max_tile, visible) of the most recent forward on `device` (synchronises)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    c = _last[dev][0][:32].view(torch.int32)[:4].cpu().tolist()
    return {"D": c[0] & 0xFFFFFFFF, "overflow": bool(c[1]), "max_tile": c[2] & 0xFFFFFFFF, "visible": c[3]}


class StageTimer:
    """Optional per-stage HIP-event timing of the rasterizer (bench.py).  While enabled the forward/backward are
    issued stage by stage through the C ABI (same kernels, same stream) with an event pair around each stage.

    `burst` (round 5): stage name -> R.  Such a stage is launched R times back to back between ITS two events and the
    elapsed time divided by R.  A single eager launch between two events on an idle queue also measures the dispatch
    latency of the kernel packet (BENCH_r04: compositing backward 121 us by the event pair against 107 us in the rocprofv3
    kernel trace of the same command); with the queue kept full the quotient is the kernel's own duration, which is what
    `roofline.achieved` is defined on.  Only for stages that may run twice on the same inputs (the compositing kernels:
    the forward rewrites the same outputs, the backward adds into an accumulator whose gradients nobody reads in this
    measuring pass)."""

    def __init__(self):
        self.enabled = False
        self.records = []          # (stage, start_event, end_event, launches between them)
        self.burst = {}

    def reset(self):
        self.records = []

    def stage(self, name, fn):
        if not self.enabled:
            return fn()
        reps = int(self.burst.get(name, 1))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()                  # on torch's current stream == the stream the kernels are launched on
        r = fn()
        for _ in range(reps - 1):
            fn()
        b.record()
        self.records.append((name, a, b, reps))
        return r

    def summary(self):
        """stage -> (launches, mean ms per launch); call after torch.cuda.synchronize().  An event pair also sees any gap in
        which the GPU waited for the host between the two records (a Python GC pause inside one eager launch shows up as a
        multi-millisecond "kernel"), so samples longer than 5x the stage's median are left out of the mean."""
        per = {}
        for name, a, b, reps in self.records:
            per.setdefault(name, []).append((a.elapsed_time(b) / reps, reps))
        out = {}
        for name, ts in per.items():
            med = sorted(t for t, _ in ts)[len(ts) // 2]
            kept = [(t, r) for t, r in ts if t <= 5.0 * med] or ts
            out[name] = (sum(r for _, r in kept), sum(t for t, _ in kept) / len(kept))
        return out


stage_timer = StageTimer()


_scratch_sizes = {}
_prm_cache = {}          # parameter blocks by value (never mutated: the backward copies one when it needs another acc_self_clearing)


def _scratch(P, W, H, cap, device, forward_only=False, binning=True):
    """(geom, binning, img) byte tensors.  forward_only: the img buffer without the per-block lists of the backward
    (128 B per duplicate of capacity -- several hundred MB per render at a few million duplicates); binning=False: None
    for the binning buffer (a geometry-cache hit shares the first pass's)."""
    key = (P, W, H, cap, bool(forward_only))
    sz = _scratch_sizes.get(key)
    if sz is None:                                   # (two ctypes calls per render otherwise: the sizes of a training loop never change)
        sizes = (ctypes.c_int64 * 3)()
        L = _lib.lib()
        check(L.d3ga_raster_scratch_bytes(P, W, H, cap, sizes), "d3ga_raster_scratch_bytes")
        img_bytes = int(L.d3ga_raster_img_bytes(W, H, cap, 1)) if forward_only else int(sizes[2])
        if len(_scratch_sizes) > 64:
            _scratch_sizes.clear()
        sz = _scratch_sizes[key] = (int(sizes[0]), int(sizes[1]), img_bytes)
    new = lambda n: torch.empty(n, dtype=torch.uint8, device=device)
    return [new(sz[0]), new(sz[1]) if binning else None, new(sz[2])]


def _f32(t, device):
    if t is None:
        return None
    if t.dtype != torch.float32 or t.device != device:
        t = t.to(device=device, dtype=torch.float32)
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------------------
# geometry reuse between consecutive renders of the same Gaussians from the same camera
# ------------------------------------------------------------------------------------------------------------
# The reference's training step renders the same package twice (RGB, then a silhouette pass with a constant colour,
# models/trainer.py:102-110).  Projection, tile histogram, scatter and the per-tile sort depend only on
# (means3D, covariance, opacities, camera, image size), so the second call copies the first call's geometry records,
# re-evaluates only the colour (d3ga_raster_recolor) and shares its binning buffer.  A call hits the cache when its
# inputs ARE the previous call's tensors (same storage address and version counter; the cache keeps them alive, so an
# address cannot be recycled for different data).  Never used while a stream capture is in progress.
#
# OPT-IN (off by default).  A hit is decided from (data_ptr, _version, shape) only, so a write that does not bump the
# version counter is invisible to it: `p.data.add_()` / `.data.copy_()` style updates, or a kernel (this library's own C
# ABI included) writing into a preallocated buffer.  With such inputs every later render would silently reuse stale
# projection and binning.  Enable it only around the two renders of one step (`with geometry_reuse():`) -- the cache is
# dropped on exit -- or use `renderer.render_pair`, which needs no cache at all.
_reuse = {"enabled": False}
_geom_cache = {}     # device index -> dict(key, geom, binning, cap, radii, pins)


def set_geometry_reuse(enabled):
    _reuse["enabled"] = bool(enabled)
    if not enabled:
        _geom_cache.clear()


def clear_geometry_cache():
    _geom_cache.clear()


class geometry_reuse:
    """`with geometry_reuse(): rgb = render(...); sil = render(...)` -- the second render of the same package reuses the
    first one's projection / binning / sort; nothing is pinned or trusted beyond the block."""

    def __enter__(self):
        self._was = _reuse["enabled"]
        _reuse["enabled"] = True
        return self

    def __exit__(self, *exc):
        _reuse["enabled"] = self._was
        _geom_cache.clear()
        return False


def _tkey(t):
    return None if t is None else (t.data_ptr(), t._version, tuple(t.shape))


def _geometry_key(prm, cap_mode, tensors):
    return (prm.P, prm.W, prm.H, prm.tanfovx, prm.tanfovy, prm.scale_modifier, prm.prefiltered, prm.opacity_activation, prm.antialiasing, cap_mode) + tuple(
        _tkey(t) for t in tensors)


def _empty_to_none(t):
    return None if (t is None or t.numel() == 0) else t


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, grad_sync=None, colors2=None, bg2=None, opacity_activation=None, l1_target=None,
                want_invdepth=True):
        s = raster_settings
        require_cuda(means3D)
        dev = means3D.device
        if means3D.shape[0] > 0:      # upstream passes absent arguments as empty tensors
            sh, colors_precomp, scales, rotations, cov3Ds_precomp = map(
                _empty_to_none, (sh, colors_precomp, scales, rotations, cov3Ds_precomp))
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp = (
            _f32(t, dev) for t in (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp))
        view, proj, campos, bg = (_f32(t, dev) for t in (s.viewmatrix, s.projmatrix, s.campos, s.bg))
        P = means3D.shape[0]
        dual = colors2 is not None and P > 0             # second image from the same pass (rasterize_gaussians_pair)
        empty_pair = colors2 is not None and P == 0      # nothing to blend: the second image is its background
        if colors2 is not None:
            colors2, bg2 = _f32(colors2.detach(), dev), _f32(bg2, dev)
        H, W = int(s.image_height), int(s.image_width)
        M = sh.shape[1] if sh is not None else 0
        tfx, tfy = float(s.tanfovx), float(s.tanfovy)
        if not (tfx > 0.0 and tfy > 0.0):
            # tanfovx <= 0 is the in-band marker of a camera slot (cameras.CameraSlot): the kernels then read both tangents
            # from campos[3], campos[4].  Anything else non-positive (or NaN) would make them read past a 3-float campos.
            if not (tfx == 0.0 and tfy == 0.0 and campos.numel() >= 5):
                raise ValueError(f"GaussianRasterizationSettings: tanfovx / tanfovy must be positive (got {tfx}, {tfy}); only a "
                                 "camera slot (tanfovx = tanfovy = 0 with a 5-float campos: cameras.CameraSlot) reads them from the device")
        # no input requires a gradient (inference, torch.no_grad): the forward skips the per-block lists of the backward
        fwd_only = not any(ctx.needs_input_grad[:8])
        # (the parameter block of a training loop repeats: building the ctypes structure costs ~6 us of an ~80 us host-bound render)
        pkey = (P, M, int(s.sh_degree), W, H, tfx, tfy, float(s.scale_modifier), bool(s.antialiasing), bool(s.prefiltered), bool(s.debug),
                opacity_activation, fwd_only)
        prm = _prm_cache.get(pkey)
        if prm is None:
            if len(_prm_cache) > 64:
                _prm_cache.clear()
            prm = _prm_cache[pkey] = RasterParams(P=P, M=M, sh_degree=int(s.sh_degree), W=W, H=H, tanfovx=tfx,
                                                  tanfovy=tfy, scale_modifier=float(s.scale_modifier),
                                                  antialiasing=int(bool(s.antialiasing)), prefiltered=int(bool(s.prefiltered)),
                                                  debug=int(bool(s.debug)), opacity_activation=_ACTIVATIONS[opacity_activation],
                                                  forward_only=int(fwd_only))
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        # fused L1 image loss (rasterize_gaussians_l1): the loss VALUE is formed by the compositing forward while the colours
        # are in registers (d3ga_raster_composite_fwd_l1: one partial per quadrant, then one small sum), its gradient inside
        # the compositing backward (d3ga_raster_backward_l1): no pass over the image, no (3,H,W) gradient image
        l1_t = l1_cell = loss = None
        if l1_target is not None:
            from .graph import TensorSlot
            l1_t = l1_target.current if isinstance(l1_target, TensorSlot) else _f32(l1_target, dev)
            l1_cell = l1_target.cell if isinstance(l1_target, TensorSlot) else None
            if tuple(l1_t.shape) != (3, H, W):
                raise ValueError(f"rasterize_gaussians_l1: the target must be (3, {H}, {W}), got {tuple(l1_t.shape)}")
            loss = torch.empty((), dtype=torch.float32, device=dev)
        l1_in_fwd = l1_target is not None and P > 0 and colors2 is None and _l1_policy["fused_value"]

        def composite_fwd_single(L, pp, geom, binning, cap, img, st):
            if l1_in_fwd:
                ws = torch.empty(4 * ((W + 15) // 16) * ((H + 15) // 16), dtype=torch.float32, device=dev)
                check(L.d3ga_raster_composite_fwd_l1(pp, dptr(bg), dptr(geom), dptr(binning), cap, dptr(img), dptr(color),
                                                     dptr(invdepth), dptr(None if l1_cell is not None else l1_t), dptr(l1_cell),
                                                     dptr(loss), dptr(ws), st), "d3ga_raster_composite_fwd_l1")
            else:
                check(L.d3ga_raster_composite_fwd(pp, dptr(bg), dptr(geom), dptr(binning), cap, dptr(img), dptr(color),
                                                  dptr(invdepth), st), "d3ga_raster_composite_fwd")
        # the inverse-depth image of branch dr_aa: on request only (renderer.render* use the colour alone, renderer.py:141)
        invdepth = torch.empty((1, H, W), dtype=torch.float32, device=dev) if want_invdepth else None
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        L = _lib.lib()
        static = _policy["mode"] == "static"
        cap = _policy["static"] if static else max(_hwm.get(dev.index, 0), 4 * P + 1024)
        geo_inputs = (means3D, opacities, scales, rotations, cov3Ds_precomp, view, proj)
        color2 = torch.empty((3, H, W), dtype=torch.float32, device=dev) if dual else None
        use_cache = _reuse["enabled"] and P > 0 and not dual and not torch.cuda.is_current_stream_capturing()
        key = _geometry_key(prm, ("static", cap) if static else "auto", geo_inputs) if use_cache else None
        hit = _geom_cache.get(dev.index) if use_cache else None
        if hit is not None and hit["key"] == key:
            cap, binning, radii = hit["cap"], hit["binning"], hit["radii"]
            geom, _unused, img = _scratch(P, W, H, cap, dev, fwd_only, binning=False)     # img carries the per-block lists: sized by cap
            st, pp = stream_handle(), ctypes.byref(prm)
            stage_timer.stage("recolor", lambda: check(L.d3ga_raster_recolor(
                pp, dptr(means3D), dptr(sh), dptr(colors_precomp), dptr(campos), dptr(hit["geom"]), dptr(geom), st),
                "d3ga_raster_recolor"))
            stage_timer.stage("composite_fwd", lambda: composite_fwd_single(L, pp, geom, binning, cap, img, st))
            _last[dev.index] = (binning, cap)
            if _capture_log is not None:
                _capture_log.append((binning, cap))
        while hit is None or hit["key"] != key:
            geom, binning, img = _scratch(P, W, H, cap, dev, fwd_only)
            if stage_timer.enabled or dual or l1_in_fwd:
                st, pp = stream_handle(), ctypes.byref(prm)
                stage_timer.stage("preprocess", lambda: check(L.d3ga_raster_preprocess(
                    pp, dptr(means3D), dptr(sh), dptr(colors_precomp), dptr(opacities), dptr(scales), dptr(rotations),
                    dptr(cov3Ds_precomp), dptr(view), dptr(proj), dptr(campos), dptr(geom), dptr(binning), cap,
                    dptr(radii), st), "d3ga_raster_preprocess"))
                stage_timer.stage("bin_sort", lambda: check(L.d3ga_raster_bin_sort(
                    pp, dptr(geom), dptr(binning), cap, st), "d3ga_raster_bin_sort"))
                if dual:
                    stage_timer.stage("composite_fwd", lambda: check(L.d3ga_raster_composite_fwd2(
                        pp, dptr(bg), dptr(bg2), dptr(geom), dptr(colors2), dptr(binning), cap, dptr(img), dptr(color),
                        dptr(color2), dptr(invdepth), st), "d3ga_raster_composite_fwd2"))
                else:
                    stage_timer.stage("composite_fwd", lambda: composite_fwd_single(L, pp, geom, binning, cap, img, st))
            else:
                check(L.d3ga_raster_forward(ctypes.byref(prm), dptr(means3D), dptr(sh), dptr(colors_precomp),
                                            dptr(opacities), dptr(scales), dptr(rotations), dptr(cov3Ds_precomp),
                                            dptr(view), dptr(proj), dptr(campos), dptr(bg), dptr(geom), dptr(binning),
                                            dptr(img), cap, dptr(color), dptr(radii), dptr(invdepth), stream_handle()),
                      "d3ga_raster_forward")
            _last[dev.index] = (binning, cap)
            if _capture_log is not None:
                _capture_log.append((binning, cap))
            if static:
                break
            cnt = binning[:32].view(torch.int32)[:2].cpu().tolist()       # host sync (upstream: num_rendered)
            D = cnt[0] & 0xFFFFFFFF
            _hwm[dev.index] = max(_hwm.get(dev.index, 0), int(D * 1.25) + 1024)
            if not cnt[1]:
                break
            cap = _hwm[dev.index]
        if use_cache and (hit is None or hit["key"] != key):
            _geom_cache[dev.index] = {"key": key, "geom": geom, "binning": binning, "cap": cap, "radii": radii,
                                      # detached aliases: they pin the storage without keeping an autograd graph alive
                                      "pins": tuple(None if t is None else t.detach() for t in geo_inputs)}
        ctx.prm = prm
        ctx.pkey = pkey
        ctx.cap = cap
        ctx.has_means2D = means2D is not None
        ctx.grad_sync = grad_sync if (grad_sync is not None and (grad_sync.world > 1 or getattr(grad_sync, "always", False)) and P > 0) else None
        if ctx.grad_sync is not None and hasattr(grad_sync, "verify_inputs"):
            # the cut exchange is only valid for view-independent inputs (dist.ViewShardedGrads): checked on the first call(s)
            grad_sync.verify_inputs({"means3D": means3D, "opacities": opacities, "colors_precomp": colors_precomp,
                                     "shs": sh, "cov3D_precomp": cov3Ds_precomp, "scales": scales, "rotations": rotations})
        ctx.dual = dual
        ctx.l1 = l1_target is not None and P > 0 and not dual
        if l1_target is not None and not l1_in_fwd:      # P == 0 (the image is its background) or D3GA_L1_VALUE=separate
            l1_mean_forward(color, l1_t, l1_cell, loss, dev)
        _last_img[dev.index] = (img, W, H, geom, P)
        ctx.save_for_backward(means3D, sh, scales, rotations, cov3Ds_precomp, view, proj, campos, bg, geom, binning, img,
                              colors2, bg2, color if ctx.l1 else None, l1_t if ctx.l1 else None, l1_cell if ctx.l1 else None)
        # (the inverse-depth image IS differentiable, as on branch dr_aa: with set_materialize_grads(False) an unused one costs nothing)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)                 # no zero-filled (P,) / (H,W) gradients for radii / invdepth per step
        if l1_target is not None:
            return color, radii, invdepth, loss
        if dual:
            return color, radii, invdepth, color2
        if empty_pair:
            return color, radii, invdepth, bg2.reshape(3, 1, 1).expand(3, H, W).contiguous()
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, _grad_invdepth, grad_color2=None):
        (means3D, sh, scales, rotations, cov3Ds_precomp, view, proj, campos, bg, geom, binning, img, colors2,
         bg2, image, l1_t, l1_cell) = ctx.saved_tensors
        prm, dev, P = ctx.prm, means3D.device, means3D.shape[0]
        dual = ctx.dual
        g_loss = None
        if ctx.l1:                                       # the 4th output was the fused L1 loss: its incoming gradient
            g_loss, grad_color2 = grad_color2, None
            if g_loss is not None:
                g_loss = _f32(g_loss, dev).reshape(1)
        g_invd = None
        if _grad_invdepth is not None:                   # a loss on the inverse-depth image (branch dr_aa's depth regularisation)
            if dual or ctx.l1:
                raise NotImplementedError("a gradient of the inverse-depth image is implemented for the single-image rasterizer call only "
                                          "(not for rasterize_gaussians_pair / rasterize_gaussians_l1)")
            g_invd = _f32(_grad_invdepth, dev).reshape(prm.H, prm.W)
        if grad_color is None and g_loss is None and g_invd is None:        # only the second image was used (or nothing at all)
            grad_color = torch.zeros((3, prm.H, prm.W), dtype=torch.float32, device=dev)
        grad_color = _f32(grad_color, dev)
        if dual:
            grad_color2 = (torch.zeros_like(grad_color) if grad_color2 is None else _f32(grad_color2, dev))
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        acc, self_clearing = _accumulator(P, dev)
        if self_clearing != bool(prm.acc_self_clearing):
            # (ctx.prm is shared -- a retained graph, the cache of parameter blocks: the variant is a block of its own, cached too)
            key2 = (ctx.pkey, self_clearing)
            p2 = _prm_cache.get(key2)
            if p2 is None:
                p2 = RasterParams(**{f: getattr(prm, f) for f, _ in RasterParams._fields_})
                p2.acc_self_clearing = int(self_clearing)
                _prm_cache[key2] = p2
            prm = p2
        from_sr = cov3Ds_precomp is None
        sync = ctx.grad_sync
        if sync is None:
            g_means3D, g_means2D, g_opac = new(P, 3), new(P, 3), new(P, 1)
            g_sh = new(P, prm.M, 3) if sh is not None else None
            g_col = new(P, 3) if sh is None else None
            g_cov = None if from_sr else new(P, 6)
            g_scales = new(P, 3) if from_sr else None
            g_rots = new(P, 4) if from_sr else None
        else:
            # view-sharded training: every gradient that leaves this op is summed over the ranks HERE, at the narrowest
            # cut of the graph (d3ga_amd/dist.py:ViewShardedGrads) -- one planar buffer for the all-reduce, and for the SH
            # path the (P,3) factor of the rank-1 SH gradient (+ this view's camera position as row P) for the all-gather
            g_means2D = new(P, 3)
            widths = [3, 1] + ([3, 4] if from_sr else [6]) + ([3] if sh is None else [])
            flat = new(P * sum(widths))
            parts, off = [], 0
            for w in widths:
                parts.append(flat[off:off + P * w].view(P, w))
                off += P * w
            g_means3D, g_opac = parts[0], parts[1]
            g_scales, g_rots = (parts[2], parts[3]) if from_sr else (None, None)
            g_cov = None if from_sr else parts[2]
            g_sh = None
            if sh is None:
                g_col, factor = parts[-1], None
            else:
                factor = new(P + 1, 3)
                g_col = factor[:P]
        L = _lib.lib()
        if stage_timer.enabled or dual or g_invd is not None:
            st, pp = stream_handle(), ctypes.byref(prm)
            if not self_clearing:
                acc.zero_()
            if dual:
                stage_timer.stage("composite_bwd", lambda: check(L.d3ga_raster_composite_bwd2(
                    pp, dptr(bg), dptr(bg2), dptr(geom), dptr(colors2), dptr(binning), ctx.cap, dptr(img), dptr(grad_color),
                    dptr(grad_color2), dptr(acc), st), "d3ga_raster_composite_bwd2"))
            elif g_invd is not None:
                stage_timer.stage("composite_bwd", lambda: check(L.d3ga_raster_composite_bwd_depth(
                    pp, dptr(bg), dptr(geom), dptr(binning), ctx.cap, dptr(img), dptr(grad_color), dptr(g_invd), dptr(acc), st),
                    "d3ga_raster_composite_bwd_depth"))
            elif g_loss is not None:
                stage_timer.stage("composite_bwd", lambda: check(L.d3ga_raster_composite_bwd_l1(
                    pp, dptr(bg), dptr(geom), dptr(binning), ctx.cap, dptr(img), dptr(image), dptr(l1_t), dptr(l1_cell),
                    dptr(g_loss), dptr(grad_color), dptr(acc), st), "d3ga_raster_composite_bwd_l1"))
            else:
                stage_timer.stage("composite_bwd", lambda: check(L.d3ga_raster_composite_bwd(
                    pp, dptr(bg), dptr(geom), dptr(binning), ctx.cap, dptr(img), dptr(grad_color), dptr(acc), st),
                    "d3ga_raster_composite_bwd"))
            stage_timer.stage("preprocess_bwd", lambda: check(L.d3ga_raster_preprocess_bwd(
                pp, dptr(means3D), dptr(sh), dptr(scales), dptr(rotations), dptr(cov3Ds_precomp), dptr(view), dptr(proj),
                dptr(campos), dptr(geom), dptr(acc), dptr(g_means3D), dptr(g_means2D), dptr(g_opac), dptr(g_sh),
                dptr(g_col), dptr(g_cov), dptr(g_scales), dptr(g_rots), st), "d3ga_raster_preprocess_bwd"))
        elif g_loss is not None:
            check(L.d3ga_raster_backward_l1(
                ctypes.byref(prm), dptr(means3D), dptr(sh), dptr(scales), dptr(rotations), dptr(cov3Ds_precomp),
                dptr(view), dptr(proj), dptr(campos), dptr(bg), dptr(geom), dptr(binning), ctx.cap, dptr(img),
                dptr(image), dptr(l1_t), dptr(l1_cell), dptr(g_loss), dptr(grad_color), dptr(acc), dptr(g_means3D),
                dptr(g_means2D), dptr(g_opac), dptr(g_sh), dptr(g_col), dptr(g_cov), dptr(g_scales), dptr(g_rots),
                stream_handle()), "d3ga_raster_backward_l1")
        else:
            check(L.d3ga_raster_backward(
                ctypes.byref(prm), dptr(means3D), dptr(sh), dptr(scales), dptr(rotations), dptr(cov3Ds_precomp),
                dptr(view), dptr(proj), dptr(campos), dptr(bg), dptr(geom), dptr(binning), ctx.cap, dptr(img),
                dptr(grad_color), dptr(acc), dptr(g_means3D), dptr(g_means2D), dptr(g_opac), dptr(g_sh), dptr(g_col),
                dptr(g_cov), dptr(g_scales), dptr(g_rots), stream_handle()), "d3ga_raster_backward")
        if sync is not None:
            if factor is not None:
                factor[P].copy_(campos.reshape(-1)[:3])
            if getattr(sync, "deferred", False):
                # two-graph step (d3ga_amd.graph.CapturedCutStep): the collectives run between the graphs, on these buffers
                parts_by_name = {"means3D": g_means3D, "opacities": g_opac}
                if from_sr:
                    parts_by_name.update(scales=g_scales, rotations=g_rots)
                else:
                    parts_by_name["cov3D_precomp"] = g_cov
                if sh is None:
                    parts_by_name["colors_precomp"] = g_col
                sync.park(flat, factor, parts_by_name,
                          None if factor is None else {"P": P, "M": prm.M, "sh_degree": prm.sh_degree, "means3D": means3D})
                # the parked gradients come back REDUCED through graph.CapturedCutStep: returning them here as well would
                # leave the unreduced copy on the (detached) leaves, where the step adds the other loss terms' gradients
                return (None, g_means2D if ctx.has_means2D else None, None, None, None, None, None, None, None, None, None,
                        None, None, None, None)
            gathered = sync.exchange(flat, factor)         # flat: summed (averaged) in place; gathered: (world, P+1, 3)
            if factor is not None:
                g_sh = new(P, prm.M, 3)
                check(L.d3ga_sh_grad_from_views(P, prm.M, prm.sh_degree, sync.world, dptr(means3D), dptr(gathered),
                                                3 * (P + 1), dptr(gathered[0, P]), 3 * (P + 1), sync.scale, dptr(g_sh),
                                                stream_handle()), "d3ga_sh_grad_from_views")
                g_col = None
        return (g_means3D, g_means2D if ctx.has_means2D else None, g_sh, g_col, g_opac, g_scales, g_rots, g_cov, None,
                None, None, None, None, None, None)


_ACTIVATIONS = {None: 0, "none": 0, "sigmoid": 1}       # D3GA_OPACITY_SIGMOID


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, grad_sync=None, opacity_activation=None, want_invdepth=True):
    """opacity_activation="sigmoid" (extension over upstream): `opacities` holds the raw logits and the sigmoid of
    models/cage_net.py:247 runs inside the per-Gaussian kernels, forward and backward."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, grad_sync, None, None, opacity_activation, None, want_invdepth)


def rasterize_gaussians_l1(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                           raster_settings, target, grad_sync=None, opacity_activation=None, want_invdepth=True):
    """The render AND its L1 image loss (utils/loss_utils.py:29: mean |image - target|) from one operator (extension):
    returns (color, radii, invdepth, loss).  The gradient of `loss` is formed per pixel inside the compositing backward,
    so no (3,H,W) gradient image is written or read (one full-image kernel and 50 MB of traffic less per frame at 1080p);
    `color` stays differentiable as usual and both gradients add.  `target`: (3,H,W) tensor or a `graph.TensorSlot`."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, grad_sync, None, None, opacity_activation, target, want_invdepth)


def rasterize_gaussians_pair(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                             raster_settings, colors2, bg2, grad_sync=None, opacity_activation=None, want_invdepth=True):
    """Two images from ONE pass: the usual one and `colors2` (P,3; constants, no gradient) blended with the same alphas over
    `bg2`.  Returns (color, radii, invdepth, color2).  Gradients of both images reach the geometry and the opacities."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, grad_sync, colors2, bg2, opacity_activation, None, want_invdepth)


class GaussianRasterizer(nn.Module):
    """`grad_sync` (optional, not in upstream): a d3ga_amd.dist.ViewShardedGrads -- the gradients returned by the
    backward are then already summed/averaged over the ranks of the group (each rank renders its own camera)."""

    def __init__(self, raster_settings, grad_sync=None, opacity_activation=None):
        super().__init__()
        self.raster_settings = raster_settings
        self.grad_sync = grad_sync
        self.opacity_activation = opacity_activation      # "sigmoid": `opacities` are logits (fused D8 activation)
        self.want_invdepth = True                         # False: [2] of the result is None (the forward skips the depth image)

    def markVisible(self, positions):
        """Boolean (P,) mask: view-space z > 0.2 (upstream _C.mark_visible)."""
        require_cuda(positions)
        with torch.no_grad():
            p = _f32(positions, positions.device)
            view = _f32(self.raster_settings.viewmatrix, positions.device)
            vis = torch.empty((p.shape[0],), dtype=torch.uint8, device=p.device)
            check(_lib.lib().d3ga_raster_mark_visible(p.shape[0], dptr(p), dptr(view), dptr(vis), stream_handle()),
                  "d3ga_raster_mark_visible")
        return vis.bool()

    def forward(self, means3D, means2D, opacities, shs: Optional[torch.Tensor] = None,
                colors_precomp: Optional[torch.Tensor] = None, scales: Optional[torch.Tensor] = None,
                rotations: Optional[torch.Tensor] = None, cov3D_precomp: Optional[torch.Tensor] = None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings, self.grad_sync, self.opacity_activation, self.want_invdepth)


def last_tile_lists(W, H, device=None):
    """tile_lists() of the most recent forward on `device`."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    binning, cap = _last[dev]
    return tile_lists(binning, W, H, cap)


def last_termination(device=None):
    """(final_T (H,W) float32, n_contrib (H,W) int32) of the most recent forward on `device`: the per-pixel termination the
    backward reads (views into the forward's image scratch, layout from d3ga_raster_img_layout; inspection / tests -- the
    shared-decision parity test hands them to the oracle's backward)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    img, W, H = _last_img[dev][:3]
    off = (ctypes.c_int64 * 2)()
    check(_lib.lib().d3ga_raster_img_layout(W, H, off), "d3ga_raster_img_layout")
    n = 4 * W * H
    return (img[off[0]:off[0] + n].view(torch.float32).view(H, W), img[off[1]:off[1] + n].view(torch.int32).view(H, W))


def last_block_lists(device=None):
    """(blk_count (tiles,16) int64, blk_list (16 x capacity, 2) int64 {1-based tile-list position, Gaussian index}) of the most
    recent forward that was followed by (or may be followed by) a backward on `device` -- views decoded from its image scratch
    (d3ga_raster_img_layout_blocks; inspection / tests).  Block b of a tile whose list is [begin, end) starts at row
    16 begin + b (end - begin) of blk_list; blk_count is the prefix the backward walks."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    img, W, H = _last_img[dev][:3]
    off = (ctypes.c_int64 * 2)()
    check(_lib.lib().d3ga_raster_img_layout_blocks(W, H, off), "d3ga_raster_img_layout_blocks")
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    if img.numel() <= off[1]:
        raise RuntimeError("the last forward was a forward_only render: it has no block lists")
    cnt = img[off[0]:off[0] + 64 * tiles].view(torch.int32).view(tiles, 16).long()
    n = (img.numel() - off[1]) // 8
    lst = img[off[1]:off[1] + 8 * n].view(torch.int32).view(n, 2).long()
    return cnt, lst


def last_alpha_decisions(gid, px, py, device=None):
    """(ok (n,) bool, alpha (n,) float32): the compositing forward's own alpha and "touches the pixel" decision for the listed
    (Gaussian, pixel) pairs over the geometry records of the most recent forward (d3ga_selftest_alpha; tests)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    geom, P = _last_img[dev][3:5]
    d = geom.device
    gid, px, py = (torch.as_tensor(t, dtype=torch.int32).to(d).contiguous() for t in (gid, px, py))
    n = gid.numel()
    ok = torch.zeros(max(n, 1), dtype=torch.uint8, device=d)
    alpha = torch.zeros(max(n, 1), dtype=torch.float32, device=d)
    check(_lib.lib().d3ga_selftest_alpha(P, dptr(geom), n, dptr(gid), dptr(px), dptr(py), dptr(ok), dptr(alpha), stream_handle()),
          "d3ga_selftest_alpha")
    return ok[:n].bool(), alpha[:n]


def tile_lists(binning, W, H, d_capacity):
    """(tile_start (tiles+1,) int64, point_list (D,) int64, keys (D,) int64) views decoded from a binning buffer
    (inspection / tests; layout from d3ga_raster_binning_layout)."""
    off = (ctypes.c_int64 * 6)()
    check(_lib.lib().d3ga_raster_binning_layout(W, H, d_capacity, off), "d3ga_raster_binning_layout")
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    start = binning[off[2]:off[2] + 4 * (tiles + 1)].view(torch.int32).long() & 0xFFFFFFFF
    D = min(int(start[-1]), d_capacity)
    keys = binning[off[4]:off[4] + 8 * D].view(torch.int64)
    plist = binning[off[5]:off[5] + 4 * D].view(torch.int32).long()
    return start, plist, keys

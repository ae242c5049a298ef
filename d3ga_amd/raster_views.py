"""View-batched rasterization: k cameras of ONE set of Gaussians in one grid per stage (include/d3ga.h: d3ga_raster_params::n_views).

No counterpart upstream -- the reference renders one camera per call (renderer.py:69) and averages the losses of a batch of
frames (train.py:218-221).  A single avatar view leaves two thirds of the chip idle in the binning and compositing launches
(DESIGN.md sec. 4); k views of the same pose are one tall frame for those stages.  Per view the arithmetic is that of the
single-view operator, so every image equals `rasterize_gaussians` from that camera and the gradients equal the sum over the k
single-view backwards.

    CameraBatch(k, width, height)            k cameras in one static device buffer (graph-replayable: `set(batches)`)
    rasterize_gaussians_views(...)           -> (colors (k,3,H,W), radii (k,P), loss | None)
"""
import ctypes

import torch

from . import _lib
from ._lib import RasterParams, check, dptr, require_cuda, stream_handle
from .cameras import Camera
from . import rasterizer as _R


class CameraBatch:
    """k cameras of one raster size in ONE device buffer: view matrices (k,16) | full projections (k,16) | centres + tangents
    (k,5).  The kernels read tan(FoV/2) per view from the buffer (the camera-slot convention of include/d3ga.h: the settings
    carry tanfovx = 0), so the cameras of a batch may differ in everything but the raster size, and a captured step is replayed
    with other cameras after `set()` -- one asynchronous H2D copy."""

    def __init__(self, n_views, width, height, device="cuda"):
        self.n_views, self.image_width, self.image_height = int(n_views), int(width), int(height)
        k = self.n_views
        self.buffer = torch.zeros(37 * k, dtype=torch.float32, device=torch.device(device))
        self.viewmatrices = self.buffer[:16 * k].view(k, 16)
        self.projmatrices = self.buffer[16 * k:32 * k].view(k, 16)
        self.campos = self.buffer[32 * k:].view(k, 5)
        self._host = torch.zeros(37 * k, dtype=torch.float32)
        if self.buffer.is_cuda:
            self._host = self._host.pin_memory()
        self._host_np = self._host.numpy()
        self._event = None

    def set(self, batches):
        """batches: k dicts with the reference's camera keys (R, T, FoVx, FoVy, width, height: lib/cameras.py:14-26)."""
        k = self.n_views
        if len(batches) != k:
            raise ValueError(f"CameraBatch holds {k} cameras, got {len(batches)}")
        if self._event is not None:
            self._event.synchronize()                 # the copy that last read the staging buffer has run
        h = self._host_np
        for v, b in enumerate(batches):
            if int(b["width"]) != self.image_width or int(b["height"]) != self.image_height:
                raise ValueError(f"CameraBatch is {self.image_width}x{self.image_height}; view {v} is {b['width']}x{b['height']}")
            m = Camera.pack_host_cached(b)        # view (16) | projection (16) | full (16) | centre (3) | tan(FoVx/2), tan(FoVy/2)
            h[16 * v:16 * v + 16] = m[0:16]
            h[16 * k + 16 * v:16 * k + 16 * v + 16] = m[32:48]
            o = 32 * k + 5 * v
            h[o:o + 3] = m[48:51]
            h[o + 3:o + 5] = m[51:53]
        self.buffer.copy_(self._host, non_blocking=True)
        if self.buffer.is_cuda:
            self._event = torch.cuda.Event()
            self._event.record()
        return self


def _scratch_views(P, W, H, k, cap, dev, fwd_only):
    sizes = (ctypes.c_int64 * 3)()
    check(_lib.lib().d3ga_raster_scratch_bytes_views(P, W, H, k, cap, int(fwd_only), sizes), "d3ga_raster_scratch_bytes_views")
    return [torch.empty(int(n), dtype=torch.uint8, device=dev) for n in sizes]


class _RasterizeViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, cams, bg, sh_degree,
                scale_modifier, antialiasing, opacity_activation, l1_targets, colors2=None, bg2=None, grad_sync=None):
        require_cuda(means3D)
        dev = means3D.device
        f32 = lambda t: _R._f32(t, dev)
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, bg = map(
            f32, (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, bg))
        if (sh is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3Ds_precomp is None) or (
                (scales is not None or rotations is not None) and cov3Ds_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        k, W, H = cams.n_views, cams.image_width, cams.image_height
        # a batch of FRAMES: every view brings its own geometry (k,P,.) -- the avatar deformed per pose -- and shares the appearance
        per_view = means3D.dim() == 3
        P = means3D.shape[-2]
        geo = [t for t in (means3D, scales, rotations, cov3Ds_precomp) if t is not None]
        if any((t.dim() == 3) != per_view for t in geo) or (per_view and any(t.shape[0] != k for t in geo)):
            raise ValueError("rasterize_gaussians_views: means3D and the covariance inputs must all be (P,.) or all be (k,P,.)")
        if opacities.shape[0] != P:
            raise ValueError("rasterize_gaussians_views: opacities (and the colours) are shared by the views: (P,.)")
        M = sh.shape[1] if sh is not None else 0
        fwd_only = not any(ctx.needs_input_grad[:7])
        dual = colors2 is not None
        if dual and l1_targets is not None:
            raise ValueError("rasterize_gaussians_views: the fused L1 loss and a second colour set cannot be combined")
        if dual:
            colors2, bg2 = f32(colors2.detach()), f32(bg2)
        prm = RasterParams(P=P, M=M, sh_degree=int(sh_degree), W=W, H=H, tanfovx=0.0, tanfovy=0.0,
                           scale_modifier=float(scale_modifier), antialiasing=int(bool(antialiasing)), prefiltered=0, debug=0,
                           opacity_activation=_R._ACTIVATIONS[opacity_activation], forward_only=int(fwd_only), n_views=k,
                           per_view_geometry=int(per_view))
        colors2_img = torch.empty((k, 3, H, W), dtype=torch.float32, device=dev) if dual else None
        colors = torch.empty((k, 3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((k, P), dtype=torch.int32, device=dev)
        loss = tgt = None
        if l1_targets is not None:
            tgt = f32(l1_targets)
            if tuple(tgt.shape) != (k, 3, H, W):
                raise ValueError(f"rasterize_gaussians_views: the targets must be ({k}, 3, {H}, {W}), got {tuple(tgt.shape)}")
            loss = torch.empty((), dtype=torch.float32, device=dev)
        L = _lib.lib()
        st, pp = stream_handle(), ctypes.byref(prm)
        static = _R._policy["mode"] == "static"
        cap = _R._policy["static"] if static else max(k * _R._hwm.get(dev.index, 0), k * (4 * P + 1024))
        while True:
            geom, binning, img = _scratch_views(P, W, H, k, cap, dev, fwd_only)
            tm = _R.stage_timer
            tm.stage("preprocess", lambda: check(L.d3ga_raster_preprocess(
                pp, dptr(means3D), dptr(sh), dptr(colors_precomp), dptr(opacities), dptr(scales), dptr(rotations),
                dptr(cov3Ds_precomp), dptr(cams.viewmatrices), dptr(cams.projmatrices), dptr(cams.campos), dptr(geom), dptr(binning),
                cap, dptr(radii), st), "d3ga_raster_preprocess"))
            tm.stage("bin_sort", lambda: check(L.d3ga_raster_bin_sort(pp, dptr(geom), dptr(binning), cap, st), "d3ga_raster_bin_sort"))
            if tgt is not None and P > 0:
                ws = torch.empty(4 * k * ((W + 15) // 16) * ((H + 15) // 16), dtype=torch.float32, device=dev)
                tm.stage("composite_fwd", lambda: check(L.d3ga_raster_composite_fwd_l1(
                    pp, dptr(bg), dptr(geom), dptr(binning), cap, dptr(img), dptr(colors), None, dptr(tgt), None, dptr(loss), dptr(ws), st),
                    "d3ga_raster_composite_fwd_l1"))
            elif dual and P > 0:
                tm.stage("composite_fwd", lambda: check(L.d3ga_raster_composite_fwd2(
                    pp, dptr(bg), dptr(bg2), dptr(geom), dptr(colors2), dptr(binning), cap, dptr(img), dptr(colors), dptr(colors2_img),
                    None, st), "d3ga_raster_composite_fwd2"))
            else:
                tm.stage("composite_fwd", lambda: check(L.d3ga_raster_composite_fwd(
                    pp, dptr(bg), dptr(geom), dptr(binning), cap, dptr(img), dptr(colors), None, st), "d3ga_raster_composite_fwd"))
                if tgt is not None:
                    _R.l1_mean_forward(colors, tgt, None, loss, dev)
                if dual:
                    colors2_img.copy_(bg2.view(1, 3, 1, 1).expand_as(colors2_img))
            _R._last[dev.index] = (binning, cap)
            if _R._capture_log is not None:
                _R._capture_log.append((binning, cap))
            if static:
                break
            cnt = binning[:32].view(torch.int32)[:2].cpu().tolist()           # host sync (upstream: num_rendered)
            D = cnt[0] & 0xFFFFFFFF
            _R._hwm[dev.index] = max(_R._hwm.get(dev.index, 0), int(D * 1.25 / k) + 1024)      # the mark is per view
            if not cnt[1]:
                break
            cap = k * _R._hwm[dev.index]
        ctx.prm, ctx.cap, ctx.cams = prm, cap, cams
        ctx.l1 = tgt is not None and P > 0
        ctx.dual, ctx.per_view = dual, per_view
        # camera-sharded training (d3ga_amd/dist.py: ViewShardedGrads): every gradient that leaves this op is summed over the ranks
        # at this cut -- this rank's k views arrive already summed, their k SH factors travel in one all-gather
        ctx.grad_sync = grad_sync if (grad_sync is not None and (grad_sync.world > 1 or getattr(grad_sync, "always", False)) and P > 0) else None
        if ctx.grad_sync is not None:
            if per_view:
                raise ValueError("rasterize_gaussians_views: grad_sync needs view-independent geometry (k cameras of one pose); a batch of "
                                 "frames is reduced at the parameters (dist.GradReducer)")
            if hasattr(grad_sync, "verify_inputs"):
                grad_sync.verify_inputs({"means3D": means3D, "opacities": opacities, "colors_precomp": colors_precomp, "shs": sh,
                                         "cov3D_precomp": cov3Ds_precomp, "scales": scales, "rotations": rotations})
        ctx.save_for_backward(means3D, sh, scales, rotations, cov3Ds_precomp, bg, geom, binning, img,
                              colors if ctx.l1 else None, tgt if ctx.l1 else None, colors2, bg2)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        if dual:
            return colors, radii, colors2_img
        return (colors, radii, loss) if tgt is not None else (colors, radii)

    @staticmethod
    def backward(ctx, grad_colors, _grad_radii, grad_third=None):
        means3D, sh, scales, rotations, cov3Ds_precomp, bg, geom, binning, img, image, tgt, colors2, bg2 = ctx.saved_tensors
        prm, cams, dev, P, k = ctx.prm, ctx.cams, means3D.device, ctx.prm.P, ctx.prm.n_views
        if P == 0:
            return (None,) * 17
        g_loss = grad_colors2 = None
        if ctx.l1 and grad_third is not None:
            g_loss = _R._f32(grad_third, dev).reshape(1)
        if ctx.dual:
            grad_colors2 = torch.zeros((k, 3, prm.H, prm.W), dtype=torch.float32, device=dev) if grad_third is None else _R._f32(grad_third, dev)
        if grad_colors is None and g_loss is None:
            grad_colors = torch.zeros((k, 3, prm.H, prm.W), dtype=torch.float32, device=dev)
        grad_colors = _R._f32(grad_colors, dev)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        # (k P, 16) screen-space accumulator: under rasterizer.set_accumulator_policy("persistent") ONE zeroed buffer per size is kept
        # and the per-Gaussian backward leaves it all zero again -- no 64 B x k P fill per backward
        acc, self_clearing = _R._accumulator(k * P, dev)
        from_sr = cov3Ds_precomp is None
        sync, flat, factor = ctx.grad_sync, None, None
        if sync is None:
            gshape = (k, P) if ctx.per_view else (P,)                 # a batch of frames: geometry gradients per view
            g_means3D, g_opac = new(*gshape, 3), new(P, 1)
            g_sh = new(P, prm.M, 3) if sh is not None else None
            g_col = new(k, P, 3) if sh is not None else new(P, 3)      # SH: the per-view factors of the rank-1 SH gradient (scratch)
            g_cov = None if from_sr else new(*gshape, 6)
            g_scales = new(*gshape, 3) if from_sr else None
            g_rots = new(*gshape, 4) if from_sr else None
        else:
            # one planar buffer for the all-reduce (as rasterizer._RasterizeGaussians.backward lays it out); the SH gradient leaves
            # as k factors of (P + 1, 3) -- row P carries the view's camera position -- for ONE all-gather of (k, P + 1, 3) per rank
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
                g_col = parts[-1]
            else:
                factor = new(k, P + 1, 3)
                g_col = factor
                p2 = RasterParams(**{f: getattr(prm, f) for f, _ in RasterParams._fields_})
                p2.factor_rows = P + 1
                prm = p2
        if self_clearing != bool(prm.acc_self_clearing):
            p3 = RasterParams(**{f: getattr(prm, f) for f, _ in RasterParams._fields_})      # (ctx.prm may be shared with a retained graph)
            p3.acc_self_clearing = int(self_clearing)
            prm = p3
        L = _lib.lib()
        st, pp = stream_handle(), ctypes.byref(prm)          # (prm: the block with factor_rows when the gradients are exchanged)
        tm = _R.stage_timer
        if not self_clearing:
            acc.zero_()
        if ctx.dual:
            tm.stage("composite_bwd", lambda: check(L.d3ga_raster_composite_bwd2(
                pp, dptr(bg), dptr(bg2), dptr(geom), dptr(colors2), dptr(binning), ctx.cap, dptr(img), dptr(grad_colors),
                dptr(grad_colors2), dptr(acc), st), "d3ga_raster_composite_bwd2"))
        elif g_loss is not None:
            tm.stage("composite_bwd", lambda: check(L.d3ga_raster_composite_bwd_l1(
                pp, dptr(bg), dptr(geom), dptr(binning), ctx.cap, dptr(img), dptr(image), dptr(tgt), None, dptr(g_loss),
                dptr(grad_colors), dptr(acc), st), "d3ga_raster_composite_bwd_l1"))
        else:
            tm.stage("composite_bwd", lambda: check(L.d3ga_raster_composite_bwd(
                pp, dptr(bg), dptr(geom), dptr(binning), ctx.cap, dptr(img), dptr(grad_colors), dptr(acc), st),
                "d3ga_raster_composite_bwd"))
        tm.stage("preprocess_bwd", lambda: check(L.d3ga_raster_preprocess_bwd(
            pp, dptr(means3D), dptr(sh), dptr(scales), dptr(rotations), dptr(cov3Ds_precomp), dptr(cams.viewmatrices),
            dptr(cams.projmatrices), dptr(cams.campos), dptr(geom), dptr(acc), dptr(g_means3D), None, dptr(g_opac),
            dptr(g_sh if sync is None else None), dptr(g_col), dptr(g_cov), dptr(g_scales), dptr(g_rots), st), "d3ga_raster_preprocess_bwd"))
        if sync is not None:
            if factor is not None:
                factor[:, P].copy_(cams.campos[:, :3])
            if getattr(sync, "deferred", False):          # two-graph step (graph.CapturedCutStep): the collectives run between the graphs
                by_name = {"means3D": g_means3D, "opacities": g_opac}
                by_name.update({"scales": g_scales, "rotations": g_rots} if from_sr else {"cov3D_precomp": g_cov})
                if sh is None:
                    by_name["colors_precomp"] = g_col
                sync.park(flat, factor, by_name, None if factor is None else {"P": P, "M": prm.M, "sh_degree": prm.sh_degree, "means3D": means3D})
                return (None,) * 17
            gathered = sync.exchange(flat, factor)        # flat: averaged over the ranks in place; gathered: (world, k, P + 1, 3)
            if factor is not None:
                g_sh = new(P, prm.M, 3)
                g = gathered.view(-1, P + 1, 3)
                check(L.d3ga_sh_grad_from_views(P, prm.M, prm.sh_degree, g.shape[0], dptr(means3D), dptr(g), 3 * (P + 1), dptr(g[0, P]),
                                                3 * (P + 1), sync.scale, dptr(g_sh), stream_handle()), "d3ga_sh_grad_from_views")
                g_col = None
        return (g_means3D, g_sh, g_col if sh is None else None, g_opac, g_scales, g_rots, g_cov,
                None, None, None, None, None, None, None, None, None, None)


def rasterize_gaussians_views(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, cameras, bg, sh_degree=0,
                              scale_modifier=1.0, antialiasing=False, opacity_activation=None, l1_targets=None, colors2=None, bg2=None,
                              grad_sync=None):
    """k views in one grid per stage.  cameras: a CameraBatch; bg (3,) shared by the views.
    Geometry: means3D (P,3) with cov3Ds_precomp (P,6) | scales + rotations -- k CAMERAS of one set of Gaussians -- or all of them
    (k,P,.) -- k FRAMES, the avatar deformed per pose (the reference's batch, train.py:218-221); opacities and the colours are shared.
    -> (colors (k,3,H,W), radii (k,P)); with l1_targets (k,3,H,W): (colors, radii, loss), loss = mean |colors - targets| over all k
    images (= the mean over the frames of the reference's per-frame l1_loss: equal sizes), its gradient formed inside the compositing
    backward; with colors2 (P,3) + bg2: (colors, radii, colors2_image (k,3,H,W)) -- the reference's RGB + silhouette pair
    (models/trainer.py:102-110) from one pass, colors2 constant.  Gradients as `rasterize_gaussians`: summed over the views for
    shared inputs, per view for (k,P,.) geometry.  grad_sync: a dist.ViewShardedGrads -- the returned gradients are then already
    averaged over the ranks of a camera-sharded run (every rank renders its own k cameras of the pose)."""
    return _RasterizeViews.apply(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, cameras, bg,
                                 sh_degree, scale_modifier, antialiasing, opacity_activation, l1_targets, colors2, bg2, grad_sync)

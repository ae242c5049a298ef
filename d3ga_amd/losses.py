"""Loss tail next to the render boundary: fused L1 and SSIM image losses.

Drop-ins for the reference's `utils/loss_utils.py`:
    l1_loss(network_output, gt)                       :29   (train.py:190-191)
    ssim(img1, img2, window_size=11, size_average=True) :59-86 (train.py:192)
each one HIP kernel forward and one backward (the reference runs 6 resp. ~25 full-image ATen/MIOpen kernels per
direction).  GPU tensors only.
"""
import torch

from . import _lib
from ._lib import check, dptr, f32c16, require_cuda, stream_handle


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, cell):
        # cell: None, or the (1,) int64 device tensor of a graph.TensorSlot naming `b` (b is then only its current tensor)
        require_cuda(a, b)
        a = f32c16(a)
        if cell is None:
            b = f32c16(b)
        if a.shape != b.shape:
            raise ValueError(f"l1_loss: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
        # [0]: the loss; [4:]: one partial sum per workgroup (two-stage reduction: no zero fill, no atomics, reproducible)
        buf = torch.empty(4 + _lib.LOSS_PARTIALS, dtype=torch.float32, device=a.device)
        out = buf[0]
        if cell is None:
            check(_lib.lib().d3ga_l1_mean_fwd_ws(a.numel(), dptr(a), dptr(b), dptr(out), dptr(buf[4:]), stream_handle()),
                  "d3ga_l1_mean_fwd_ws")
            ctx.save_for_backward(a, b)
        else:
            check(_lib.lib().d3ga_l1_mean_fwd_ws_cell(a.numel(), dptr(a), dptr(cell), dptr(out), dptr(buf[4:]),
                                                      stream_handle()), "d3ga_l1_mean_fwd_ws_cell")
            ctx.save_for_backward(a, cell)
        ctx.by_cell = cell is not None
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        if ctx.by_cell:
            check(_lib.lib().d3ga_l1_mean_bwd_cell(a.numel(), dptr(a), dptr(b), dptr(f32c16(g)), dptr(ga), stream_handle()),
                  "d3ga_l1_mean_bwd_cell")
            return ga, None, None
        check(_lib.lib().d3ga_l1_mean_bwd(a.numel(), dptr(a), dptr(b), dptr(f32c16(g)), dptr(ga),
                                          stream_handle()), "d3ga_l1_mean_bwd")
        return ga, (-ga if ctx.needs_input_grad[1] else None), None


def l1_loss(network_output, gt):
    """mean |network_output - gt| (utils/loss_utils.py:29).  GPU tensors only.  `gt` may be a `graph.TensorSlot`: the kernels
    then read the target's address from the slot's device cell (a captured step follows `slot.set(image)` without a copy)."""
    from .graph import TensorSlot
    if isinstance(gt, TensorSlot):
        return _L1Mean.apply(network_output, gt.current, gt.cell)
    return _L1Mean.apply(network_output, gt, None)


class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        require_cuda(img1, img2)
        a = f32c16(img1)
        b = f32c16(img2)
        if a.shape != b.shape or a.dim() != 3:
            raise ValueError(f"ssim: expected two (C,H,W) images of equal shape, got {tuple(a.shape)} / {tuple(b.shape)}")
        C, H, W = a.shape
        out = torch.empty((), dtype=torch.float32, device=a.device)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        d = [torch.empty_like(a) for _ in range(3)] if ctx.needs_input_grad[0] else [None, None, None]
        check(_lib.lib().d3ga_ssim_fwd(C, H, W, dptr(a), dptr(b), dptr(out), dptr(d[0]), dptr(d[1]), dptr(d[2]),
                                       stream_handle()), "d3ga_ssim_fwd")
        if need:
            ctx.save_for_backward(a, b, *[t for t in d if t is not None])
        return out

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        a, b = saved[0], saved[1]
        C, H, W = a.shape
        g = f32c16(g)
        L = _lib.lib()
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = torch.empty_like(a)
            check(L.d3ga_ssim_bwd(C, H, W, dptr(a), dptr(b), dptr(saved[2]), dptr(saved[3]), dptr(saved[4]), dptr(g),
                                  dptr(ga), stream_handle()), "d3ga_ssim_bwd")
        if ctx.needs_input_grad[1]:                    # SSIM is symmetric: the same kernels with the images swapped
            d = [torch.empty_like(a) for _ in range(3)]
            tmp = torch.empty((), dtype=torch.float32, device=a.device)
            check(L.d3ga_ssim_fwd(C, H, W, dptr(b), dptr(a), dptr(tmp), dptr(d[0]), dptr(d[1]), dptr(d[2]),
                                  stream_handle()), "d3ga_ssim_fwd")
            gb = torch.empty_like(a)
            check(L.d3ga_ssim_bwd(C, H, W, dptr(b), dptr(a), dptr(d[0]), dptr(d[1]), dptr(d[2]), dptr(g), dptr(gb),
                                  stream_handle()), "d3ga_ssim_bwd")
        return ga, gb


def ssim(img1, img2, window_size=11, size_average=True):
    """Mean SSIM with the reference's 11x11 Gaussian window (sigma 1.5, zero padding), utils/loss_utils.py:59-86.
    (C,H,W) images, or (N,C,H,W) batches (mean over the batch, or one value per image with size_average=False)."""
    if window_size != 11:
        raise NotImplementedError("ssim: only the reference's window_size=11 is implemented (train.py:192 uses the default)")
    if img1.dim() == 3:
        v = _SSIM.apply(img1, img2)
        return v if size_average else v[None]
    if img1.dim() != 4 or img1.shape != img2.shape:
        raise ValueError(f"ssim: expected (C,H,W) or (N,C,H,W) inputs of equal shape, got {tuple(img1.shape)}")
    per = torch.stack([_SSIM.apply(img1[n], img2[n]) for n in range(img1.shape[0])])
    return per.mean() if size_average else per


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        require_cuda(img1, img2)
        a = f32c16(img1)
        b = f32c16(img2)
        if a.shape != b.shape or a.dim() != 3:
            raise ValueError(f"l1_ssim: expected two (C,H,W) images of equal shape, got {tuple(a.shape)} / {tuple(b.shape)}")
        C, H, W = a.shape
        out = torch.empty((2,), dtype=torch.float32, device=a.device)
        d = [torch.empty_like(a) for _ in range(3)]
        check(_lib.lib().d3ga_ssim_l1_fwd(C, H, W, dptr(a), dptr(b), dptr(out[1:]), dptr(d[0]), dptr(d[1]), dptr(d[2]),
                                          dptr(out[:1]), stream_handle()), "d3ga_ssim_l1_fwd")
        ctx.save_for_backward(a, b, *d)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        a, b, d0, d1, d2 = ctx.saved_tensors
        C, H, W = a.shape
        g = torch.stack([g_l1.float().reshape(()), g_ssim.float().reshape(())])
        ga = torch.empty_like(a)
        check(_lib.lib().d3ga_ssim_l1_bwd(C, H, W, dptr(a), dptr(b), dptr(d0), dptr(d1), dptr(d2), dptr(g[1:]),
                                          dptr(g[:1]), dptr(ga), stream_handle()), "d3ga_ssim_l1_bwd")
        return ga, None


def l1_ssim(network_output, gt):
    """(l1_loss(network_output, gt), ssim(network_output, gt)) from ONE kernel each way -- the pair train.py:190-193
    combines as (1 - lambda) * l1 + lambda * (1 - ssim).  Differentiable in network_output only (gt is the target)."""
    return _L1SSIM.apply(network_output, gt)

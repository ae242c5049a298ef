"""Loss tail next to the render boundary: fused L1 image loss.

Drop-in for `utils/loss_utils.py:29  l1_loss(network_output, gt) = torch.abs(network_output - gt).mean()` of the
reference (used at train.py:190): one streaming HIP kernel forward, one backward, instead of six ATen kernels.
"""
import torch

from . import _lib
from ._lib import check, dptr, require_cuda, stream_handle


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        require_cuda(a, b)
        a = a.float().contiguous()
        b = b.float().contiguous()
        if a.shape != b.shape:
            raise ValueError(f"l1_loss: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty((), dtype=torch.float32, device=a.device)
        check(_lib.lib().d3ga_l1_mean_fwd(a.numel(), dptr(a), dptr(b), dptr(out), stream_handle()), "d3ga_l1_mean_fwd")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        check(_lib.lib().d3ga_l1_mean_bwd(a.numel(), dptr(a), dptr(b), dptr(g.float().contiguous()), dptr(ga),
                                          stream_handle()), "d3ga_l1_mean_bwd")
        return ga, (-ga if ctx.needs_input_grad[1] else None)


def l1_loss(network_output, gt):
    """mean |network_output - gt| (utils/loss_utils.py:29).  GPU tensors only."""
    return _L1Mean.apply(network_output, gt)

"""Camera-sharded data parallelism: one process per GPU, one view per rank, one gradient sum per step.

The reference has no distributed code; D3GA trains on one random camera per step
(datasets/actorshq_dataset.py:229).  Multi-view training here shards the cameras of a frame across the ranks of one
node: every rank holds a full replica of the avatar parameters, renders its own view(s), and the parameter
gradients are summed with ONE all-reduce over RCCL/xGMI (backend "nccl" on ROCm) and divided by the world size
(the reference averages the loss over the frames of a batch, train.py:218-221).  The deform is view-independent
and recomputed per rank (60 MB of HBM traffic at 500k Gaussians -- cheaper than broadcasting its outputs).

Two reducers are provided.  `GradReducer` (used by bench.py) reduces the gradient tensors in place, one large
asynchronous all-reduce per parameter tensor, with no zero-fill / accumulate / packing traffic at all.  `FlatGrads`
keeps every `.grad` as a view into ONE flat buffer, so the reduction is a single collective -- on the point-to-point
xGMI mesh few large messages beat many small ones, and RCCL is free to use its direct algorithms across the 7 links.
"""
import os

import torch
import torch.distributed as dist


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 0, 1
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_views(n_views, rank, world):
    """Indices of the views rank `rank` renders (round-robin, so any n_views works)."""
    return list(range(rank, n_views, world))


class GradReducer:
    """Per-tensor gradient averaging without any staging copy.

    `zero()` drops the gradients (`p.grad = None`), so autograd hands each parameter the gradient tensor its producer
    wrote (no zero-fill, no accumulate kernel -- at 500k Gaussians that is ~240 MB/frame of avoidable HBM traffic);
    `all_reduce_mean()` launches one asynchronous all-reduce per parameter tensor (a handful of large messages: the
    SH gradient alone is 96 MB) and divides by the world size.  No-op for a single process."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]

    def zero(self):
        for p in self.params:
            p.grad = None

    def nbytes(self):
        return sum(p.numel() for p in self.params) * 4

    def all_reduce_mean(self):
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True) for g in grads]
        for w in works:
            w.wait()
        torch._foreach_div_(grads, float(dist.get_world_size()))


class FlatGrads:
    """Gives every parameter a `.grad` that is a view into one contiguous float32 buffer."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.buffer = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.buffer[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.buffer.zero_()

    def nbytes(self):
        return self.buffer.numel() * 4

    def all_reduce_mean(self, async_op=False):
        """Sum over ranks then divide by the world size.  No-op for a single process."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return None
        # autograd may have replaced p.grad by a fresh tensor (first backward after zero_grad(set_to_none));
        # re-home any stray gradient into the flat buffer before reducing
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is not None and p.grad.data_ptr() != self.buffer[off:off + n].data_ptr():
                self.buffer[off:off + n].copy_(p.grad.reshape(-1))
                p.grad = self.buffer[off:off + n].view_as(p)
            off += n
        work = dist.all_reduce(self.buffer, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            return work
        self.buffer.div_(dist.get_world_size())
        return None

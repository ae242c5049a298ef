"""Camera-sharded data parallelism: one process per GPU, one view per rank, one gradient sum per step.

The reference has no distributed code; D3GA trains on one random camera per step
(datasets/actorshq_dataset.py:229).  Multi-view training here shards the cameras of a frame across the ranks of one
node: every rank holds a full replica of the avatar parameters, renders its own view(s), and the parameter
gradients are summed with ONE all-reduce over RCCL/xGMI (backend "nccl" on ROCm) and divided by the world size
(the reference averages the loss over the frames of a batch, train.py:218-221).  The deform is view-independent
and recomputed per rank (60 MB of HBM traffic at 500k Gaussians -- cheaper than broadcasting its outputs).

Three reducers are provided.  `ViewShardedGrads` (the default of bench.py for N > 1) does not reduce PARAMETER
gradients at all: it sums the gradients where the graph is narrowest -- at the rasterizer's inputs, before they fan out
into the deform backward -- and exchanges the wide (P,M,3) SH gradient in its rank-1 factored form, 52 + 12*(N-1) bytes
per Gaussian on the wire instead of ~2*248.  `GradReducer` (used by bench.py) reduces the gradient tensors in place, one large
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


class ViewDependentInputError(RuntimeError):
    """Raised by ViewShardedGrads.verify_inputs: a tensor entering the rasterizer differs between the ranks."""


class ViewShardedGrads:
    """Gradient exchange of camera-sharded training at the narrowest cut of the graph.

    Pass it as `grad_sync` to `render` / `GaussianRasterizer`.  The rasterizer's backward then returns gradients that
    are already summed over the ranks (averaged when `average`, the reference averages the losses of a batch,
    train.py:218-221), so everything upstream of the rasterizer -- the cage deform, LBS, activations, any network that
    produced the per-Gaussian attributes -- back-propagates an identical, already reduced signal on every rank and
    NO parameter all-reduce is needed (terms of the loss that do not pass through the rasterizer, e.g. the FEM energy,
    are view-independent and therefore identical on every rank by construction).

    PRECONDITION -- everything upstream of the rasterizer's inputs must be VIEW-INDEPENDENT.  Summing dL/d(input) over
    the ranks and back-propagating the sum through rank r's own graph computes J_r^T sum_v g_v; that equals the wanted
    sum_v J_v^T g_v only when the Jacobian J of (parameters -> rasterizer inputs) is the same on every rank.  True for the
    SH path (means / covariance / opacity / SH coefficients are functions of the pose only; the view dependence of the
    colour lives INSIDE the rasterizer and is handled by the factored SH exchange).  NOT true when colours or opacities come
    from a network that sees the camera -- the reference's main configuration (configs/actorshq_actor02.yml, use_shs false:
    ColorField(view_dir, camera / frame encodings), models/cage_net.py:232-258) -- nor for per-camera parameters AFTER the
    rasterizer (learnable blur, pixel calibration): those configurations must use `GradReducer` / `FlatGrads` over ALL
    parameters instead.  Because the silent failure mode is replicas that drift apart,
    `verify_inputs` compares checksums of the rasterizer's inputs across the ranks on the first `verify_steps` calls
    (one small all-reduce + a host read-back each) and raises ViewDependentInputError when they differ.

    Per step and rank, for P Gaussians:
      * one all-reduce of a planar buffer [dL/dmeans3D | dL/dopacity | dL/dcov3D  (or dL/dscales | dL/drots)
        | dL/dcolors_precomp]: 40 B per Gaussian on the cov3D_precomp path;
      * SH path: one all-gather of the clamp-masked dL/dcolour (P,3) plus the camera position (12 B per Gaussian and
        view).  For one view the SH gradient is rank-1 per Gaussian, Y_k(dir) * dL/dcolour_c, so every rank rebuilds
        sum_v Y(dir_v) (x) g_v locally (d3ga_sh_grad_from_views) instead of moving 12*M B per Gaussian.
    At C3 (P = 500k, M = 16) on 8 ranks: 20 MB all-reduced + 42 MB gathered per rank, against 124 MB all-reduced by
    the parameter-level reducers below.

    `timing=True` brackets every exchange with an event pair on the launch stream (`exchange_ms()`; bench.py's
    exchange_ms / compute_ms split).
    """

    def __init__(self, group=None, average=True, verify_steps=1, timing=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.scale = 1.0 / self.world if average else 1.0
        self.bytes_last = 0           # payload bytes this rank contributed in the most recent exchange
        self.verify_left = int(verify_steps)
        self.timing = bool(timing)
        self.always = False           # tests: run the exchange even in a one-rank group (the collectives are then identities)
        self._events = []             # (start, end) event pairs of the exchanges since the last exchange_ms()
        # Deferred mode (d3ga_amd.graph.CapturedCutStep): the rasterizer's backward parks its outputs here instead of
        # exchanging them, so that the step can be captured as two hipGraphs with the collectives issued between them.
        self.deferred = False
        self.parked = []              # one entry per rasterizer backward of the step: dict(flat, factor, parts {input name ->
                                      # gradient view of flat}, sh (rebuild arguments), gathered)
        self.frozen = False           # set once a graph has captured the parked buffers: their shapes may not change any more

    def verify_inputs(self, named):
        """Collective.  `named`: dict name -> tensor (or None) entering the rasterizer on this rank.  Raises
        ViewDependentInputError if any of them differs between the ranks (see the class docstring).  No-op after
        `verify_steps` calls, for a single rank, and during stream capture."""
        if self.verify_left <= 0 or self.world == 1:
            return
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
        self.verify_left -= 1
        names = [k for k, t in named.items() if t is not None and t.numel() > 0]
        if not names:
            return
        t0 = named[names[0]]
        sums = torch.stack([torch.stack((named[k].detach().double().sum(), named[k].detach().double().abs().sum()))
                            for k in names]).reshape(-1)
        both = torch.cat([sums, -sums]).to(t0.device)
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self.group)       # max(x) and max(-x) = -min(x) in one collective
        n = sums.numel()
        hi, lo = both[:n].cpu(), (-both[n:]).cpu()
        scale = torch.maximum(hi.abs(), lo.abs()).clamp_min(1e-30)
        bad = ((hi - lo) / scale > 1e-9).reshape(-1, 2).any(dim=1)
        if bool(bad.any()):
            which = [k for k, b in zip(names, bad.tolist()) if b]
            raise ViewDependentInputError(
                f"ViewShardedGrads: rasterizer input(s) {which} differ between the ranks, i.e. they depend on the view "
                "(e.g. ColorField colours / opacities).  Summing gradients at the rasterizer's inputs is only valid for "
                "view-independent inputs; use d3ga_amd.dist.GradReducer or FlatGrads over all parameters instead.")

    def exchange(self, flat, factor=None, out=None):
        """Sums `flat` over the ranks in place (times `scale`); gathers `factor` (P+1,3) of every rank into a
        (world, P+1, 3) tensor (returned; `out` if given; None without `factor`).  Both collectives are in flight together."""
        ev = None
        if self.timing and flat.is_cuda and not torch.cuda.is_current_stream_capturing():     # (an event recorded into a graph cannot be timed)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        works = [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        gathered = None
        self.bytes_last = flat.numel() * 4
        if factor is not None:
            shape = (self.world,) + tuple(factor.shape)
            gathered = out if (out is not None and tuple(out.shape) == shape) else torch.empty(shape, dtype=factor.dtype, device=factor.device)
            try:
                works.append(dist.all_gather_into_tensor(gathered, factor, group=self.group, async_op=True))
            except (RuntimeError, NotImplementedError):      # backends without the flat variant
                works.append(dist.all_gather(list(gathered.unbind(0)), factor, group=self.group, async_op=True))
            self.bytes_last += factor.numel() * 4
        for w in works:
            w.wait()
        if self.scale != 1.0:
            flat.mul_(self.scale)
        if ev is not None:
            ev[1].record()
            self._events.append(ev)
        return gathered

    def begin_step(self):
        """Deferred mode: forget the previous step's parked buffers (eager steps; a captured step keeps the buffers it was
        captured with and never calls this again)."""
        if self.frozen:
            raise RuntimeError("ViewShardedGrads: the parked buffers are baked into a captured graph (CapturedCutStep); "
                               "use another ViewShardedGrads for eager steps")
        self.parked = []

    def park(self, flat, factor, parts, sh=None):
        """Deferred mode, called by EVERY rasterizer backward of the step (the reference's step renders twice,
        models/trainer.py:102-110): remember its buffers (static tensors when the backward is being captured).  `parts`:
        rasterizer input name -> its gradient, a view of `flat`."""
        self.parked.append({"flat": flat, "factor": factor, "parts": parts, "sh": sh, "gathered": None})

    def exchange_parked(self):
        """Deferred mode, eager, between the two graphs: ONE all-reduce per distinct buffer layout -- the parked buffers of a
        step's renders that share a layout (k views per rank, or the reference's two renders of one package when both use the
        same colour path) are first added locally into the first of them -- plus one all-gather per SH render (every view has
        its own directions).  While a graph is being captured its first half has not run, so the collectives of that one call
        move unspecified (never read) data -- an extra exchange every rank issues alike."""
        if not self.parked:
            raise RuntimeError("ViewShardedGrads.exchange_parked: no rasterizer backward has parked its gradients")
        ev = None
        if self.timing and self.parked[0]["flat"].is_cuda and not torch.cuda.is_current_stream_capturing():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        heads = {}
        for e in self.parked:
            key = (e["flat"].numel(), tuple(sorted((k, tuple(v.shape)) for k, v in e["parts"].items())))
            head = heads.setdefault(key, e)
            e["head"] = head
            if head is not e:
                head["flat"].add_(e["flat"])               # local sum: one collective for all renders of this layout
        works, self.bytes_last = [], 0
        for e in self.parked:
            if e["head"] is e:
                works.append(dist.all_reduce(e["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.bytes_last += e["flat"].numel() * 4
            f = e["factor"]
            if f is not None:
                want = (self.world,) + tuple(f.shape)
                if e["gathered"] is not None and tuple(e["gathered"].shape) != want:
                    if self.frozen:
                        raise RuntimeError(f"ViewShardedGrads: the gather buffer {tuple(e['gathered'].shape)} baked into the captured "
                                           f"graph no longer fits {want} (the number of Gaussians or the world size changed): re-capture")
                    e["gathered"] = None
                if e["gathered"] is None:                  # kept afterwards: a captured second half reads it at a fixed address
                    e["gathered"] = torch.empty(want, dtype=f.dtype, device=f.device)
                try:
                    works.append(dist.all_gather_into_tensor(e["gathered"], f, group=self.group, async_op=True))
                except (RuntimeError, NotImplementedError):      # backends without the flat variant
                    works.append(dist.all_gather(list(e["gathered"].unbind(0)), f, group=self.group, async_op=True))
                self.bytes_last += f.numel() * 4
        for w in works:
            w.wait()
        if self.scale != 1.0:
            for e in self.parked:
                if e["head"] is e:
                    e["flat"].mul_(self.scale)
        if ev is not None:
            ev[1].record()
            self._events.append(ev)

    def parked_gradients(self):
        """Deferred mode, after `exchange_parked` (capturable): rasterizer input name -> reduced gradient summed over the
        step's renders, with the SH block rebuilt from the gathered factors (d3ga_sh_grad_from_views)."""
        from . import _lib
        from ._lib import check, dptr, stream_handle
        out = {}
        for e in self.parked:
            mine = dict(e["parts"]) if e.get("head", e) is e else {}       # (a merged entry's parts are inside its head's sum)
            sh = e["sh"]
            if sh is not None:
                P, M, deg, means3D = sh["P"], sh["M"], sh["sh_degree"], sh["means3D"]
                g_sh = torch.empty((P, M, 3), dtype=torch.float32, device=means3D.device)
                g = e["gathered"].view(-1, P + 1, 3)       # (world, P+1, 3), or (world, k, P+1, 3) from a view-batched render: world x k views
                check(_lib.lib().d3ga_sh_grad_from_views(P, M, deg, g.shape[0], dptr(means3D), dptr(g), 3 * (P + 1),
                                                         dptr(g[0, P]), 3 * (P + 1), self.scale, dptr(g_sh), stream_handle()),
                      "d3ga_sh_grad_from_views")
                mine["shs"] = g_sh
            for k, v in mine.items():
                out[k] = v if k not in out else out[k] + v
        return out

    def exchange_ms(self):
        """Mean milliseconds per exchange since the last call (synchronises); None if nothing was timed."""
        if not self._events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._events]
        self._events = []
        return sum(ms) / len(ms)


class BucketedGradReducer:
    """Parameter-level gradient averaging that OVERLAPS with the backward: one asynchronous all-reduce per bucket, issued from
    post-accumulate hooks the moment the last gradient of the bucket has been written ("bucketed all-reduce
    launched from autograd post-accumulate hooks").

    This is the exchange for configurations whose rasterizer inputs are VIEW-DEPENDENT -- the reference's main configuration
    (configs/actorshq_actor02.yml: use_shs false; colour and opacity from ColorField(view_dir, camera / frame encodings),
    models/cage_net.py:232-258) -- where `ViewShardedGrads`' cut is invalid and every parameter gradient has to be summed.

        red = BucketedGradReducer([color_field.parameters(), canon_field.parameters(), [color_feat], ...])
        for step in ...:
            red.begin_step()                  # drops the gradients (autograd then hands every parameter a fresh tensor: no zero fill)
            loss.backward()                   # hooks fire: ColorField's bucket is on the wire while the deform backward still runs
            red.finish()                      # waits for the collectives, divides by the world size
            clip_grad_norm_(...); optimizer.step()        # AFTER the reduce (models/trainer.py:188)

    A bucket of ONE tensor is reduced in place (the (P,64) colour features: 35 MB at 135k Gaussians -- no staging copy); a
    bucket of several tensors (the ~10 weights and biases of a network, ~0.3 MB) goes through one flat staging buffer: one
    collective instead of ten.  Buckets are reduced in the order their gradients complete, which is the same on every rank
    (the same autograd graph), so the collectives match up.  `finish()` raises if a bucket never completed (a parameter
    that took no part in the loss): silently skipping it would desynchronise the ranks' collectives.
    No-op exchange (hooks still count) when torch.distributed is not initialised or the world size is 1, unless `always`."""

    def __init__(self, buckets, group=None, always=False, timing=False):
        self.group, self.always, self.timing = group, always, timing
        self.buckets = []
        for b in buckets:
            ps = [p for p in b if p.requires_grad]
            if ps:
                self.buckets.append(ps)
        self._flat = [None if len(b) == 1 else torch.empty(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device)
                      for b in self.buckets]
        self._left, self._work, self._events = [0] * len(self.buckets), [], []
        self._handles = []
        for i, b in enumerate(self.buckets):
            for p in b:
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.begin_step()

    def _active(self):
        return self.always or (dist.is_initialized() and dist.get_world_size(self.group) > 1)

    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def nbytes(self):
        return 4 * sum(p.numel() for b in self.buckets for p in b)

    def _make_hook(self, i):
        def hook(param):
            # a non-contiguous .grad (autograd may hand one over for an expanded / transposed leaf) cannot be all-reduced in
            # place, and reshape(-1) of it is a COPY -- the averaged values would land in a temporary
            if param.grad is not None and not param.grad.is_contiguous():
                param.grad = param.grad.contiguous()
            self._left[i] -= 1
            if self._left[i] == 0:
                self._launch(i)
        return hook

    def _launch(self, i):
        if not self._active():
            return
        b = self.buckets[i]
        timed = self.timing and b[0].is_cuda and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if len(b) == 1:
            t = b[0].grad
        else:
            t = self._flat[i]
            torch._foreach_copy_(list(t.split([p.numel() for p in b])), [p.grad.reshape(-1) for p in b])      # (contiguous: the hook saw to it)
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if dist.is_initialized() else None
        self._work.append((i, w, e0 if timed else None))

    def begin_step(self):
        for b in self.buckets:
            for p in b:
                p.grad = None
        self._left = [len(b) for b in self.buckets]
        self._work = []

    def finish(self):
        """Wait for every bucket's collective and turn the sums into means.  Returns the number of buckets reduced."""
        if any(n != 0 for n in self._left):
            missing = [i for i, n in enumerate(self._left) if n != 0]
            raise RuntimeError(f"BucketedGradReducer.finish(): buckets {missing} did not receive all their gradients in this backward "
                               "(a parameter outside the loss?): their collectives were not issued")
        if not self._active():
            return 0
        inv = 1.0 / float(self.world())
        for i, w, e0 in self._work:
            if w is not None:
                w.wait()
            b = self.buckets[i]
            if len(b) == 1:
                b[0].grad.mul_(inv)
            else:
                parts = list(self._flat[i].split([p.numel() for p in b]))
                torch._foreach_mul_(parts, inv)
                torch._foreach_copy_([p.grad for p in b], [q.view_as(p.grad) for p, q in zip(b, parts)])
            if e0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self._events.append((e0, e1))
        n = len(self._work)
        self._work = []
        return n

    def exchange_ms(self):
        """Mean milliseconds from a bucket's launch to its averaged gradients (HIP events; `timing=True`) since the last call --
        an upper bound of the exposed wire time: buckets launched early complete behind the rest of the backward."""
        if not self._events:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._events) / len(self._events)
        self._events = []
        return ms

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []

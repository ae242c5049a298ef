"""A training step captured once in a hipGraph and replayed with new inputs every step.

The step of this path is ~45 small launches; issued eagerly from Python it is host-bound (~0.75 ms of enqueue work against
0.6 ms of GPU work at 500k Gaussians, and 2x host-bound at the reference's own 135k-Gaussian size).  A captured graph removes
the per-launch host cost -- but the reference draws a NEW camera (and target image, and pose) every step
(datasets/actorshq_dataset.py:229, models/trainer.py:91-110), so whatever changes from step to step has to live in STATIC
device buffers ("slots") that the captured kernels read:

    slot = CameraSlot(W, H)                       # cameras.py: matrices + tan(FoV/2) in one device buffer
    target = torch.empty(3, H, W, device="cuda")  # any tensor the step reads can be a slot
    batch["camera_slot"] = slot
    step = CapturedStep(lambda: train_step(batch, target), params=model.parameters(), slots={"target": target}, camera=slot)
    for frame in loader:
        step.replay(camera=frame, target=frame["image"])      # two async copies + one graph launch

Constraints (hipGraph): fixed shapes, fixed raster size (grid dimensions are baked), the binning capacity must be static
(`rasterizer.set_capacity_policy("static", n)`), no host synchronisation inside `step_fn`.  The frozen capacity is WATCHED:
every `check_every` replays the rasterizer's 16-byte counter block is copied to pinned host memory asynchronously and looked
at once it has arrived -- a trainer whose splats grow past the capacity gets a `CapacityOverflowError` within
2 x check_every replays instead of silently truncated tile lists (`check_overflow()` asks synchronously).  Gradients land in the same static `.grad` tensors on every replay.  Do not keep the LOSS of an earlier eager
step alive across the capture: it keeps that step's autograd nodes (created on the default stream) alive, autograd then
synchronises the capture with the default stream, and ending the capture crashes inside the HIP runtime.
"""
import torch

from . import _lib


# Stream-capture error mode.  The default ("global") makes a HIP call from ANY thread illegal while a capture is in progress --
# and a process that holds an RCCL process group has such a thread: ProcessGroupNCCL's watchdog polls the events of earlier
# collectives (hipEventQuery).  Seen in round 4: "operation not permitted when stream is capturing" from the watchdog thread,
# which then terminates the process (intermittent: it depends on whether an eager collective is still being retired when a
# later capture starts).  "thread_local" restricts the check to the capturing thread.
_CAPTURE_MODE = "thread_local"


class CapacityOverflowError(RuntimeError):
    """A replayed step produced more (tile, Gaussian) duplicates than the binning capacity frozen into its graph."""


class _capture_forwards:
    """Context manager: collects (binning tensor, capacity) of EVERY rasterizer forward issued inside (the renders a
    captured graph contains: RGB + silhouette, k views per rank, several packages)."""

    def __enter__(self):
        from . import rasterizer
        self._mod, self._prev = rasterizer, rasterizer._capture_log
        rasterizer._capture_log = self.log = []
        return self.log

    def __exit__(self, *exc):
        self._mod._capture_log = self._prev
        return False


class _OverflowWatch:
    """Asynchronous look at the counter blocks (D, overflow flag, longest list, visible) of ALL rasterizer forwards a captured
    graph contains (`forwards`: the list `_capture_forwards` collected during the capture -- ADVICE r3: the first version
    watched only the LAST forward seen, so a truncated RGB render in front of a silhouette render went unreported, and a
    step without any render read a stale block of an earlier eager call).  The blocks live in the graph's private pool, so
    the tensors seen at capture time stay the ones every replay writes."""

    def __init__(self, check_every, forwards):
        self.every = max(int(check_every), 1)
        seen, self.blocks, self.caps = set(), [], []
        for binning, cap in forwards:
            if binning.data_ptr() in seen:      # (geometry_reuse: the second render shares the first one's binning buffer)
                continue
            seen.add(binning.data_ptr())
            self.blocks.append(binning[:16].view(torch.int32))
            self.caps.append(cap)
        n = len(self.blocks)
        self.host = torch.zeros(max(n, 1), 4, dtype=torch.int32).pin_memory() if n else None
        self.event, self.count = None, 0

    def _copy(self):
        for i, b in enumerate(self.blocks):
            self.host[i].copy_(b, non_blocking=True)

    def _inspect(self):
        self.event = None
        for i, cap in enumerate(self.caps):
            d, flag = int(self.host[i, 0]) & 0xFFFFFFFF, int(self.host[i, 1])
            if flag:
                raise CapacityOverflowError(
                    f"captured step, render {i} of {len(self.caps)}: {d} (tile, Gaussian) duplicates exceed the binning capacity {cap} "
                    f"frozen into the graph -- the tile lists of the last replays were truncated.  Re-capture with "
                    f"rasterizer.set_capacity_policy('static', n) for n >= {int(1.25 * d)}.")

    def after_replay(self):
        if not self.blocks:
            return
        self.count += 1
        if self.event is not None and self.event.query():
            self._inspect()
        if self.event is None and self.count % self.every == 0:
            self._copy()
            self.event = torch.cuda.Event()
            self.event.record()

    def check_now(self):
        """dict(D, capacity, max_tile) of the render with the most duplicates (+ "renders": one dict per render)."""
        if not self.blocks:
            return None
        self._copy()
        torch.cuda.current_stream().synchronize()
        self._inspect()
        per = [{"D": int(self.host[i, 0]) & 0xFFFFFFFF, "capacity": self.caps[i], "max_tile": int(self.host[i, 2]) & 0xFFFFFFFF}
               for i in range(len(self.caps))]
        worst = dict(max(per, key=lambda r: r["D"]))
        worst["renders"] = per
        return worst


class TensorSlot:
    """A device cell that holds the ADDRESS of the tensor a captured kernel reads.

    `CapturedStep(slots={"target": buf})` copies this step's target image into a static buffer before every replay (25 MB at
    1080p).  When the candidates are resident anyway (the ground-truth images of the cameras of a capture rig), a kernel
    can read the address instead: `slot.set(t)` repoints the cell with one asynchronous 8-byte copy, and ops that accept a
    TensorSlot (`losses.l1_loss(img, slot)`) read `*cell` when they start.  Every tensor passed to `set` must have the
    slot's shape, be float32, contiguous and 16-byte aligned, and must stay alive while replays that read it are in flight
    (the slot keeps the last `_RING` alive itself)."""

    _RING = 16

    def __init__(self, first, arena=None, index=0):
        """arena / index: keep the cell inside a `cameras.CameraSlot(..., cells=n)` -- `set()` then only stages the address on the
        host and the slot's next `set(batch)` / `flush()` carries it to the device together with the camera (one H2D copy per
        replay instead of two; `CapturedStep.replay` orders the two calls)."""
        self.shape, self.device = tuple(first.shape), first.device
        self.arena, self.index = arena, int(index)
        if arena is not None:
            self.cell = arena.cell(self.index)
        else:
            self.cell = torch.zeros(1, dtype=torch.int64, device=self.device)
            self._stage = [torch.zeros(1, dtype=torch.int64) for _ in range(self._RING)]
            if self.device.type == "cuda":
                self._stage = [t.pin_memory() for t in self._stage]
        self._events, self._alive, self._next = [None] * self._RING, [None] * self._RING, 0
        self.current = None
        self.set(first)
        if arena is not None:
            arena.flush()

    def numel(self):
        n = 1
        for d in self.shape:
            n *= d
        return n

    def set(self, t):
        if tuple(t.shape) != self.shape or t.dtype != torch.float32 or t.device != self.device or not t.is_contiguous():
            raise ValueError(f"TensorSlot{self.shape}: got {tuple(t.shape)} {t.dtype} on {t.device} (contiguous float32 of the slot's shape)")
        if t.data_ptr() % 16:
            raise ValueError("TensorSlot: the tensor must be 16-byte aligned")
        i = self._next
        self._next = (i + 1) % self._RING
        if self.arena is not None:
            self.arena.stage_cell(self.index, t.data_ptr())    # travels with the arena's next copy
        else:
            if self._events[i] is not None:
                self._events[i].synchronize()              # the copy that last read this staging word has run
            self._stage[i][0] = t.data_ptr()
            self.cell.copy_(self._stage[i], non_blocking=True)
            if self.cell.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
                self._events[i] = ev
        self._alive[i] = t                                 # (the last _RING targets stay alive: replays in flight read them)
        self.current = t
        return self


def _update_slots(step, camera, values):
    """Per-replay inputs: tensors first (a TensorSlot that lives in a CameraSlot only stages its address), then the camera,
    whose ONE host-to-device copy carries the staged cells along; arenas that were touched without a camera are flushed."""
    if camera is not None and step.camera is None:
        raise ValueError("this step was captured without a CameraSlot")
    dirty = []
    for name, v in values.items():
        slot = step.slots[name]
        if isinstance(slot, TensorSlot):
            slot.set(v)                                    # repoint: 8 bytes instead of the tensor
            if slot.arena is not None and slot.arena not in dirty:
                dirty.append(slot.arena)
        else:
            slot.copy_(v if torch.is_tensor(v) else torch.as_tensor(v), non_blocking=True)
    if camera is not None:
        step.camera.set(camera)
        dirty = [a for a in dirty if a is not step.camera]
    for a in dirty:
        a.flush()


class CapturedStep:
    def __init__(self, step_fn, params=(), slots=None, camera=None, warmup=2, check_every=16):
        """step_fn(): the whole step (forward, loss, backward[, optimizer]) reading its per-step inputs from `slots`
        (name -> static device tensor) and, for the camera, from `camera` (a cameras.CameraSlot placed in the batch).
        params: the leaves whose `.grad` the step produces -- their gradients are dropped between the warm-up steps and the
        capture, so that the captured backward WRITES fresh static gradient tensors instead of accumulating into the
        warm-up's (every replay then leaves exactly this step's gradients in `p.grad`)."""
        self.slots = dict(slots or {})
        self.camera = camera
        self.step_fn = step_fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # allocator warm-up on a capture-capable stream
            for _ in range(max(int(warmup), 1)):
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with _capture_forwards() as forwards, torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
            self.result = step_fn()
        torch.cuda.synchronize()
        self._watch = _OverflowWatch(check_every, forwards)

    def check_overflow(self):
        """Synchronous look at the last replay's duplicate count: raises CapacityOverflowError, else returns
        dict(D, capacity, max_tile) (None when the step contains no rasterizer forward)."""
        return self._watch.check_now()

    def replay(self, camera=None, **values):
        """camera: a batch dict (R, T, FoVx, FoVy, width, height) written into the CameraSlot; values: name -> tensor / array
        copied into the slot of that name.  Everything is enqueued on the current stream; returns what step_fn returned at
        capture time (static tensors holding this replay's results once the stream reaches them)."""
        _update_slots(self, camera, values)
        self.graph.replay()
        _lib.replay_epoch[0] += 1          # caches keyed by a tensor's version counter: a replay may have written it in place
        self._watch.after_replay()
        return self.result


class CapturedCutStep:
    """A camera-sharded training step as TWO hipGraphs with the gradient exchange between them.

    With `ViewShardedGrads` the collectives sit inside the rasterizer's backward, i.e. in the middle of the step: a captured
    step would contain them (possible with RCCL, but a mis-captured collective hangs a multi-GPU job instead of slowing
    it), an eager step is bound by the host (~0.75 ms of Python per step against ~0.5 ms of GPU work at C3).  Here the step
    is cut where the exchange is:

        graph A   upstream()            parameters -> the tensors that enter the rasterizer (with their autograd graph)
                  loss_fn(leaves)       render + loss on detached copies; backward down to the rasterizer's inputs, whose
                                        gradients the rasterizer parks in `sync` (ViewShardedGrads.deferred)
        eager     sync.exchange_parked()    one all-reduce + one all-gather on static buffers, any backend
        graph B   sync.parked_gradients()   reduced gradients (+ the SH block rebuilt from the gathered factors), pushed
                                        through upstream's autograd graph into the parameters' .grad

    upstream() -> dict name -> tensor; the names the rasterizer differentiates are "means3D", "opacities" (also
    "opacity_logits"), "cov3D_precomp" | "scales" + "rotations", "shs" | "colors_precomp" (`rgb`); other entries are
    passed through.  loss_fn(pkg) -> scalar loss; it must render with `grad_sync=sync`; it may render more than once (the
    reference's RGB + silhouette pass: every render parks its own buffers and is exchanged) and may add loss terms that
    read package tensors directly (regularisers, energies): their gradients stay on the detached leaves and are added to the
    reduced rasterizer gradients before graph B continues the backward.  Such extra terms must be view-independent, like
    everything upstream of the cut (they are NOT reduced: identical on every rank by construction).
    """

    _ALIASES = {"opacity_logits": "opacities", "rgb": "colors_precomp"}

    def __init__(self, upstream, loss_fn, sync, params=(), slots=None, camera=None, warmup=2, check_every=16):
        self.slots = dict(slots or {})
        self.camera = camera
        self.sync = sync
        was = sync.deferred
        sync.deferred = True
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                  # allocator warm-up on a capture-capable stream
                for _ in range(max(int(warmup), 1)):
                    self._eager(upstream, loss_fn)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for p in params:
                p.grad = None
            self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            sync.begin_step()
            with _capture_forwards() as forwards, torch.cuda.graph(self.graph_a, capture_error_mode=_CAPTURE_MODE):
                up, pkg, self.result = self._to_the_cut(upstream, loss_fn)
            sync.exchange_parked()             # (graph A has not run: this call's collectives move unspecified data, on every rank alike)
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), capture_error_mode=_CAPTURE_MODE):
                self._from_the_cut(up, pkg)
            torch.cuda.synchronize()
            sync.frozen = True                 # the parked buffers are baked into the two graphs now
            self._watch = _OverflowWatch(check_every, forwards)
        finally:
            sync.deferred = was

    def check_overflow(self):
        return self._watch.check_now()

    def _to_the_cut(self, upstream, loss_fn):
        up = upstream()
        pkg = {k: (v.detach().requires_grad_(True) if torch.is_tensor(v) and v.is_floating_point() and v.requires_grad else v)
               for k, v in up.items()}
        loss = loss_fn(pkg)
        loss.backward()
        return up, pkg, loss

    def _from_the_cut(self, up, pkg):
        """Everything that reached the cut goes on: the reduced rasterizer-input gradients of ALL renders of the step
        (`parked_gradients`) PLUS whatever other loss terms left on the detached leaves (a regulariser on the scales, an
        energy the package carries: models/cage_net.py:225-226, train.py:203) -- the rasterizer's backward returns nothing
        to the leaves in deferred mode, so `leaf.grad` holds exactly those other terms."""
        grads = self.sync.parked_gradients()
        outs, gs = [], []
        for k, v in up.items():
            if not (torch.is_tensor(v) and v.requires_grad):
                continue
            g = grads.get(self._ALIASES.get(k, k))
            g = None if g is None else g.reshape(v.shape)
            extra = pkg[k].grad if torch.is_tensor(pkg.get(k)) else None
            if extra is not None:
                g = extra if g is None else g + extra
            if g is not None:
                outs.append(v)
                gs.append(g)
        torch.autograd.backward(outs, gs)

    def _eager(self, upstream, loss_fn):
        self.sync.begin_step()
        up, pkg, loss = self._to_the_cut(upstream, loss_fn)
        self.sync.exchange_parked()
        self._from_the_cut(up, pkg)
        return loss

    def replay(self, camera=None, **values):
        """As CapturedStep.replay; the two collectives are issued on the current stream between the two graph launches."""
        _update_slots(self, camera, values)
        self.graph_a.replay()
        self.sync.exchange_parked()
        self.graph_b.replay()
        _lib.replay_epoch[0] += 1
        self._watch.after_replay()
        return self.result



class ReplayWatchdog:
    """A deadline for asynchronous device work that contains collectives -- a captured step with RCCL nodes
    (`bench.py --graph-collectives`) or the eager exchange between the two graphs of `CapturedCutStep`.  A peer that died, or
    ranks that disagree about the collectives of a step, do not raise: the GPU simply never finishes and the job hangs until
    somebody kills it.  `arm(tag)` records an event on the current stream (one event record per call, nothing else on the hot
    path); a daemon thread polls the armed events and, when one is still pending `timeout_s` after it was armed, calls
    `on_timeout(tag, seconds)` -- by default it reports rank and tag on stderr and ends the process with exit code 124, so a
    launcher (torchrun) sees a failed rank and tears the group down instead of waiting for ever.
    `event_factory` is injectable for tests (anything with record() and query())."""

    def __init__(self, timeout_s=120.0, poll_s=0.05, on_timeout=None, event_factory=None):
        import collections
        import threading
        self.timeout_s, self.poll_s = float(timeout_s), float(poll_s)
        self.on_timeout = on_timeout or self._abort
        self._factory = event_factory or (lambda: torch.cuda.Event())
        self._pending = collections.deque()
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self.fired = None
        self._thread = threading.Thread(target=self._run, name="d3ga-replay-watchdog", daemon=True)
        self._thread.start()

    def arm(self, tag=None):
        import time
        ev = self._factory()
        ev.record()
        with self._lock:
            # Work completes in order, so only the OLDEST pending event decides; when 64 are pending (a hung device with the
            # host still arming) the NEW event is dropped -- trimming from the left kept moving the head's arm time forward
            # and the timeout fired late or never.
            if len(self._pending) < 64:
                self._pending.append((ev, tag, time.monotonic()))

    def _run(self):
        import time
        while not self._stop.wait(self.poll_s):
            with self._lock:
                while self._pending and self._pending[0][0].query():
                    self._pending.popleft()
                head = self._pending[0] if self._pending else None
            if head is not None and time.monotonic() - head[2] > self.timeout_s and self.fired is None:
                self.fired = (head[1], time.monotonic() - head[2])
                self.on_timeout(*self.fired)

    @staticmethod
    def _abort(tag, seconds):
        import os
        import sys
        rank = os.environ.get("RANK", "0")
        print(f"[d3ga watchdog] rank {rank}: device work armed as {tag!r} still pending after {seconds:.1f} s -- a collective "
              f"that never completes (dead peer or mismatched collectives); aborting this rank", file=sys.stderr, flush=True)
        os._exit(124)

    def close(self):
        self._stop.set()
        self._thread.join(timeout=1.0)

"""world_size-2 gloo test of the camera-sharded gradient reduction (d3ga_amd.dist), CPU only."""
import os
import socket

import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from d3ga_amd import dist as dd
    r, _, w = dd.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    params = [torch.randn(5, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    flat = dd.FlatGrads(params)
    views = dd.shard_views(5, rank, world)
    # "render" each view: a loss whose gradient depends on the view index
    flat.zero_()
    for v in views:
        loss = sum(((v + 1.0) * p).sum() for p in params)
        loss.backward()
    assert params[0].grad.data_ptr() == flat.buffer.data_ptr(), "autograd must accumulate into the flat buffer"
    flat.all_reduce_mean()
    expect = sum(v + 1.0 for v in range(5)) / world
    ok = all(torch.allclose(p.grad, torch.full_like(p, expect)) for p in params)
    # a stray gradient tensor (set_to_none style) must be re-homed before reducing
    params[1].grad = torch.full((7,), float(rank + 1))
    flat.buffer[:15] = float(rank + 1)
    flat.all_reduce_mean()
    ok = ok and torch.allclose(params[1].grad, torch.full((7,), 1.5)) and params[1].grad.data_ptr() == flat.buffer[15:].data_ptr()
    # per-tensor reducer (the one bench.py uses)
    red = dd.GradReducer(params)
    red.zero()
    assert all(p.grad is None for p in params)
    for v in views:
        sum(((v + 1.0) * p).sum() for p in params).backward()
    red.all_reduce_mean()
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p, expect)) for p in params)
    # ViewShardedGrads.verify_inputs: identical inputs pass, rank-dependent ("view-dependent") inputs raise on every rank
    sync = dd.ViewShardedGrads(verify_steps=2)
    same = torch.arange(12.0).reshape(4, 3)
    sync.verify_inputs({"means3D": same, "opacities": None})
    raised = False
    try:
        sync.verify_inputs({"means3D": same, "colors_precomp": same + 1e-3 * rank})
    except dd.ViewDependentInputError as e:
        raised = "colors_precomp" in str(e) and "means3D" not in str(e).split("differ")[0]
    ok = ok and raised
    sync.verify_inputs({"colors_precomp": same + rank})          # verify_steps exhausted: no collective, no error
    # deferred exchange (graph.CapturedCutStep): what the rasterizer's backward parks is reduced in place by exchange_parked,
    # the gather buffer keeps its address from call to call, and the colours-precomp path needs no SH rebuild
    flat_buf = torch.full((4 * 4,), float(rank + 1))
    parts = {"means3D": flat_buf[:12].view(4, 3), "opacities": flat_buf[12:].view(4, 1)}
    factor = torch.full((5, 3), float(10 * (rank + 1)))
    sync.begin_step()
    sync.park(flat_buf, factor, parts, sh=None)
    # a second render of the same step (the reference's silhouette pass) parks its own buffers; the step's gradient of an
    # input is the SUM over its renders
    flat2 = torch.full((4 * 4,), float(10 * (rank + 1)))
    sync.park(flat2, None, {"means3D": flat2[:12].view(4, 3), "opacities": flat2[12:].view(4, 1)}, sh=None)
    sync.exchange_parked()
    first = sync.parked[0]["gathered"].data_ptr()
    ok = ok and torch.allclose(sync.parked_gradients()["means3D"], torch.full((4, 3), 1.5 + 15.0))   # mean of 1, 2 + mean of 10, 20
    ok = ok and torch.allclose(sync.parked[0]["gathered"][:, 0, 0], torch.tensor([10.0, 20.0])) and sync.parked[1]["gathered"] is None
    flat_buf.fill_(float(rank + 3))
    flat2.zero_()
    sync.exchange_parked()
    ok = ok and sync.parked[0]["gathered"].data_ptr() == first and torch.allclose(parts["opacities"], torch.full((4, 1), 3.5))
    # once a graph has captured the buffers their shapes are frozen: a changed gather shape is an error, not a silent reallocation
    sync.frozen = True
    sync.parked[0]["factor"] = torch.zeros(7, 3)
    try:
        sync.exchange_parked()
        ok = False
    except RuntimeError as e:
        ok = ok and "re-capture" in str(e)
    try:
        sync.begin_step()
        ok = False
    except RuntimeError:
        pass
    out[rank] = bool(ok)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_flat_grad_allreduce_two_ranks():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        out = m.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(out) == {0: True, 1: True}


def _bucket_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from d3ga_amd import dist as dd
    dd.init_process_group(backend="gloo")
    torch.manual_seed(0)                                   # identical replicas
    net_a, net_b = torch.nn.Linear(6, 8), torch.nn.Linear(8, 3)
    feat = torch.randn(20, 6, requires_grad=True)
    unused = torch.randn(3, requires_grad=True)
    red = dd.BucketedGradReducer([net_b.parameters(), net_a.parameters(), [feat]])
    order = []
    launch = red._launch
    red._launch = lambda i: (order.append(i), launch(i))[1]
    ok = True
    for step in range(3):
        # a "view-dependent" loss: the ranks see different cameras (inputs scaled by rank + 1)
        def loss_of(view):
            x = feat * (view + 1.0)
            return (net_b(torch.tanh(net_a(x))) ** 2).mean()
        red.begin_step()
        ok = ok and all(p.grad is None for p in net_a.parameters())
        loss_of(rank).backward()
        ok = ok and red.finish() == 3
        mine = [p.grad.clone() for p in list(net_a.parameters()) + list(net_b.parameters()) + [feat]]
        # the sequential two-view mean on this rank alone
        for p in list(net_a.parameters()) + list(net_b.parameters()) + [feat]:
            p.grad = None
        red.close()
        ((loss_of(0) + loss_of(1)) / world).backward()
        ref = [p.grad.clone() for p in list(net_a.parameters()) + list(net_b.parameters()) + [feat]]
        ok = ok and all(torch.allclose(a, b, rtol=1e-5, atol=1e-7) for a, b in zip(mine, ref))
        red = dd.BucketedGradReducer([net_b.parameters(), net_a.parameters(), [feat]])
        launch = red._launch
        red._launch = lambda i, launch=launch: (order.append(i), launch(i))[1]
    # buckets go on the wire in the order their gradients complete: the last layer's first (overlap with the rest of the backward)
    ok = ok and order[:3][0] == 0 and sorted(order[:3]) == [0, 1, 2]
    # a bucket that never completes is an error on every rank, not a silent skip
    red.close()
    red = dd.BucketedGradReducer([net_a.parameters(), [unused]])
    red.begin_step()
    (net_a(feat) ** 2).mean().backward()
    try:
        red.finish()
        ok = False
    except RuntimeError as e:
        ok = ok and "buckets [1]" in str(e)
    for _, w, _ in red._work:                             # (the completed bucket's collective is in flight on both ranks: drain it)
        w.wait()
    out[rank] = bool(ok)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_bucketed_grad_reducer_two_ranks():
    """BucketedGradReducer (the exchange of the ColorField configuration: view-dependent rasterizer inputs): per-bucket asynchronous
    all-reduces from post-accumulate hooks give the mean over the ranks' views, equal to the sequential two-view mean."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        out = m.dict()
        procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(out) == {0: True, 1: True}


def test_shard_views_covers_everything():
    from d3ga_amd.dist import shard_views
    for n, w in ((8, 8), (8, 2), (5, 2), (3, 4)):
        got = sorted(v for r in range(w) for v in shard_views(n, r, w))
        assert got == list(range(n))


def _eight_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from d3ga_amd import dist as dd
    r, _, w = dd.init_process_group(backend="gloo")
    ok = (r, w) == (rank, world)
    ok = ok and dd.shard_views(world, rank, world) == [rank]
    # per-tensor reducer: the mean over the ranks' views = the mean of `world` sequential views
    torch.manual_seed(0)
    params = [torch.randn(5, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    red = dd.GradReducer(params)
    red.zero()
    sum(((rank + 1.0) * p).sum() for p in params).backward()
    red.all_reduce_mean()
    expect = sum(v + 1.0 for v in range(world)) / world
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p, expect)) for p in params)
    # the cut exchange of the SH path: a planar buffer summed (x 1 / world), and the (P + 1, 3) colour factor of EVERY rank
    # gathered in rank order (row P carries the rank's camera position) -- what d3ga_sh_grad_from_views rebuilds dL/dsh from
    sync = dd.ViewShardedGrads()
    ok = ok and sync.world == world and abs(sync.scale - 1.0 / world) < 1e-12
    P = 6
    flat = torch.full((P * 4,), float(rank + 1))
    factor = torch.full((P + 1, 3), float(100 * (rank + 1)))
    factor[P] = torch.tensor([float(rank), 0.5, -1.0])
    gathered = sync.exchange(flat, factor)
    ok = ok and tuple(gathered.shape) == (world, P + 1, 3)
    ok = ok and torch.allclose(flat, torch.full((P * 4,), expect))
    ok = ok and torch.allclose(gathered[:, 0, 0], 100.0 * torch.arange(1, world + 1, dtype=torch.float32))
    ok = ok and torch.allclose(gathered[:, P, 0], torch.arange(world, dtype=torch.float32))
    # bucketed reducer (the ColorField configuration): equals the sequential mean over `world` views
    torch.manual_seed(0)
    net = torch.nn.Linear(6, 4)
    feat = torch.randn(10, 6, requires_grad=True)
    bred = dd.BucketedGradReducer([net.parameters(), [feat]])
    loss_of = lambda view: (net(feat * (view + 1.0)) ** 2).mean()
    bred.begin_step()
    loss_of(rank).backward()
    ok = ok and bred.finish() == 2
    mine = [p.grad.clone() for p in list(net.parameters()) + [feat]]
    bred.close()
    for p in list(net.parameters()) + [feat]:
        p.grad = None
    (sum(loss_of(v) for v in range(world)) / world).backward()
    ref = [p.grad.clone() for p in list(net.parameters()) + [feat]]
    ok = ok and all(torch.allclose(a, b, rtol=1e-5, atol=1e-7) for a, b in zip(mine, ref))
    out[rank] = bool(ok)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_reducers_and_the_cut_exchange_at_eight_ranks():
    """nothing rank-count dependent had run above N = 2.  The reducers, the cut exchange (all-reduce + the
    all-gather of eight colour factors in rank order) and the bucketed reducer at world_size 8 over gloo on CPU: the
    reduced gradients equal the mean of eight sequential views."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        out = m.dict()
        procs = [ctx.Process(target=_eight_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
            assert p.exitcode == 0
        assert dict(out) == {r: True for r in range(world)}

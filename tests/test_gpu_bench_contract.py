"""bench.py prints ONE JSON line with the fields the driver reads (and the roofline / cpu_baseline objects)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_small_workload():
    d = _run("--workload", "C1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "frames/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 0.02
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    # the view-batched regime beside the single-view headline (round 6): k = 4 cameras of the pose in one grid per stage
    b = d["batched_views"]
    assert "error" not in b, b
    assert b["views"] == 4 and b["value"] > 0 and abs(b["value"] - 4 * 1e3 / b["ms_per_step"]) / b["value"] < 0.02
    assert b["roofline"]["bound"] == "hbm" and 0 < b["roofline"]["frac"] < 1
    # the roofline kernels are timed with and without back-to-back launches (the burst keeps caches warm)
    for k in ("composite_fwd", "composite_bwd"):
        assert d["kernels"][k]["ms_single_launch"] >= 0.5 * d["kernels"][k]["ms"]


def test_bench_two_ranks_as_the_driver_launches_it():
    """The N > 1 launch line of the driver (`python -m torch.distributed.run ... bench.py --gpus N ...`), with both ranks on
    the test box's one GPU over gloo (`--backend gloo --single-device`; the scaling runs use RCCL, one GPU per rank):
    rank 0 prints one line, the value is the whole-job rate over the max-over-ranks time, the gradient exchange ran."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--single-device", "--workload", "C1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["views_per_step"] == 2
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) / d["value"] < 0.02          # whole-job frames / max-over-ranks time
    assert d["config"]["grad_exchange"].startswith("cut") and d["config"]["grad_exchange_bytes_per_rank"] > 0


def test_bench_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset -- the way the driver starts the N = 1 run -- must start two ranks itself
    (re-exec under torch.distributed.run) and report them: a line labelled n_gpus 2 from one rank would waste a scaling slot."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
           "--single-device", "--workload", "C1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["distributed"]["nranks_seen"] == 2 and d["config"]["views_per_step"] == 2


def test_bench_refuses_a_world_size_that_is_not_gpus():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--single-device", "--workload", "C1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_color_train_step_two_ranks():
    """BASELINE configs[3] entry point: `bench.py --train-step color --gpus 2` (self-launched, both ranks on the test box's one
    GPU over gloo): the actor02-shaped step with all three field networks, parameter gradients averaged by per-bucket
    asynchronous all-reduces; every rank counted, replicas bit-identical after the optimised steps, loss finite."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--train-step", "color", "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--backend", "gloo", "--single-device", "--workload", "C1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["distributed"]["nranks_seen"] == 2 and d["replicas_identical"] is True and d["loss_finite"] is True
    assert d["config"]["views_per_step"] == 2 and d["config"]["buckets"] == 7 and d["config"]["grad_exchange_bytes_per_rank"] > 0
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) / d["value"] < 0.02


def test_bench_eight_ranks_on_one_gpu():
    """the N = 8 forms of the bench line (BASELINE configs[3] / [4] name 8 views on 8 GPUs) had never executed at any
    rank count above 2.  Eight self-launched ranks share the test box's GPU over gloo: every rank is counted, the value is the
    whole-job rate over the max-over-ranks time, eight views per step went through the cut exchange."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--backend", "gloo",
           "--single-device", "--workload", "C1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["distributed"]["nranks_seen"] == 8 and d["config"]["views_per_step"] == 8
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) / d["value"] < 0.02
    assert d["config"]["grad_exchange"].startswith("cut") and d["config"]["grad_exchange_bytes_per_rank"] > 0


def test_bench_color_train_step_eight_ranks():
    """... and the colour training step (the actor02-shaped configuration, parameter gradients averaged by the bucketed reducer) at
    eight ranks, three optimised steps: every rank counted, replicas bit-identical afterwards, loss finite."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--train-step", "color", "--gpus", "8", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--single-device", "--workload", "C1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["distributed"]["nranks_seen"] == 8 and d["replicas_identical"] is True and d["loss_finite"] is True
    assert d["config"]["views_per_step"] == 8 and d["config"]["grad_exchange_bytes_per_rank"] > 0

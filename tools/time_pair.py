import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from d3ga_amd import rasterizer as R
dev = torch.device("cuda", 0)
frame = bench.Frame("C3", dev, 0)
def zero():
    for q in frame.params.values(): q.grad = None
for _ in range(5):
    zero(); frame.train_step(pair=True)
torch.cuda.synchronize()
cnt = R.last_counters(); R.set_capacity_policy("static", int(cnt["D"] * 1.25) + 4096)
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(25):
        zero(); frame.train_step(pair=True)
    torch.cuda.synchronize(); print("pair ms/step", round((time.perf_counter() - t0) / 25 * 1e3, 3), flush=True)
import gc
for label in ("gc on", "gc off"):
    if label == "gc off":
        gc.collect(); gc.disable()
    ts = []
    for rep in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(25):
            zero(); frame.train_step()
        torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) / 25 * 1e3, 3))
    print("two-call ms/step,", label, ts, flush=True)

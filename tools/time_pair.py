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
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(25):
        zero(); frame.train_step()
    torch.cuda.synchronize(); print("two-call ms/step", round((time.perf_counter() - t0) / 25 * 1e3, 3), flush=True)

"""cProfile of the host side of eager steps (where does the per-step enqueue time go?)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from d3ga_amd import rasterizer as R
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
f = bench.Frame(wl, torch.device("cuda", 0), 0)
for _ in range(3):
    for p in f.params.values(): p.grad = None
    f.step()
torch.cuda.synchronize()
R.set_capacity_policy("static", int(R.last_counters()["D"] * 1.25) + 4096)
def run(n):
    for _ in range(n):
        for p in f.params.values(): p.grad = None
        f.step()
run(20); torch.cuda.synchronize()
t0 = time.perf_counter(); run(200); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue ms/step", (t1 - t0) / 200 * 1e3, "wall ms/step", (t2 - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable(); run(200); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

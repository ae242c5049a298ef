"""cProfile of the host side of eager steps (where does the per-step enqueue time go?).
usage: python tools/prof_host.py [workload] [frame|train|pair]   -> stdout (tools/gpu_prof_host.sh keeps it under gpurun_out/)"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from d3ga_amd import rasterizer as R
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
mode = sys.argv[2] if len(sys.argv) > 2 else "frame"
f = bench.Frame(wl, torch.device("cuda", 0), 0)
R.set_accumulator_policy("persistent")
cycle = mode.endswith("_cycle")       # a new camera and target every step, as bench.py's eager training step does it
mode = mode.replace("_cycle", "")
step = {"frame": f.step, "train": f.train_step, "pair": lambda: f.train_step(pair=True)}[mode]
views = f.camera_cycle() if cycle else None
it = [0]
def run(n):
    for _ in range(n):
        if cycle:
            b, t = views[it[0] % len(views)]; it[0] += 1
            f.target.copy_(t, non_blocking=True); f.target_slot.set(t); f.slot.set(b)
        for p in f.params.values(): p.grad = None
        step()
run(3)
torch.cuda.synchronize()
R.set_capacity_policy("static", int(R.last_counters()["D"] * 1.25) + 4096)
import gc
run(20); torch.cuda.synchronize(); gc.collect(); gc.freeze()
t0 = time.perf_counter(); run(300); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"== {wl} {mode}: host enqueue ms/step {(t1 - t0) / 300 * 1e3:.4f}  wall ms/step {(t2 - t0) / 300 * 1e3:.4f}")
pr = cProfile.Profile(); pr.enable(); run(300); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
st.sort_stats("cumulative").print_stats(35)

"""Host-side cost of one EAGER step (no hipGraph): cProfile over N steps of bench.Frame.step at a small workload, where the
GPU is never the bottleneck.  Usage: python tools/host_profile.py [workload] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
f = bench.Frame(wl, torch.device("cuda", 0), 0)


def step():
    for p in f.params.values():
        p.grad = None
    f.step()


for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
t_host = time.perf_counter() - t0                     # enqueue time: the host runs ahead of the GPU
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{wl}: host enqueue {1e3 * t_host / n:.3f} ms/step, wall {1e3 * t_all / n:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

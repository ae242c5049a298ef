"""Loop statistics of composite_bwd_rows3_kernel on a workload (needs a library built with D3GA_DIAG=1)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("D3GA_COMPOSITE_VARIANT", "287")
import bench  # noqa: E402
from d3ga_amd import _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
L = ctypes.CDLL(_lib._PATH)
f = bench.Frame(wl, torch.device("cuda", 0), 0)
f.step()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
assert L.d3ga_diag_read(out, 1) == 0
for p in f.params.values():
    p.grad = None
f.step()
torch.cuda.synchronize()
assert L.d3ga_diag_read(out, 1) == 0
names = ["active_waves", "batches", "iterations", "hit_iterations", "hit_lanes", "hit_rows", "row_list_entries", "max_iterations_of_a_wave"]
d = dict(zip(names, [int(x) for x in out]))
print(d)
w = d["active_waves"]
print({k: round(v / w, 2) for k, v in d.items()}, "per active wave")
print("hit fraction of iterations", round(d["hit_iterations"] / d["iterations"], 3),
      "| lanes per hit iteration", round(d["hit_lanes"] / d["hit_iterations"], 2),
      "| rows per hit iteration", round(d["hit_rows"] / d["hit_iterations"], 2),
      "| row-list entries per iteration", round(d["row_list_entries"] / d["iterations"], 2),
      "| lanes per hit row", round(d["hit_lanes"] / d["hit_rows"], 2))

"""k cameras of the C3 pose through the view-batched rasterizer, eagerly, a few steps (for rocprofv3 passes: kernel trace / --pmc).
usage: python tools/batched_step.py [k] [workload] [steps]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from d3ga_amd import rasterizer as R
from d3ga_amd.raster_views import CameraBatch
from d3ga_amd.renderer import render_views
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wl = sys.argv[2] if len(sys.argv) > 2 else "C3"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
f = bench.Frame(wl, torch.device("cuda", 0), 0)
batches = [f.syn.make_batch(f.wl.width, f.wl.height, azimuth=2 * math.pi * v / max(8, k), camera_id=v, fill=f.fill) for v in range(k)]
W, H = int(batches[0]["width"]), int(batches[0]["height"])
cams = CameraBatch(k, W, H, device=f.dev).set(batches)
targets = torch.stack([torch.rand(3, H, W, generator=torch.Generator().manual_seed(100 + v)) for v in range(k)]).to(f.dev)
for i in range(steps + 2):
    for p in f.params.values():
        p.grad = None
    render_views(None, f.upstream(), f.bg, targets=targets, cameras=cams)["l1"].backward()
    if i == 1:
        R.set_capacity_policy("static", int(R.last_counters()["D"] * 1.25) + 4096)
torch.cuda.synchronize()
print("D", R.last_counters())

"""Training step (two renders / pair, optional networks) under stream capture -- run under a short timeout."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from d3ga_amd import rasterizer as R
mode = sys.argv[1]
kw = {"color_pair": dict(with_fields="color", pair=True), "fields": dict(with_fields=True), "pair": dict(pair=True), "plain": dict()}[mode]
dev = torch.device("cuda", 0)
frame = bench.Frame(sys.argv[2] if len(sys.argv) > 2 else "C1", dev, 0)
def zero():
    for q in list(frame.params.values()) + getattr(frame, "field_params", []) + [getattr(frame, n) for n in ("color_feat", "frame_enc") if hasattr(frame, n)]:
        q.grad = None
def step():
    frame.train_step(**kw)
for _ in range(3):
    zero(); step()
torch.cuda.synchronize(); print("eager ok", flush=True)
cnt = R.last_counters()
R.set_capacity_policy("static", int(cnt["D"] * 1.5) + 4096)
zero(); step(); torch.cuda.synchronize(); print("static ok", flush=True)
leaves = [q for q in list(frame.params.values()) + getattr(frame, "field_params", []) + [getattr(frame, n) for n in ("color_feat", "frame_enc") if hasattr(frame, n)] if q.grad is not None]
ref = [q.grad.clone() for q in leaves]
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        zero(); step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize(); print("side-stream warm-up ok", flush=True)
zero()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
worst = max(float((q.grad - r).abs().max() / (r.abs().max() + 1e-30)) for q, r in zip(leaves, ref))
print(mode, "replay ok; worst relative gradient difference vs eager", worst, flush=True)
import time
for _ in range(5): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); print("graph replay ms/step", round((time.perf_counter() - t0) / 50 * 1e3, 4), flush=True)
t0 = time.perf_counter()
for _ in range(50):
    zero(); step()
torch.cuda.synchronize(); print("eager ms/step", round((time.perf_counter() - t0) / 50 * 1e3, 4), flush=True)

"""probe: do external events recorded inside a captured hipGraph give per-node timestamps on replay?"""
import torch, inspect
print(torch.__version__)
print(inspect.signature(torch.cuda.Event.__new__) if hasattr(torch.cuda.Event, "__new__") else "")
dev = "cuda:0"
x = torch.randn(64 << 20, device=dev)
y = torch.empty_like(x)
try:
    evs = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(3)]
except TypeError as e:
    print("no external kw:", e); raise SystemExit(0)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y.copy_(x); y.mul_(2.0)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    evs[0].record()
    y.copy_(x)
    evs[1].record()
    for _ in range(4):
        y.mul_(1.0001)
    evs[2].record()
torch.cuda.synchronize()
for i in range(5):
    g.replay()
    torch.cuda.synchronize()
    try:
        print(i, "copy ms", evs[0].elapsed_time(evs[1]), "4 muls ms", evs[1].elapsed_time(evs[2]))
    except Exception as e:
        print("elapsed_time failed:", type(e).__name__, e)
        break
# reference: eager events
a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
a.record(); y.copy_(x); b.record()
for _ in range(4):
    y.mul_(1.0001)
c.record(); torch.cuda.synchronize()
print("eager: copy ms", a.elapsed_time(b), "4 muls ms", b.elapsed_time(c))

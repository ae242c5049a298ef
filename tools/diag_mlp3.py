import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd.mlp import linear_act
P = 500_000
x = torch.randn(P, 128, device="cuda"); w = torch.randn(128, 128, device="cuda") / 11; b = torch.randn(128, device="cuda")
with torch.no_grad():
    for _ in range(3): y = linear_act(x, w, b, 0.1)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): y = linear_act(x, w, b, 0.1)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        h = x
        for _ in range(10): h = linear_act(h, w, b, 0.1)
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("10 chained 128x128 layers in a hipGraph: per layer us", e0.elapsed_time(e1) / 100 * 1e3)

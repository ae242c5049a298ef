import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd.mlp import linear_act
P = 500_000
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    for K, N in ((128, 128), (128, 64), (128, 32), (64, 128), (32, 128), (32, 32), (64, 64), (96, 96)):
        x = torch.randn(P, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 11; b = torch.randn(N, device="cuda")
        ms = t(lambda: linear_act(x, w, b, 0.1))
        mfma_us = P / 32 * (K / 2) * (N / 32) * 64 / 1024 / 2.4e3
        mem_us = P * (K + N) * 4 / 4.7e6
        print(f"K={K:3d} N={N:3d}: {ms*1e3:7.1f} us   mfma-ideal {mfma_us:6.1f} us   mem-ideal {mem_us:6.1f} us")

import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd import _lib
from d3ga_amd._lib import dptr, stream_handle
P, M, N = 500_000, 16, 8
dev = "cuda"
means = torch.randn(P, 3, device=dev)
g = torch.randn(N, P + 1, 3, device=dev)
g[:, P] = torch.randn(N, 3, device=dev) * 3
out = torch.empty(P, M, 3, device=dev)
L = _lib.lib()
def run():
    assert L.d3ga_sh_grad_from_views(P, M, 3, N, dptr(means), dptr(g), 3 * (P + 1), dptr(g[0, P]), 3 * (P + 1), 1.0 / N, dptr(out), stream_handle()) == 0
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(f"sh_grad_from_views P={P} N={N}: {us:.1f} us  ({(P*(12*N+12+12*M))/us/1e3:.0f} GB/s)")

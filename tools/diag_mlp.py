import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd import _lib
from d3ga_amd._lib import dptr, stream_handle
from d3ga_amd.mlp import _panel
P = 500_000
x = torch.randn(P, 128, device="cuda"); w = torch.randn(128, 128, device="cuda") / 11; b = torch.randn(128, device="cuda")
y = torch.empty(P, 128, device="cuda")
L = _lib.lib()
panel = _panel(w, True)   # interleaved layout prepared by the host layer
def run(n_out, rows=P, slope=0.0):
    # n_out = 97..128 keeps NB = 4 but stores only columns < n_out
    return L.d3ga_mlp_linear(rows, 128, n_out, dptr(x), None, slope, None, dptr(panel), dptr(b), 0.1, dptr(y), stream_handle())
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("full 128 cols stored      ms", t(lambda: run(128)))
print("97 cols stored (NB=4)     ms", t(lambda: run(97)))
for rows in (131072, 262144, 500000):
    print(rows, "rows ms", t(lambda: run(128, rows)))
print("reads folded onto 512 rows (needs D3GA_DIAG build) ms", t(lambda: run(128, P, -12345.0)))

import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd import _lib
from d3ga_amd._lib import dptr, check, stream_handle
L = _lib.lib()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for P in (500000, 500001):
    dpre = torch.randn(P, 128, device="cuda"); x = torch.randn(P, 128, device="cuda")
    dw = torch.zeros(128, 128, device="cuda"); db = torch.zeros(128, device="cuda")
    us = t(lambda: check(L.d3ga_mlp_wgrad_acc(P, 128, 128, dptr(dpre), dptr(x), dptr(dw), dptr(db), stream_handle()), "wg"))
    print(f"P={P}: wgrad 128x128 {us:.1f} us ({'no MFMA phase' if P & 1 else 'full'})")

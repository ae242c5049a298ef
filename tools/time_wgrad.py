"""Time d3ga_mlp_wgrad_acc over the layer shapes of the fields at P rows (HIP events, 30 launches) and check each against torch f64.
usage: python tools/time_wgrad.py [P] ; D3GA_KNOBS selects variants (wgrad_ws=0: the barrier-phased kernel; wgrad_ws=2: the
wavefront-specialised kernel for the narrow end layers as well)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd import _lib
from d3ga_amd._lib import check, dptr, stream_handle
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
dev = "cuda"
L = _lib.lib()
for N, K in ((128, 128), (128, 11), (11, 128), (128, 80), (4, 128), (128, 48), (3, 128)):
    g = torch.Generator().manual_seed(1)
    dpre = torch.randn(P, N, generator=g).to(dev); X = torch.randn(P, K, generator=g).to(dev)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    run = lambda: check(L.d3ga_mlp_wgrad_acc(P, N, K, dptr(dpre), dptr(X), dptr(dw), dptr(db), stream_handle()), "wgrad")
    run(); torch.cuda.synchronize()
    ref = dpre.double().t() @ X.double()
    err = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
    errb = ((db.double() - dpre.double().sum(0)).abs().max() / dpre.double().sum(0).abs().max()).item()
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print(f"D3GA_KNOBS={os.environ.get('D3GA_KNOBS','(defaults)')} P={P} N={N} K={K}: {ms*1e3:.1f} us  {4*P*(N+K)/ms/1e6:.0f} GB/s  rel err dW {err:.2e} db {errb:.2e}")

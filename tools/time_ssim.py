import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd.losses import ssim, l1_loss
a = torch.rand(3, 1080, 1920, device="cuda").requires_grad_(True)
b = torch.rand(3, 1080, 1920, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    print("ssim fwd only (no grad maps) us", t(lambda: ssim(a, b)))
def fb():
    a.grad = None
    ssim(a, b).backward()
print("ssim fwd+bwd us", t(fb))
def l1():
    a.grad = None
    l1_loss(a, b).backward()
print("l1 fwd+bwd us", t(l1))

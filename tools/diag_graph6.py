"""Which piece of the training step does not survive stream capture?  (run each mode under a short timeout)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
dev = torch.device("cuda", 0)
from d3ga_amd.losses import l1_loss, l1_ssim, ssim
g = torch.Generator().manual_seed(0)
a = torch.rand(3, 270, 480, generator=g).to(dev).requires_grad_(True)
b = torch.rand(3, 270, 480, generator=g).to(dev)
def step():
    a.grad = None
    if mode == "l1":
        l1_loss(a, b).backward()
    elif mode == "ssim":
        (1 - ssim(a, b)).backward()
    elif mode == "l1_ssim":
        l, s = l1_ssim(a, b); (0.8 * l + 0.2 * (1 - s)).backward()
    elif mode == "mix":
        l, s = l1_ssim(a, b); (0.8 * l + 0.2 * (1 - s) + l1_loss(a * 0.5, b)).backward()
step(); torch.cuda.synchronize()
ref = a.grad.clone()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step(); step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print("warm-up on side stream ok", flush=True)
a.grad = None
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    step()
print("captured", flush=True)
gr.replay(); torch.cuda.synchronize()
print(mode, "replay ok, max diff", float((a.grad - ref).abs().max()), flush=True)

"""ColorField at 500k Gaussians: where the time goes (direction encoding in ATen vs the MFMA trunk)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd.mlp import ColorField, sh4_direction_encoding, view_directions
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
dev = "cuda"
col = ColorField().to(dev)
g = torch.Generator().manual_seed(0)
feat = (0.33 * torch.rand(P, 64, generator=g)).to(dev).requires_grad_(True)
means = torch.randn(P, 3, generator=g).to(dev).requires_grad_(True)
pose = torch.randn(98, generator=g).to(dev)
frame = torch.randn(32, generator=g).to(dev).requires_grad_(True)
cam = torch.tensor([[0.0, 0.0, 4.0]], device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def dirs():
    return view_directions(means, cam)
def enc_fb():
    means.grad = None
    sh4_direction_encoding(dirs()).sum().backward()
def full_fb():
    for p in list(col.parameters()) + [feat, means, frame]: p.grad = None
    rgb, op = col(feat, pose, dirs(), frame_encoding=frame)
    (rgb.sum() + op.sum()).backward()
vd = dirs().detach()
enc = sh4_direction_encoding(vd).detach().requires_grad_(True)
col_id = ColorField(direction_encoding=lambda e: e).to(dev)
def trunk_fb():
    for p in list(col_id.parameters()) + [feat, frame]: p.grad = None
    enc.grad = None
    rgb, op = col_id(feat, pose, enc, frame_encoding=frame)
    (rgb.sum() + op.sum()).backward()
with torch.no_grad():
    print(f"P={P}  fwd only: dirs+encoding {t(lambda: sh4_direction_encoding(dirs())):.3f} ms, full {t(lambda: col(feat, pose, dirs(), frame_encoding=frame)):.3f} ms")
print(f"fwd+bwd: dirs+encoding {t(enc_fb):.3f} ms, trunk {t(trunk_fb):.3f} ms, full {t(full_fb):.3f} ms")

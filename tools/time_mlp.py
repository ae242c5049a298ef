import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd.mlp import CanonicalField, linear_act
from oracle import mlp as om
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
dev = "cuda"
cf = CanonicalField().to(dev)
g = torch.Generator().manual_seed(0)
barys = torch.rand(P, 4, generator=g).to(dev).requires_grad_(True)
rots = torch.randn(P, 4, generator=g).to(dev).requires_grad_(True)
scales = torch.randn(P, 3, generator=g).to(dev).requires_grad_(True)
pose = torch.randn(98, generator=g).to(dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def fwd():
    with torch.no_grad():
        return cf(barys, rots, scales, pose)
def fb():
    for p in list(cf.parameters()) + [barys, rots, scales]: p.grad = None
    o = cf(barys, rots, scales, pose)
    (o[0].sum() + o[1].sum() + o[2].sum()).backward()
hidden = [(l.weight, l.bias) for l in cf.network]
def ref_fwd():
    with torch.no_grad():
        return om.canonical_field(barys, rots, scales, pose, hidden, cf.output.weight, cf.output.bias)
def ref_fb():
    for p in list(cf.parameters()) + [barys, rots, scales]: p.grad = None
    o = om.canonical_field(barys, rots, scales, pose, hidden, cf.output.weight, cf.output.bias)
    (o[0].sum() + o[1].sum() + o[2].sum()).backward()
flops_fwd = 2.0 * P * (11 * 128 + 3 * 128 * 128 + 128 * 11)
print(f"P={P}")
a, b = t(fwd), t(ref_fwd)
print(f"fwd   fused {a:.3f} ms ({flops_fwd / a / 1e9:.1f} TFLOP/s)   torch-ATen same GPU {b:.3f} ms")
a, b = t(fb), t(ref_fb)
print(f"fwd+bwd fused {a:.3f} ms ({3 * flops_fwd / a / 1e9:.1f} TFLOP/s)   torch-ATen same GPU {b:.3f} ms")
x = torch.randn(P, 128, device=dev); w = torch.randn(128, 128, device=dev) / 11; bb = torch.randn(128, device=dev)
with torch.no_grad():
    a = t(lambda: linear_act(x, w, bb, 0.1)); b = t(lambda: torch.nn.functional.leaky_relu(torch.nn.functional.linear(x, w, bb), 0.1))
print(f"one 128x128 layer: fused {a:.3f} ms ({2.0 * P * 128 * 128 / a / 1e9:.1f} TFLOP/s, {2 * P * 512 / a / 1e6:.0f} GB/s)   torch {b:.3f} ms")

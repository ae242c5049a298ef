import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd.mlp import linear_act
P, K, N = 4096, 128, 128
for spread in (0.0, 2.0, 4.0, 8.0):
    g = torch.Generator().manual_seed(int(spread) + 5)
    mag = lambda *shape: 10.0 ** (spread * (torch.rand(*shape, generator=g) - 0.5))
    x = (torch.randn(P, K, generator=g) * mag(P, K)).cuda(); w = (torch.randn(N, K, generator=g) * mag(N, K) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    with torch.no_grad():
        y = linear_act(x, w, b, 1.0).double(); ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        y32 = torch.nn.functional.linear(x, w, b).double()
        # a strict left-to-right fmaf chain in f32, emulated: cumulative sum in float32
        chain = (x[:256, None, :] * w[None, :, :]).cumsum(-1)[..., -1].double() + b.double()
        bound = (x.double().abs() @ w.double().abs().t() + b.double().abs()) * 2.0 ** -24
    e, e32, ec = (y - ref).abs() / bound, (y32 - ref).abs() / bound, (chain - ref[:256]).abs() / bound[:256]
    print(f"spread {spread}: mine max {e.max():.2f} mean {e.mean():.3f} | ATen f32 max {e32.max():.2f} mean {e32.mean():.3f} | serial f32 sum max {ec.max():.2f} mean {ec.mean():.3f}")

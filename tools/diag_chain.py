import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd.mlp import mlp_chain
def run(seed, force_bias=False, P_override=None):
    rng = np.random.default_rng(5000 + seed)
    P = int(rng.choice([1, 7, 31, 32, 33, 64, 100, 511, 513, int(rng.integers(1, 3000))]))
    L = int(rng.integers(1, 6))
    pick = lambda: int(rng.choice([1, 3, 4, 11, 16, 17, 32, 33, 48, 64, 80, 96, 127, 128, int(rng.integers(1, 129))]))
    widths = [pick() for _ in range(L + 1)]
    slopes = [float(rng.choice([0.1, 1.0, 0.01])) for _ in range(L)]
    if P_override: P = P_override
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(P, widths[0], generator=g)
    layers = [(torch.randn(b, a, generator=g) / a ** 0.5, torch.randn(b, generator=g) if (rng.random() < 0.8 or force_bias) else None)
              for a, b in zip(widths[:-1], widths[1:])]
    up = torch.randn(P, widths[-1], generator=g)
    xr = x.double().requires_grad_(True)
    lr = [(w.double().requires_grad_(True), None if b is None else b.double().requires_grad_(True)) for w, b in layers]
    h = xr
    hs = []
    for (w, b), sl in zip(lr, slopes):
        h = torch.nn.functional.leaky_relu(torch.nn.functional.linear(h, w, b), sl); hs.append(h)
    h.backward(up.double())
    xd = x.cuda().requires_grad_(True)
    ld = [(w.cuda().requires_grad_(True), None if b is None else b.cuda().requires_grad_(True)) for w, b in layers]
    y = mlp_chain(xd, ld, slopes); y.backward(up.cuda())
    d = (xd.grad.cpu().double() - xr.grad).abs().max(1).values / xr.grad.abs().max()
    bad = torch.nonzero(d > 1e-4).flatten().tolist()
    # how close to zero are the pre-activations of the bad rows?
    near = [float(hh.detach().abs().min()) for hh in hs]
    print(seed, P, widths, slopes, "bias:", [b is not None for _, b in layers], "bad dx rows:", len(bad), bad[:12], "min|h| per layer", ["%.1e" % v for v in near], flush=True)
    if bad:
        r = bad[0]
        for i, hh in enumerate(hs[:-1]):
            row = hh.detach()[r]
            j = int(row.abs().argmin()); print("   layer", i, "row", r, "smallest |h| =", float(row[j]), "at col", j)
run(1068)
run(1068, force_bias=True)
run(1068, P_override=512)

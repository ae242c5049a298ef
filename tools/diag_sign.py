import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3ga_amd.mlp import _linear, _panel
for P, K, N in [(64, 128, 128), (300, 128, 128), (1000, 128, 11), (512, 32, 32)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(P, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    y, sign = _linear(x, _panel(w, True), b, 0.1, N, want_sign=True)
    torch.cuda.synchronize()
    bits = ((sign.cpu().numpy().view(np.uint32)[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(P, -1)[:, :N].astype(bool)
    ref = (y > 0).cpu().numpy()
    bad = np.nonzero((bits != ref).any(1))[0]
    print(P, K, N, "bad rows:", len(bad), bad[:40], "zero rows among bad:", int((~bits[bad].any(1)).sum()))

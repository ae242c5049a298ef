import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import test_gpu_views as T
for seed in [int(x) for x in sys.argv[1:]]:
    tag, from_sr, per_view, geo, k, seq, bat = T._fuzz_case(seed)
    _, r0 = seq()
    worst_self, worst_b = {}, {}
    for _ in range(8):
        _, r = seq()
        _, m = bat()
        for n in r0:
            views = range(k) if (per_view and n in geo) else (None,)
            for v in views:
                a = r0[n] if v is None else r0[n][v]
                sc = float(a.abs().max()) or 1.0
                x = r[n] if v is None else r[n][v]
                y = m[n] if v is None else m[n][v]
                worst_self[n] = max(worst_self.get(n, 0), float((a - x).abs().max()) / sc)
                worst_b[n] = max(worst_b.get(n, 0), float((a - y).abs().max()) / sc)
    print(tag, "\n  self   ", {n: f"{v:.1e}" for n, v in worst_self.items()}, "\n  batched", {n: f"{v:.1e}" for n, v in worst_b.items()}, flush=True)

"""Run-to-run spread of the rasterizer's gradients (float atomics across tiles land in another order every run): the SINGLE-VIEW
operator against itself on the scene of tests/test_gpu_views.py, n runs -> max |g_run - g_first| / max |g_first| per leaf.
Basis of the bars in tests/test_gpu_views.py (same-arithmetic comparisons of sums formed in another order)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_views as T          # noqa: E402
from util import scene_inputs       # noqa: E402


def main(n=30):
    for use_sh, from_sr, scale_mult in ((True, True, 3.0), (True, False, 3.0), (False, True, 3.0)):
        inp = scene_inputs("T1", scale_mult=scale_mult)
        batch = T._batches(inp, 5, fov_jitter=True)[1]
        bg = torch.tensor([0.3, 0.6, 0.1], device="cuda")
        gpix = torch.randn(3, inp["H"], inp["W"], generator=torch.Generator().manual_seed(3)).cuda()
        first, worst = None, {}
        for _ in range(n):
            leaves = T._leaves(inp, use_sh, from_sr)
            img, _ = T._single_view(inp, batch, leaves, bg, use_sh, from_sr)
            (img * gpix).sum().backward()
            g = {k: v.grad.clone() for k, v in leaves.items()}
            if first is None:
                first = g
                continue
            for k in g:
                s = float(first[k].abs().max())
                worst[k] = max(worst.get(k, 0.0), float((g[k] - first[k]).abs().max()) / s)
        print(f"use_sh={use_sh} from_sr={from_sr}: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()), flush=True)


def frame_spread(n=300, workload="C1"):
    """the whole bench step (LBS + cage deform + render + fused L1 + backward) against itself: what the eager-against-replay and
    sharded-against-sequential tests of tests/test_gpu_view_sharded.py compare (bars 1e-5 .. 1e-4 of the largest element)"""
    import bench
    frame = bench.Frame(workload, torch.device("cuda"), view_index=0)
    first, worst = None, {}
    for _ in range(n):
        for p in frame.params.values():
            p.grad = None
        frame.step()
        g = {k: p.grad.clone() for k, p in frame.params.items()}
        if first is None:
            first = g
            continue
        for k in g:
            worst[k] = max(worst.get(k, 0.0), float((g[k] - first[k]).abs().max()) / float(first[k].abs().max()))
    print(f"frame {workload} x{n}: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "frame":
        frame_spread(int(sys.argv[2]) if len(sys.argv) > 2 else 300, sys.argv[3] if len(sys.argv) > 3 else "C1")
    else:
        main(int(sys.argv[1]) if len(sys.argv) > 1 else 30)

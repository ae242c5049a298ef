"""Per-wave timeline of the two-launch forward (tile_cull_kernel, composite_fwd_lists_kernel) on a workload: library built with
D3GA_DIAG=timeline (tools/_build/libd3ga_hip_timeline.so), selected with D3GA_LIB_PATH."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from d3ga_amd import _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
L = ctypes.CDLL(_lib._PATH)
f = bench.Frame(wl, torch.device("cuda", 0), 0)
for _ in range(30):
    for p in f.params.values():
        p.grad = None
    f.step()
torch.cuda.synchronize()
assert L.d3ga_diag_lists_read(None, 0, None, 0, 1) == 0
for p in f.params.values():
    p.grad = None
f.step()
torch.cuda.synchronize()
nb, nc = 32768, 16384
bb, cb = (ctypes.c_ulonglong * (4 * nb))(), (ctypes.c_ulonglong * (4 * nc))()
assert L.d3ga_diag_lists_read(bb, nb, cb, nc, 1) == 0


def timeline(a, name, unit):
    a = a[a[:, 1] != 0]
    t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    base = t0.min(); t0 -= base; t1 -= base
    dur = t1 - t0
    print(f"== {name}: {len(a)} {unit}, span {t1.max() / 100.0:.1f} us; duration us p50/p90/p99/max", [round(float(np.percentile(dur, q)) / 100.0, 1) for q in (50, 90, 99, 100)],
          "| start us p50/p90/max", [round(float(np.percentile(t0, q)) / 100.0, 1) for q in (50, 90, 100)])
    for q in (0.1, 0.3, 0.5, 0.7, 0.8, 0.9, 0.95):
        t = q * t1.max()
        print(f"   resident at {q:.2f} of the span: {int(((t0 <= t) & (t1 > t)).sum())}")
    return a, t0, t1, dur


a, t0, t1, dur = timeline(np.array(bb, dtype=np.uint64).reshape(-1, 4), "composite_fwd_lists_kernel", "wavefronts with work")
groups = (a[:, 2] & np.uint64(0xffff)).astype(np.int64); nmax = ((a[:, 2] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64); ll = (a[:, 2] >> np.uint64(32)).astype(np.int64)
stage = ((a[:, 3] >> np.uint64(40)) & np.uint64(0xfffff)).astype(np.int64)
hwid = (a[:, 3] & np.uint64(0xffff)).astype(np.int64); xcc = ((a[:, 3] >> np.uint64(32)) & np.uint64(15)).astype(np.int64)
print("groups per wave mean/p90/max", round(float(groups.mean()), 2), int(np.percentile(groups, 90)), int(groups.max()), "| sum", int(groups.sum()),
      "| us per group: all", round(float(dur.sum() / max(groups.sum(), 1)) / 100.0, 3), "heaviest decile",
      round(float(dur[groups >= np.percentile(groups, 90)].sum() / max(groups[groups >= np.percentile(groups, 90)].sum(), 1)) / 100.0, 3),
      "| share of the lifetime waiting for / staging records: median", round(float(np.median(stage / np.maximum(dur, 1))), 3),
      "heaviest decile", round(float(np.median((stage / np.maximum(dur, 1))[groups >= np.percentile(groups, 90)])), 3))
order = np.argsort(-t1)[:8]
print("last to end (end us, start us, groups, longest row list, tile list):", [(round(t1[i] / 100.0, 1), round(t0[i] / 100.0, 1), int(groups[i]), int(nmax[i]), int(ll[i])) for i in order])
simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
uk, inv = np.unique(key, return_inverse=True)
n_w = np.bincount(inv); g_sum = np.bincount(inv, weights=groups); last = np.zeros(len(uk)); np.maximum.at(last, inv, t1)
print("SIMDs seen", len(uk), "| waves per SIMD min/med/max", int(n_w.min()), float(np.median(n_w)), int(n_w.max()), "| groups per SIMD p10/p50/p90/max",
      [int(np.percentile(g_sum, q)) for q in (10, 50, 90, 100)], "| SIMD finish us p10/p50/p90/max", [round(float(np.percentile(last, q)) / 100.0, 1) for q in (10, 50, 90, 100)],
      "| corr(finish, groups)", round(float(np.corrcoef(last, g_sum)[0, 1]), 3))

c, c0, c1, cdur = timeline(np.array(cb, dtype=np.uint64).reshape(-1, 4), "tile_cull_kernel", "workgroups with work")
clen = c[:, 2].astype(np.int64)
print("tile list length mean/p90/max", round(float(clen.mean()), 1), int(np.percentile(clen, 90)), int(clen.max()), "| us per 256 entries: all",
      round(float(cdur.sum() / max((clen / 256.0).sum(), 1)) / 100.0, 3), "| longest lists:", [(int(clen[i]), round(cdur[i] / 100.0, 1), round(c0[i] / 100.0, 1)) for i in np.argsort(-clen)[:6]])
order = np.argsort(-c1)[:6]
print("last to end (end us, start us, length):", [(round(c1[i] / 100.0, 1), round(c0[i] / 100.0, 1), int(clen[i])) for i in order])

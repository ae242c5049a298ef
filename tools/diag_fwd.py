"""Loop statistics and per-wave timeline of composite_fwd_q_kernel on a workload (library built with D3GA_DIAG=counters or
D3GA_DIAG=timeline, selected with D3GA_LIB_PATH)."""
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
out = (ctypes.c_ulonglong * 8)()
assert L.d3ga_diag_fwd_read(out, None, 0, 1) == 0
for p in f.params.values():
    p.grad = None
f.step()
torch.cuda.synchronize()
n = 32768
buf = (ctypes.c_ulonglong * (4 * n))()
assert L.d3ga_diag_fwd_read(out, buf, n, 1) == 0
a = np.array(buf, dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 1] != 0]                      # (records are indexed by workgroup, no atomics: inactive waves stay zero)
w = n = len(a)
iters = (a[:, 2] & np.uint64(0xffffffff)).astype(np.int64); batches = (a[:, 2] >> np.uint64(32)).astype(np.int64); ll = (a[:, 3] & np.uint64(0xffffffff)).astype(np.int64); blend_ticks = (a[:, 3] >> np.uint64(32)).astype(np.int64)
out[2], out[3] = int(batches.sum()), int(iters.sum())
print({"active_waves": w, "stage1_chunks": int(out[1]), "batches": int(out[2]), "blend_iterations": int(out[3]),
       "entry_block_pairs": int(out[4]), "blended_pixel_pairs": int(out[5]), "stage1_survivors": int(out[6]), "list_entries_scanned": int(out[7])})
if int(out[3]):
    it = int(out[3])
    print("lane efficiency of the blend loop: blended (entry, pixel) pairs / (iterations x 2 x 64) =", round(int(out[5]) / (it * 128.0), 4),
          "| row-slot efficiency: (entry, block) pairs / (iterations x 2 x 4) =", round(int(out[4]) / (it * 8.0), 4) if int(out[4]) else None)
t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
b = t0.min(); t0 -= b; t1 -= b
dur = t1 - t0
print("span (10 ns ticks)", int(t1.max()), "| iterations per wave: mean", round(float(iters.mean()), 1), "max", int(iters.max()),
      "p50/p90/p99", [int(np.percentile(iters, q)) for q in (50, 90, 99)])
print("start percentiles", [int(np.percentile(t0, q)) for q in (0, 25, 50, 75, 90, 100)], "end percentiles", [int(np.percentile(t1, q)) for q in (0, 25, 50, 75, 90, 100)])
print("wave duration (ticks): p50/p90/p99/max", [int(np.percentile(dur, q)) for q in (50, 90, 99, 100)])
print("share of a wave's lifetime inside the blend loop: median", round(float(np.median(blend_ticks / np.maximum(dur, 1))), 3),
      "| heaviest decile", round(float(np.median((blend_ticks / np.maximum(dur, 1))[dur >= np.percentile(dur, 90)])), 3),
      "| ticks per iteration inside the loop (median, heaviest decile)", round(float(np.median(blend_ticks / np.maximum(iters, 1))), 2),
      round(float(np.median((blend_ticks / np.maximum(iters, 1))[dur >= np.percentile(dur, 90)])), 2),
      "| ticks per batch outside the loop (median, heaviest decile)", round(float(np.median((dur - blend_ticks) / np.maximum(batches, 1))), 1),
      round(float(np.median(((dur - blend_ticks) / np.maximum(batches, 1))[dur >= np.percentile(dur, 90)])), 1))
order = np.argsort(-dur)[:6]
print("longest waves (dur, start, iterations, batches, list length):", [(int(dur[i]), int(t0[i]), int(iters[i]), int(batches[i]), int(ll[i])) for i in order])
order = np.argsort(-t1)[:6]
print("last waves to end (end, start, iterations, list length):", [(int(t1[i]), int(t0[i]), int(iters[i]), int(ll[i])) for i in order])
for q in (0.1, 0.3, 0.5, 0.7, 0.8, 0.9, 0.95):
    t = q * t1.max()
    print(f"resident active waves at {q:.2f} of the span:", int(((t0 <= t) & (t1 > t)).sum()))
# ticks per iteration as a function of concurrency: crude fit dur ~ a * iters + b * batches
A = np.stack([iters, batches, np.ones_like(iters)], 1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, dur.astype(np.float64), rcond=None)
print("least squares: ticks per iteration", round(coef[0], 3), "| per batch", round(coef[1], 2), "| constant", round(coef[2], 1))

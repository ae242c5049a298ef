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
iters = (a[:, 2] & np.uint64(0xffffffff)).astype(np.int64); batches = (a[:, 2] >> np.uint64(32)).astype(np.int64); ll = (a[:, 3] & np.uint64(0xfffff)).astype(np.int64); blend_ticks = ((a[:, 3] >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.int64)
hwid = ((a[:, 3] >> np.uint64(40)) & np.uint64(0xffff)).astype(np.int64); xcc = (a[:, 3] >> np.uint64(56)).astype(np.int64)
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
# per SIMD: how evenly is the work dealt, and when does each SIMD finish
simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
uk, inv = np.unique(key, return_inverse=True)
n_w = np.bincount(inv); it_sum = np.bincount(inv, weights=iters); last = np.zeros(len(uk)); np.maximum.at(last, inv, t1); big = np.zeros(len(uk)); np.maximum.at(big, inv, iters)
print("SIMDs seen", len(uk), "| waves per SIMD min/med/max", int(n_w.min()), float(np.median(n_w)), int(n_w.max()),
      "| iterations per SIMD p10/p50/p90/max", [int(np.percentile(it_sum, q)) for q in (10, 50, 90, 100)],
      "| SIMD finish ticks p10/p50/p90/max", [int(np.percentile(last, q)) for q in (10, 50, 90, 100)])
print("correlation of a SIMD's finish time with: its iteration sum", round(float(np.corrcoef(last, it_sum)[0, 1]), 3),
      "| its heaviest wave", round(float(np.corrcoef(last, big)[0, 1]), 3), "| its wave count", round(float(np.corrcoef(last, n_w)[0, 1]), 3))
# ticks per iteration as a function of concurrency: crude fit dur ~ a * iters + b * batches
A = np.stack([iters, batches, np.ones_like(iters)], 1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, dur.astype(np.float64), rcond=None)
print("least squares: ticks per iteration", round(coef[0], 3), "| per batch", round(coef[1], 2), "| constant", round(coef[2], 1))

if len(sys.argv) > 2:
    import hashlib
    import json
    it = int(out[3])
    summary = {"workload": wl, "kernel": "composite_fwd_q_kernel", "lib_sha256": hashlib.sha256(open(_lib._PATH, "rb").read()).hexdigest(),
               "active_waves": w, "batches": int(out[2]), "blend_iterations": it, "span_us": round(t1.max() / 100.0, 2),
               "wave_duration_us_p50_p90_max": [round(float(np.percentile(dur, q)) / 100.0, 2) for q in (50, 90, 100)],
               "iterations_per_wave_mean_p90_max": [round(float(iters.mean()), 1), int(np.percentile(iters, 90)), int(iters.max())],
               "share_of_wave_lifetime_in_blend_loop": round(float(np.median(blend_ticks / np.maximum(dur, 1))), 3),
               "resident_waves_at_fraction_of_span": {str(q): int(((t0 <= q * t1.max()) & (t1 > q * t1.max())).sum()) for q in (0.1, 0.3, 0.5, 0.7, 0.8, 0.9, 0.95)}}
    if int(out[5]):
        summary.update({"blended_entry_pixel_pairs": int(out[5]), "entry_block_pairs": int(out[4]), "lane_slots": it * 128,
                        "lane_efficiency": round(int(out[5]) / (it * 128.0), 4), "row_slot_efficiency": round(int(out[4]) / (it * 8.0), 4),
                        "stage1_survivors": int(out[6]), "list_entries_scanned": int(out[7])})
    json.dump(summary, open(sys.argv[2], "w"), indent=1)

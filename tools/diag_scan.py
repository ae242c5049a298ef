"""Loop statistics of composite_bwd_tile_kernel on a workload (needs a library built with D3GA_DIAG=counters)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from d3ga_amd import _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
L = ctypes.CDLL(_lib._PATH)
f = bench.Frame(wl, torch.device("cuda", 0), 0)
for _ in range(30):                      # clocks up, allocator warm
    for p in f.params.values():
        p.grad = None
    f.step()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
assert L.d3ga_diag_scan_read(out, 1) == 0
for p in f.params.values():
    p.grad = None
from d3ga_amd import rasterizer as R
R.stage_timer.enabled = True
R.stage_timer.reset()
f.step()
torch.cuda.synchronize()
print("stage times of the measured step (ms):", {k: round(v[1], 4) for k, v in R.stage_timer.summary().items()})
R.stage_timer.enabled = False
assert L.d3ga_diag_scan_read(out, 0) == 0
import numpy as np
NW = 32768
buf = (ctypes.c_ulonglong * (4 * NW))()
assert L.d3ga_diag_scan_waves(buf, NW) == 0
a = np.array(buf, dtype=np.uint64).reshape(NW, 4)
a = a[a[:, 1] != 0]                      # (records are indexed by workgroup and wave, no atomics: inactive waves stay zero)
n = w = len(a)
if n:      # tile-level merge kernel (rounds 3+)
    t0 = (a[:, 0] & np.uint64((1 << 40) - 1)).astype(np.int64); t1 = (a[:, 1] & np.uint64((1 << 40) - 1)).astype(np.int64)
    g = (a[:, 2] & np.uint64(0xffff)).astype(np.int64); trips = ((a[:, 2] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64)
    wave_groups = int(g.sum())
    print({"active_waves": w, "installs": int(out[3]), "hits": int(out[4]), "evictions": int(out[5]), "lock_loop_trips": int(trips.sum()),
           "wave_groups": wave_groups, "trips_per_group": round(int(trips.sum()) / max(wave_groups, 1), 3),
           "lines_to_hbm": int(out[3]) + int(out[5])})
    summary = {"workload": wl, "kernel": "composite_bwd_tile_kernel", "lib_sha256": __import__("hashlib").sha256(open(_lib._PATH, "rb").read()).hexdigest(),
               "active_waves": w, "wave_groups": wave_groups, "lock_loop_trips_per_group": round(int(trips.sum()) / max(wave_groups, 1), 3)}
    if int(out[8]):
        summary.update({"valid_entry_pixel_pairs": int(out[8]), "entry_block_pairs": int(out[9]), "row_groups": int(out[10]),
                        "lane_slots": wave_groups * 1024, "lane_efficiency": round(int(out[8]) / (wave_groups * 1024.0), 4),
                        "merge_cache": {"installs": int(out[3]), "hits": int(out[4]), "evictions": int(out[5]), "lines_to_hbm": int(out[3]) + int(out[5])}})
    if int(out[8]):      # D3GA_DIAG=counters: lane efficiency = valid (entry, pixel) pairs / issued lane slots (wave groups x 64 lanes x 16 pixel steps)
        print({"valid_pixel_pairs": int(out[8]), "entry_block_pairs": int(out[9]), "row_groups": int(out[10]),
               "lane_efficiency": round(int(out[8]) / (wave_groups * 1024.0), 4),
               "valid_pixels_per_entry_block_pair": round(int(out[8]) / max(int(out[9]), 1), 3),
               "row_balance(4*wave_groups/row_groups)": round(4.0 * wave_groups / max(int(out[10]), 1), 3)})
    b = t0.min(); t0 -= b; t1 -= b
    dur = t1 - t0
    print("span (10 ns ticks)", int(t1.max()), "| active waves", n, "| groups: mean", round(float(g.mean()), 2), "max", int(g.max()),
          "p50/p90/p99", [int(np.percentile(g, q)) for q in (50, 90, 99)])
    print("start percentiles", [int(np.percentile(t0, q)) for q in (0, 25, 50, 75, 90, 100)], "end percentiles", [int(np.percentile(t1, q)) for q in (0, 25, 50, 75, 90, 100)])
    print("wave duration (ticks): p50/p90/p99/max", [int(np.percentile(dur, q)) for q in (50, 90, 99, 100)])
    print("ticks per group: median", float(np.median(dur / g)), "p10", float(np.percentile(dur / g, 10)), "p90", float(np.percentile(dur / g, 90)))
    early = t0 < 0.05 * t1.max()
    print("waves that start in the first 5 % of the span:", int(early.sum()), "| their ticks per group by groups-of-the-wave quartile:",
          [round(float(np.median((dur / g)[early & (g >= lo) & (g <= hi)])), 1) if (early & (g >= lo) & (g <= hi)).any() else None
           for lo, hi in ((1, 4), (5, 8), (9, 12), (13, 99))])
    order = np.argsort(-t1)[:6]
    print("last waves to end (end, start, groups, ticks/group):", [(int(t1[i]), int(t0[i]), int(g[i]), round(float(dur[i] / g[i]), 1)) for i in order])
    order = np.argsort(-g)[:6]
    print("heaviest waves (groups, start, end):", [(int(g[i]), int(t0[i]), int(t1[i])) for i in order])
    res = {}
    for q in (0.1, 0.3, 0.5, 0.7, 0.8, 0.9, 0.95):
        t = q * t1.max()
        res[str(q)] = int(((t0 <= t) & (t1 > t)).sum())
        print(f"resident active waves at {q:.2f} of the span:", res[str(q)])
    summary.update({"span_us": round(t1.max() / 100.0, 2), "wave_duration_us_p50_p90_max": [round(float(np.percentile(dur, q)) / 100.0, 2) for q in (50, 90, 100)],
                    "wave_start_us_p50_p90": [round(float(np.percentile(t0, q)) / 100.0, 2) for q in (50, 90)],
                    "groups_per_wave_mean_p90_max": [round(float(g.mean()), 2), int(np.percentile(g, 90)), int(g.max())],
                    "resident_waves_at_fraction_of_span": res})
    if len(sys.argv) > 2:
        import json
        json.dump(summary, open(sys.argv[2], "w"), indent=1)
    sys.exit(0)
print('flushed entries with a gradient:', int(out[1]), '| of them also present in a lower row of the same flush:', int(out[2]))
print("active waves", w)
# per-wave timeline (s_memtime ticks): when do waves start / end, how long does a group take, how busy is each SIMD
import numpy as np
n = min(w, 32768)
buf = (ctypes.c_ulonglong * (4 * n))()
assert L.d3ga_diag_scan_waves(buf, n) == 0
a = np.array(buf, dtype=np.uint64).reshape(n, 4)
memtime_dur = (a[:, 0] >> np.uint64(40)).astype(np.int64)
t0, t1, hw = (a[:, 0] & np.uint64((1 << 40) - 1)).astype(np.int64), (a[:, 1] & np.uint64((1 << 40) - 1)).astype(np.int64), a[:, 3]
g = (a[:, 2] & np.uint64(0xffff)).astype(np.int64); rowg = ((a[:, 2] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64); ents = (a[:, 2] >> np.uint64(32)).astype(np.int64)
print({'wave_groups': int(g.sum()), 'row_groups': int(rowg.sum()), 'list_entries': int(ents.sum()), 'max_groups_of_a_wave': int(g.max()),
       'row_balance': round(4 * g.sum() / max(rowg.sum(), 1), 3)})
print("s_memtime ticks per 100 MHz tick (median over waves):", float(np.median(memtime_dur / np.maximum(t1 - t0, 1))))
b = t0.min()                             # s_memrealtime: one 100 MHz counter for the whole device
t0 -= b; t1 -= b
dur = t1 - t0
print("kernel span (10 ns ticks)", int(t1.max()), "| waves", n)
print("start time percentiles", [int(np.percentile(t0, q)) for q in (0, 25, 50, 75, 90, 100)])
print("end   time percentiles", [int(np.percentile(t1, q)) for q in (0, 25, 50, 75, 90, 100)])
print("ticks per group: median", float(np.median(dur / np.maximum(g, 1))), "p10", float(np.percentile(dur / np.maximum(g, 1), 10)),
      "p90", float(np.percentile(dur / np.maximum(g, 1), 90)))
order = np.argsort(-dur)[:5]
print("longest waves (dur, groups, start):", [(int(dur[i]), int(g[i]), int(t0[i])) for i in order])
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64); xcc = (hw >> np.uint64(32)).astype(np.int64)
simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
key = key * 4 + simd
uniq = np.unique(key)
busy = np.array([dur[key == k].sum() for k in uniq]); last = np.array([t1[key == k].max() for k in uniq]); grp = np.array([g[key == k].sum() for k in uniq])
print("SIMDs seen", len(uniq), "| groups per SIMD min/med/max", int(grp.min()), float(np.median(grp)), int(grp.max()),
      "| last end per SIMD min/med/max", int(last.min()), float(np.median(last)), int(last.max()))
# concurrency over time: number of resident waves at 10 sample points
for q in (0.1, 0.3, 0.5, 0.7, 0.9):
    t = q * t1.max()
    print(f"resident waves at {q:.1f} of the span:", int(((t0 <= t) & (t1 > t)).sum()))

#!/bin/bash
# counters + timeline of both compositing kernels at a workload -> gpurun_out/composite_diag_<wl>.json (tools/collect_profiles.py copies it)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
wl=${1:-C3}
B=$GRAFT_REPO_ROOT/tools/_build
D3GA_LIB_PATH=$B/libd3ga_hip_diag.so timeout 300 python tools/diag_scan.py $wl gpurun_out/_bwd_cnt.json > gpurun_out/diag_bwd_counters.log 2>&1
D3GA_LIB_PATH=$B/libd3ga_hip_timeline.so timeout 300 python tools/diag_scan.py $wl gpurun_out/_bwd_tl.json > gpurun_out/diag_bwd_timeline.log 2>&1
D3GA_LIB_PATH=$B/libd3ga_hip_diag.so timeout 300 python tools/diag_fwd.py $wl gpurun_out/_fwd_cnt.json > gpurun_out/diag_fwd_counters.log 2>&1
D3GA_LIB_PATH=$B/libd3ga_hip_timeline.so timeout 300 python tools/diag_fwd.py $wl gpurun_out/_fwd_tl.json > gpurun_out/diag_fwd_timeline.log 2>&1
python - $wl <<'PY'
import json, sys
wl = sys.argv[1]
L = lambda n: json.load(open(f"gpurun_out/{n}.json"))
bc, bt, fc, ft = L("_bwd_cnt"), L("_bwd_tl"), L("_fwd_cnt"), L("_fwd_tl")
tl = ("span_us", "wave_duration_us_p50_p90_max", "wave_start_us_p50_p90", "resident_waves_at_fraction_of_span", "share_of_wave_lifetime_in_blend_loop")
out = {"workload": wl, "note": "counters from the counter build (its timing is perturbed), timeline from the timeline build (per-wave s_memrealtime "
       "stamps, no atomics); both are the product kernels with `#ifdef D3GA_DIAG*` blocks compiled in (tools/_build/, never shipped)",
       "backward": {**{k: v for k, v in bc.items() if k not in tl and k != "lib_sha256"}, **{k: bt[k] for k in tl if k in bt}},
       "forward": {**{k: v for k, v in fc.items() if k not in tl and k != "lib_sha256"}, **{k: ft[k] for k in tl if k in ft}}}
json.dump(out, open(f"gpurun_out/composite_diag_{wl}.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY

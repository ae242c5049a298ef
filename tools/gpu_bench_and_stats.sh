#!/bin/bash
# headline bench line (with its in-graph stage events) + rocprofv3 kernel stats of the same command; $1: extra bench args
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step $1 > gpurun_out/bench_line.log 2> gpurun_out/bench_line.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_line.log").read().strip().split("\n")[-1])
    print(d["value"], d["ms_per_step"], d.get("stage_events"))
    print({k: v["ms"] for k, v in d["kernels"].items()}, d["roofline"]["frac"])
except Exception as e:
    print("bench line FAILED", e); print(open("gpurun_out/bench_line.err").read()[-3000:])
PY
rm -rf gpurun_out/ks_now
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/ks_now" -o ks -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-train-step $1 > "$GRAFT_REPO_ROOT/gpurun_out/ks_now.log" 2>&1 )
find gpurun_out/ks_now -name "*_kernel_trace.csv" -size +20M -delete
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ks_now/**/ks_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:20]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.2f}')
PY

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_step && mkdir -p gpurun_out/prof_step
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_step" -o step -- python "$GRAFT_REPO_ROOT/tools/prof_step.py" ${1:-color} > "$GRAFT_REPO_ROOT/gpurun_out/prof_step/log.txt" 2>&1 )
find gpurun_out/prof_step -name "*_kernel_trace.csv" -delete
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_step/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms per step", tot / 25 / 1e6)
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/25/1e3:9.1f} us/step  x{int(r["Calls"])/25:5.1f}  {r["Name"][:110]}')
PY

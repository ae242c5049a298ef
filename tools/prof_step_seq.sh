#!/bin/bash
# the kernels of ONE training step in launch order with their durations (rocprofv3 kernel trace; the last of 25 steps)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_seq && mkdir -p gpurun_out/prof_seq
( cd /tmp && $EXTRA_ENV timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_seq" -o step -- python "$GRAFT_REPO_ROOT/tools/prof_step.py" ${1:-color} > "$GRAFT_REPO_ROOT/gpurun_out/prof_seq/log.txt" 2>&1 )
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_seq/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
# one step = the kernels after the last occurrence of the step's first kernel pattern: find the period
names = [r["Kernel_Name"] for r in rows]
per = None
for p in range(20, n // 3):
    if names[-p:] == names[-2 * p:-p] == names[-3 * p:-2 * p]:
        per = p
        break
out = open("gpurun_out/prof_seq/seq.txt", "w")
print("kernels per step:", per, file=out)
tot = 0
for r in rows[-per:] if per else rows[-300:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{d:9.1f} us  grid {r.get('Grid_Size','?'):>10}  {r['Kernel_Name'][:150]}", file=out)
print("sum us", tot, file=out)
out.close()
PY
find gpurun_out/prof_seq -name "*_kernel_trace.csv" -delete
tail -3 gpurun_out/prof_seq/seq.txt

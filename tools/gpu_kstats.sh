#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the headline bench for one or more env settings
# usage: gpu_kstats.sh "NAME1:VAR=val,VAR2=val NAME2:" [extra bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for spec in $1; do
  name=${spec%%:*}; envs=${spec#*:}
  rm -rf gpurun_out/ks_$name
  ( cd /tmp && env $(echo "$envs" | tr ',' ' ') timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/ks_$name" -o ks -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-stage-events $2 > "$GRAFT_REPO_ROOT/gpurun_out/ks_$name.log" 2>&1 )
  find gpurun_out/ks_$name -name "*_kernel_trace.csv" -size +20M -delete
  python - "$name" <<'PY'
import csv, glob, sys
name = sys.argv[1]
f = glob.glob(f"gpurun_out/ks_{name}/**/ks_kernel_stats.csv", recursive=True)
if not f:
    print(name, "no stats"); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("==", name)
for r in rows[:22]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.2f} total_ms {float(r["TotalDurationNs"])/1e6:8.3f}')
PY
done

#!/bin/bash
# A/B of knob settings on the single-view headline step (C3 unless BENCH_ARGS says otherwise), interleaved rounds in one box,
# then rocprofv3 per-kernel averages per setting.
# usage: gpu_knob_ab.sh "knobspec1 knobspec2" [rounds] [pytest -k expression run first]     (knobspec: name=value[,..]; "-" = defaults)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
KNOBS=${1:-"-"}; ROUNDS=${2:-2}
: > gpurun_out/knob_ab.log
if [ -n "$3" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -k "$3" 2>&1 | tail -15 | cut -c1-300 >> gpurun_out/knob_ab.log
fi
for round in $(seq 1 $ROUNDS); do
 for kn in $KNOBS; do
  if [ "$kn" = "-" ]; then unset D3GA_KNOBS; else export D3GA_KNOBS=$kn; fi
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-train-step $BENCH_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$kn', $round, d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})" >> gpurun_out/knob_ab.log
 done
done
for kn in $KNOBS; do
  if [ "$kn" = "-" ]; then unset D3GA_KNOBS; name=default; else export D3GA_KNOBS=$kn; name=$(echo $kn | tr '=,' '__'); fi
  rm -rf gpurun_out/ks_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/ks_$name" -o ks -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-stage-events $BENCH_ARGS > "$GRAFT_REPO_ROOT/gpurun_out/ks_$name.log" 2>&1 )
  find gpurun_out/ks_$name -name "*_kernel_trace.csv" -size +20M -delete
  python - "$name" >> gpurun_out/knob_ab.log <<'PY'
import csv, glob, sys
name = sys.argv[1]
f = glob.glob(f"gpurun_out/ks_{name}/**/ks_kernel_stats.csv", recursive=True)
if not f:
    print(name, "no stats"); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("==", name)
for r in rows[:18]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.2f}')
PY
done
unset D3GA_KNOBS
cat gpurun_out/knob_ab.log

#!/bin/bash
# render_pair forward (DUAL instantiation): register budget for 5 (spills 7-10 VGPRs) against 4 wavefronts per SIMD (119 VGPRs, no scratch)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for round in 1 2; do
 for lib in default dual4; do
  if [ $lib = default ]; then unset D3GA_LIB_PATH; else export D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_$lib.so; fi
  timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); t=d['training_step']
print('$lib', d['ms_per_step'], {k:v for k,v in t.items() if 'pair' in k and 'color' not in k})"
 done
done
unset D3GA_LIB_PATH
for lib in default dual4; do
  if [ $lib = default ]; then unset D3GA_LIB_PATH; else export D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_$lib.so; fi
  rm -rf /tmp/ksd; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksd -o k -- python $GRAFT_REPO_ROOT/tools/time_pair.py > /dev/null 2>&1 )
  python - $lib <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ksd/**/k_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "composite" in r["Name"]: print(sys.argv[1], r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
done

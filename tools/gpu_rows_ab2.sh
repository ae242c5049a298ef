#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for wl in "--workload C5" "--workload C3 --scale-mult 2"; do
 for r in 1 2; do
  D3GA_BWD_ROWS=$r timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step $wl 2>gpurun_out/ab_rows_$r.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$wl rows$r', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e:
    print('rows$r FAILED', e)"
 done
done

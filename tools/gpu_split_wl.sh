#!/bin/bash
# backward split (default: device-side count) against none on several workloads; full GPU suite first
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_tests.sh
for wl in "--workload C3" "--workload C1" "--workload C2" "--workload C4" "--workload C5" "--workload C3 --scale-mult 2" "--workload C3 --gaussian-order random"; do
 for h in -1 0; do
  D3GA_BWD_SPLIT=$h timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-train-step $wl 2>gpurun_out/ab_split.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$wl split$h', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e:
    print('split$h FAILED', e)"
 done
done

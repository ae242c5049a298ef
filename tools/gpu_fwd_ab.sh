#!/bin/bash
# round 5: the two-launch forward (D3GA_FWD_IMPL=1) against the one-launch forward (0): GPU tests under the new default, then
# interleaved headline bench lines and rocprofv3 kernel stats of both.  $1: pytest -k expression ("" = whole suite, "skip" = none)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$1" != "skip" ]; then bash tools/gpu_tests.sh "$1"; fi
for round in 1 2; do
 for v in 0 1; do
  D3GA_FWD_IMPL=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step 2>gpurun_out/ab_fwd_$v.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('impl $v', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e:
    print('impl $v FAILED', e)"
 done
done
tail -5 gpurun_out/ab_fwd_1.err
bash tools/gpu_kstats.sh "old:D3GA_FWD_IMPL=0 new:D3GA_FWD_IMPL=1" 2>&1 | grep -E "^==|composite|cull|sort|scan" 

#!/bin/bash
# round 5: the sort-emitted block lists + lists blend (D3GA_FWD_IMPL=2, the default) against the one-launch forward (0) and the
# two-launch forward (1): GPU tests under the default, interleaved headline bench lines, rocprofv3 kernel stats.
# $1: pytest -k expression ("" = whole suite, "skip" = none); $2: the impls to compare (default "0 2")
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$1" != "skip" ]; then bash tools/gpu_tests.sh "$1"; fi
IMPLS=${2:-"0 2"}
for round in 1 2; do
 for v in $IMPLS; do
  D3GA_FWD_IMPL=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step 2>gpurun_out/ab_fwd_$v.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('impl $v', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e:
    print('impl $v FAILED', e)"
 done
done
tail -5 gpurun_out/ab_fwd_2.err
KS=""
for v in $IMPLS; do KS="$KS i$v:D3GA_FWD_IMPL=$v"; done
bash tools/gpu_kstats.sh "$KS" 2>&1 | grep -E "^==|composite|cull|sort|scan|scatter|preprocess_kernel"

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
echo "== parity D3GA_BWD_SPLIT=${TEST_SPLIT:-256}"
D3GA_BWD_SPLIT=${TEST_SPLIT:-256} timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_known_answers.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
fi
for round in 1 2; do
 for h in ${SPLITS:-0 128 256 400 700}; do
  D3GA_BWD_SPLIT=$h $EXTRA_ENV timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step ${BENCH_ARGS} 2>gpurun_out/ab_split_$h.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('split$h', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e:
    print('split$h FAILED', e)"
 done
done

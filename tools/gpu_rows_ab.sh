#!/bin/bash
# backward rows-per-block A/B (D3GA_BWD_ROWS = 1 | 2 | 4): parity tests under each setting, then interleaved bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in ${ROWS:-2 4}; do
  echo "== parity D3GA_BWD_ROWS=$r"
  D3GA_BWD_ROWS=$r timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_known_answers.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
done
for round in 1 2; do
 for r in 1 ${ROWS:-2 4}; do
  D3GA_BWD_ROWS=$r $EXTRA_ENV timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step ${BENCH_ARGS} 2>gpurun_out/ab_rows_$r.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('rows$r', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e:
    print('rows$r FAILED', e)"
 done
done

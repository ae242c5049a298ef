#!/bin/bash
# persistent backward (D3GA_BWD_PERSIST = workgroups per CU): parity tests, then interleaved bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
echo "== parity D3GA_BWD_PERSIST=4"
D3GA_BWD_PERSIST=4 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_known_answers.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
fi
for round in 1 2; do
 for spec in ${SPECS:-base:D3GA_BWD_PERSIST=0 p4:D3GA_BWD_PERSIST=4 p4s128:D3GA_BWD_PERSIST=4,D3GA_PERSIST_SLOTS=128 p3:D3GA_BWD_PERSIST=3 p2s512:D3GA_BWD_PERSIST=2,D3GA_PERSIST_SLOTS=512}; do
  name=${spec%%:*}; envs=${spec#*:}
  env $(echo "$envs" | tr ',' ' ') timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step ${BENCH_ARGS} 2>gpurun_out/ab_$name.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$name', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e:
    print('$name FAILED', e)"
 done
done

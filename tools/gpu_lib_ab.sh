#!/bin/bash
# A/B of the shipped library against tools/_build/libd3ga_hip_<tag>.so builds inside one box, interleaved rounds.
# usage: gpu_lib_ab.sh "tag1 tag2" [rounds] [--tests]   (--tests: the GPU suite on the shipped library first)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAGS=${1:-""}; ROUNDS=${2:-2}
: > gpurun_out/lib_ab.log
if [ "$3" = "--tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
  grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/pytest_gpu.log | cut -c1-300 | tail -20 >> gpurun_out/lib_ab.log
fi
for round in $(seq 1 $ROUNDS); do
 for lib in default $TAGS; do
  if [ $lib = default ]; then unset D3GA_LIB_PATH; else export D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_$lib.so; fi
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-train-step 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib', $round, d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})" >> gpurun_out/lib_ab.log
 done
done
unset D3GA_LIB_PATH
cat gpurun_out/lib_ab.log

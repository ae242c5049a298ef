#!/bin/bash
# A/B of composite variants inside one box: interleaved rounds.  usage: gpu_ab.sh "160 128 32" [test_variant]  (32 = work-ordered dispatch, 128 = exact cull)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
VARS=${1:-"160"}
: > gpurun_out/ab.log
for round in 1 2; do
 for v in $VARS; do
  echo "variant $v round $round" >> gpurun_out/ab.log
  D3GA_COMPOSITE_VARIANT=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-train-step 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})" >> gpurun_out/ab.log
 done
done
if [ -n "$2" ]; then D3GA_COMPOSITE_VARIANT=$2 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 >> gpurun_out/ab.log; fi
cat gpurun_out/ab.log

#!/bin/bash
# A/B of composite variants inside one box: interleaved rounds
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/ab.log
for round in 1 2; do
 for v in 0 1 2 3; do
  echo "variant $v round $round" >> gpurun_out/ab.log
  D3GA_COMPOSITE_VARIANT=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})" >> gpurun_out/ab.log
 done
done
D3GA_COMPOSITE_VARIANT=3 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 >> gpurun_out/ab.log
cat gpurun_out/ab.log

#!/bin/bash
# A/B of environment-selected variants inside one box, interleaved rounds.
# usage: gpu_ab_env.sh "NAME1:VAR=val,VAR2=val NAME2:..." [pytest -k expression run under the FIRST setting]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/ab_env.log
for round in 1 2; do
 for spec in $1; do
  name=${spec%%:*}; envs=${spec#*:}
  echo "== $name round $round" >> gpurun_out/ab_env.log
  env $(echo "$envs" | tr ',' ' ') timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-train-step ${BENCH_ARGS} 2>gpurun_out/ab_env.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})" >> gpurun_out/ab_env.log 2>&1
 done
done
if [ -n "$2" ]; then
  spec=${1%% *}; envs=${spec#*:}
  env $(echo "$envs" | tr ',' ' ') timeout 1500 python -m pytest tests -m gpu -q -x -k "$2" 2>&1 | tail -15 >> gpurun_out/ab_env.log
fi
cat gpurun_out/ab_env.log

#!/bin/bash
# A/B of environment settings on the single-view headline step, interleaved rounds in one box
# usage: gpu_env_ab.sh "NAME1:VAR=val,VAR2=val NAME2:" [rounds] [extra bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
SPECS=${1:-"default:"}; ROUNDS=${2:-3}
: > gpurun_out/env_ab.log
for round in $(seq 1 $ROUNDS); do
 for spec in $SPECS; do
  name=${spec%%:*}; envs=${spec#*:}
  env $(echo "$envs" | tr ',' ' ') timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-train-step --no-stage-events $3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$name', $round, d['value'], d['ms_per_step'], d.get('step_ms'))" >> gpurun_out/env_ab.log
 done
done
cat gpurun_out/env_ab.log

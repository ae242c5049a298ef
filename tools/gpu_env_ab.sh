#!/bin/bash
# interleaved headline bench lines + kernel stats for environment settings: gpu_env_ab.sh "base: x:VAR=1,VAR2=2"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for round in 1 2; do
 for spec in $1; do
  name=${spec%%:*}; envs=${spec#*:}
  env $(echo "$envs" | tr ',' ' ') timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step 2>gpurun_out/ab_env_$name.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$name', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e:
    print('$name FAILED', e)"
 done
done
bash tools/gpu_kstats.sh "$1" 2>&1 | grep -E "^==|sort|scan|scatter"

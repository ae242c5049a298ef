#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_ab_env.sh "base:D3GA_WAVE_PRIO=0 f1:D3GA_WAVE_PRIO=1 f2:D3GA_WAVE_PRIO=2 b1:D3GA_WAVE_PRIO=4 b2:D3GA_WAVE_PRIO=8 b3:D3GA_WAVE_PRIO=12" > /dev/null 2>&1
cat gpurun_out/ab_env.log

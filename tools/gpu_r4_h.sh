#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
W5=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_abl0w5.so
bash tools/gpu_ab_env.sh "base:D3GA_X=0 s256:D3GA_MERGE_SLOTS=256 w5s256:D3GA_MERGE_SLOTS=256,D3GA_LIB_PATH=$W5 w5s512:D3GA_LIB_PATH=$W5" > /dev/null 2>&1
cat gpurun_out/ab_env.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q -x -k "parity or known" 2>&1 | tail -5 ) > gpurun_out/r4_tests_d.log
bash tools/gpu_ab_env.sh "exit512:D3GA_X=0 noexit512:D3GA_TILE_ASSIGN=10 exit256:D3GA_MERGE_SLOTS=256 noexit256:D3GA_MERGE_SLOTS=256,D3GA_TILE_ASSIGN=10" > /dev/null 2>&1
D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_timeline.so timeout 300 python tools/diag_scan.py C3 > gpurun_out/r4_diag_bwd_timeline.log 2>&1
D3GA_MERGE_SLOTS=256 D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_timeline.so timeout 300 python tools/diag_scan.py C3 > gpurun_out/r4_diag_bwd_timeline256.log 2>&1
cat gpurun_out/r4_tests_d.log; cat gpurun_out/ab_env.log;  grep -v amdgpu.ids gpurun_out/r4_diag_bwd_timeline.log | tail -20; grep -v amdgpu.ids gpurun_out/r4_diag_bwd_timeline256.log | tail -20

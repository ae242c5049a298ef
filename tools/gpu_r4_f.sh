#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( D3GA_TILE_NW=3 D3GA_MERGE_SLOTS=256 timeout 1200 python -m pytest tests -m gpu -q -x -k "parity or known" 2>&1 | tail -5 ) > gpurun_out/r4_tests_f.log
bash tools/gpu_ab_env.sh "nw4s512:D3GA_X=0 nw4s256:D3GA_MERGE_SLOTS=256 nw3s256:D3GA_TILE_NW=3,D3GA_MERGE_SLOTS=256 nw3s512:D3GA_TILE_NW=3,D3GA_MERGE_SLOTS=512" > /dev/null 2>&1
D3GA_TILE_NW=3 D3GA_MERGE_SLOTS=256 D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_timeline.so timeout 300 python tools/diag_scan.py C3 > gpurun_out/r4_diag_bwd_timeline_nw3.log 2>&1
tail -3 gpurun_out/r4_tests_f.log; cat gpurun_out/ab_env.log;  grep -v amdgpu.ids gpurun_out/r4_diag_bwd_timeline_nw3.log | tail -20

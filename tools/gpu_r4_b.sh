#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_timeline.so timeout 300 python tools/diag_scan.py C3 > gpurun_out/r4_diag_bwd_timeline.log 2>&1
D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_timeline.so timeout 300 python tools/diag_fwd.py C3 > gpurun_out/r4_diag_fwd_timeline.log 2>&1
( timeout 1200 python -m pytest tests -m gpu -q -k "shared_decisions" 2>&1 | grep -E "shared decisions|passed|failed|^E  " ) > gpurun_out/r4_tests_b.log
grep -v amdgpu.ids gpurun_out/r4_diag_bwd_timeline.log | tail -22; grep -v amdgpu.ids gpurun_out/r4_diag_fwd_timeline.log | tail -24; tail -25 gpurun_out/r4_tests_b.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_diag.so timeout 300 python tools/diag_scan.py ${1:-C3} 2>&1 | tail -16 | tee gpurun_out/diag_scan.log

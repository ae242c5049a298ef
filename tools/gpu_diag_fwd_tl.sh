#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_timeline.so timeout 300 python tools/diag_fwd.py C3 2>&1 | grep -v "Warning\|warn\|amdgpu\|return Variable" | cut -c1-900

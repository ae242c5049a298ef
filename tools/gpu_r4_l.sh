#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_timeline.so timeout 300 python tools/diag_fwd.py C3 2>&1 | grep -v amdgpu.ids | tail -22

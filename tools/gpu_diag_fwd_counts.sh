#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for sp in 1 0; do
  echo "== D3GA_SPANS=$sp"
  D3GA_SPANS=$sp D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_diag.so timeout 300 python tools/diag_fwd.py C3 2>&1 | grep -E "active_waves|lane efficiency" | cut -c1-400
done

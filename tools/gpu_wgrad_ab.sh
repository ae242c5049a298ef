#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for round in 1 2; do for m in 1 3 4 5; do D3GA_WGRAD_WS=$m timeout 120 python tools/time_wgrad.py 2>&1 | tail -1; done; done
for m in 3 4 5; do D3GA_WGRAD_WS=$m timeout 120 python tools/time_wgrad.py 135000 2>&1 | tail -1; done
for m in ${TEST_MODES:-4}; do echo "== tests D3GA_WGRAD_WS=$m"; D3GA_WGRAD_WS=$m timeout 900 python -m pytest tests/test_gpu_mlp.py -m gpu -q -x 2>&1 | tail -2; done

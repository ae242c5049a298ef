#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for round in 1 2; do for m in 1 0; do D3GA_WGRAD_WS=$m timeout 120 python tools/time_wgrad.py 2>&1 | tail -1; done; done



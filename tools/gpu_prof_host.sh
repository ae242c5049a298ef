#!/bin/bash
# host-side profile of the eager steps (frame, training step, render-pair training step) at C3 -> gpurun_out/prof_host_*.txt
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in ${MODES:-frame train pair}; do timeout 600 python tools/prof_host.py ${WL:-C3} $m > gpurun_out/prof_host_$m.txt 2>&1; grep "^==" gpurun_out/prof_host_$m.txt; done

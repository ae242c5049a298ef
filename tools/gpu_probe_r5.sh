#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/probe/ext_event.py > gpurun_out/ext_event.log 2>&1
cat gpurun_out/ext_event.log | tail -20
bash tools/gpu_tests.sh

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
( D3GA_FWD_PERSIST=4096 timeout 1200 python -m pytest tests -m gpu -q -x -k "parity or known" 2>&1 | tail -3 ) > gpurun_out/r4_tests_m.log
bash tools/gpu_ab_env.sh "p0:D3GA_FWD_PERSIST=0 p3072:D3GA_FWD_PERSIST=3072 p4096:D3GA_FWD_PERSIST=4096 p5120:D3GA_FWD_PERSIST=5120" > /dev/null 2>&1
tail -2 gpurun_out/r4_tests_m.log; cat gpurun_out/ab_env.log

#!/bin/bash
# the GPU test suite (optionally a -k expression as $1), full failure text into gpurun_out/pytest_gpu.log
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$1" ]; then timeout 1500 python -m pytest tests -m gpu -q -k "$1" 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
else timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -150 > gpurun_out/pytest_gpu.log; fi
grep -E "^(FAILED|ERROR)|passed|failed|AssertionError" gpurun_out/pytest_gpu.log | cut -c1-400 | tail -60

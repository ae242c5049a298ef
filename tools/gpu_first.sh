#!/bin/bash
# first GPU contact: tests, then a short bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.log
nproc >> gpurun_out/gpu_info.log
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -80 > gpurun_out/pytest_gpu.log
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/bench.err
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err

#!/bin/bash
# GPU tests + rocprofv3 kernel trace of a short bench + a full bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-train-step > "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.log" 2> "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.err" )
find gpurun_out/prof -name "*kernel_stats*" | head
find gpurun_out/prof -name "*_kernel_trace.csv" -size +20M -delete
timeout 900 python bench.py --steps 100 --warmup 20 > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/bench.err
tail -8 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.log; tail -3 gpurun_out/bench.err

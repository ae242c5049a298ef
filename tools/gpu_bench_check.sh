#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_view_sharded.py -m gpu -q -x 2>&1 | tail -15
timeout 900 python bench.py --steps 30 --warmup 8 > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cut -c1-3000 gpurun_out/bench.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --single-device --backend gloo --steps 10 --warmup 3 --no-cpu-baseline --workload C4 > gpurun_out/bench_n2.log 2> gpurun_out/bench_n2.err; tail -2 gpurun_out/bench_n2.err; cut -c1-1500 gpurun_out/bench_n2.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
# 2-rank path of bench.py on ONE GPU (gloo, both ranks on cuda:0): exercises barrier / all-reduce / max-over-ranks
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
   bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo --single-device --workload C1 --no-stage-events 2>&1 | tail -3 | cut -c1-900
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

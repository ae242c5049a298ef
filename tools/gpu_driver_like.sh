#!/bin/bash
# what the driver runs at round end, in its order: the GPU suite, smoke(), the default bench line (+ the poisoned suite once)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/driver_like.log
( timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -2 ) >> gpurun_out/driver_like.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) >> gpurun_out/driver_like.log
( s=$(date +%s); timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | cut -c1-700 | tail -3; echo "bench wall $(( $(date +%s) - s )) s" ) >> gpurun_out/driver_like.log
echo "== poison" >> gpurun_out/driver_like.log
( D3GA_POISON=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2 ) >> gpurun_out/driver_like.log
cat gpurun_out/driver_like.log

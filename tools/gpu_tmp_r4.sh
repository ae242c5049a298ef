#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/fuzz50_loop.log
for i in $(seq 1 40); do
  D3GA_FUZZ_SCENE=C1 D3GA_FUZZ_N=200 timeout 300 python -m pytest tests -m gpu -q -x --tb=short -k "fuzz_ragged and (49] or 50])" > gpurun_out/fuzz50_one.log 2>&1
  if grep -q "failed" gpurun_out/fuzz50_one.log; then echo "== run $i FAILED" >> gpurun_out/fuzz50_loop.log; grep -v "^\.\.\.\." gpurun_out/fuzz50_one.log | tail -40 >> gpurun_out/fuzz50_loop.log; else echo "run $i ok" >> gpurun_out/fuzz50_loop.log; fi
done
tail -50 gpurun_out/fuzz50_loop.log



#!/bin/bash
# end-of-round checks on the final library: the GPU suite with poisoned allocations, the plain suite twice more, short campaigns
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/final_checks.log
echo "== poison" >> gpurun_out/final_checks.log
D3GA_POISON=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 >> gpurun_out/final_checks.log
for i in 1 2; do
  echo "== plain $i" >> gpurun_out/final_checks.log
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2 >> gpurun_out/final_checks.log
done
N_RASTER=${N_RASTER:-1000} N_C1=${N_C1:-200} N_PAIR=200 N_CHAIN=300 N_DEFORM=300 N_SHARD=100 N_LOSS=300 N_INIT=100 bash tools/gpu_campaigns.sh > /dev/null 2>&1
grep -E "^==|passed|failed" gpurun_out/campaigns.log >> gpurun_out/final_checks.log
cat gpurun_out/final_checks.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q -k "shared_decisions" 2>&1 | grep -E "shared decisions|passed|failed|^E  " ) > gpurun_out/r4_tests_c.log
bash tools/gpu_ab_env.sh "base:D3GA_X=0 f4:D3GA_FWD_LDS_TOTAL=10240 f3:D3GA_FWD_LDS_TOTAL=12800 f2:D3GA_FWD_LDS_TOTAL=20480 b3:D3GA_BWD_LDS_TOTAL=53760 b2:D3GA_BWD_LDS_TOTAL=81920" > /dev/null 2>&1
cat gpurun_out/r4_tests_c.log; cat gpurun_out/ab_env.log

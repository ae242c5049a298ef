#!/bin/bash
# round 5: the test of the direction-Jacobian path on the shipped library and on the planar-layout build, then their A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "jacobian" 2>&1 | tail -15 > gpurun_out/dcol_test.log
D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_planar.so timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 >> gpurun_out/dcol_test.log
grep -E "passed|failed|Error|assert" gpurun_out/dcol_test.log | cut -c1-300
bash tools/gpu_lib_ab.sh "planar" 3

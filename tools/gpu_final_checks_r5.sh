#!/bin/bash
# round 5: the final checks + the rasterizer campaign once more with the opt-in two-launch forward
cd "$GRAFT_REPO_ROOT" || exit 1
N_RASTER=${N_RASTER:-1500} N_C1=${N_C1:-200} bash tools/gpu_final_checks.sh > /dev/null 2>&1
cp gpurun_out/final_checks.log gpurun_out/final_checks_r5.log
echo "== raster_fuzz with D3GA_FWD_IMPL=1 (400 + 60 seeds)" >> gpurun_out/final_checks_r5.log
( D3GA_FWD_IMPL=1 D3GA_FUZZ_N=400 timeout 1800 python -m pytest tests -m gpu -q -k fuzz_ragged 2>&1 | tail -2 ) >> gpurun_out/final_checks_r5.log
( D3GA_FWD_IMPL=1 D3GA_FUZZ_SCENE=C1 D3GA_FUZZ_N=60 timeout 1800 python -m pytest tests -m gpu -q -k fuzz_ragged 2>&1 | tail -2 ) >> gpurun_out/final_checks_r5.log
echo "== whole suite with D3GA_FWD_IMPL=1" >> gpurun_out/final_checks_r5.log
( D3GA_FWD_IMPL=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2 ) >> gpurun_out/final_checks_r5.log
cat gpurun_out/final_checks_r5.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q -x -s -k "c5_window or color_train or view_dependent_colour or without_a_launcher or refuses_a_world or shared_decisions" 2>&1 | grep -E "shared decisions|C5 window|passed|failed|^E  |Error" | cut -c1-600 ) > gpurun_out/r4_tests_i.log
( timeout 900 python bench.py --train-step color --gpus 2 --single-device --backend gloo --steps 100 --warmup 5 > gpurun_out/r4_color2.log 2> gpurun_out/r4_color2.err )
( timeout 900 python bench.py --train-step color --steps 50 --warmup 5 > gpurun_out/r4_color1.log 2> gpurun_out/r4_color1.err )
cat gpurun_out/r4_tests_i.log; cut -c1-1500 gpurun_out/r4_color2.log; tail -3 gpurun_out/r4_color2.err | cut -c1-300; cut -c1-600 gpurun_out/r4_color1.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "l1 or L1" 2>&1 | tail -8
bash tools/gpu_kstats.sh "fused:D3GA_X=0" 2>&1 | grep -E "^==|composite_fwd|sum_partials|l1_mean"
BENCH_ARGS="" bash tools/gpu_ab_env.sh "fused:D3GA_X=0 separate:D3GA_L1_VALUE=separate" 2>&1 | tail -12 | cut -c1-220

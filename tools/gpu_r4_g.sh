#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for n in 2 4; do
( D3GA_BWD_SEGMENTS=$n timeout 1200 python -m pytest tests -m gpu -q -x -k "parity or known" 2>&1 | tail -4 ) > gpurun_out/r4_tests_g$n.log
done
bash tools/gpu_ab_env.sh "seg1:D3GA_BWD_SEGMENTS=1 seg2:D3GA_BWD_SEGMENTS=2 seg4:D3GA_BWD_SEGMENTS=4 seg4s256:D3GA_BWD_SEGMENTS=4,D3GA_MERGE_SLOTS=256" > /dev/null 2>&1
D3GA_BWD_SEGMENTS=4 D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_timeline.so timeout 300 python tools/diag_scan.py C3 > gpurun_out/r4_diag_bwd_timeline_seg4.log 2>&1
tail -3 gpurun_out/r4_tests_g2.log gpurun_out/r4_tests_g4.log; cat gpurun_out/ab_env.log;  grep -v amdgpu.ids gpurun_out/r4_diag_bwd_timeline_seg4.log | tail -20

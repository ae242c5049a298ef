#!/bin/bash
# round 4, first GPU call: the whole GPU suite under the shipped defaults, the headline bench, diagnostics of both compositing kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4_tests.log
( timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4_bench.log 2> gpurun_out/r4_bench.err )
for mode in timeline diag; do
  D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_$mode.so timeout 300 python tools/diag_scan.py C3 > gpurun_out/r4_diag_bwd_$mode.log 2>&1
  D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_$mode.so timeout 300 python tools/diag_fwd.py C3 > gpurun_out/r4_diag_fwd_$mode.log 2>&1
done
for a in 2 1 2 1; do
  echo "== assign $a" >> gpurun_out/r4_ab.log
  D3GA_TILE_ASSIGN=$a timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-train-step 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r4_ab.log
done
tail -5 gpurun_out/r4_tests.log; cat gpurun_out/r4_ab.log; tail -30 gpurun_out/r4_diag_*.log

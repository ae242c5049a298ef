#!/bin/bash
# timing ablations of the compositing backward (libd3ga_hip_abl<k>.so, built with D3GA_SCAN_ABL=k python d3ga_amd/csrc/build.py)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/ablate.log
run() {
  echo -n "$1 " >> gpurun_out/ablate.log
  D3GA_LIB_PATH=$1 timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-train-step --fixed-camera 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})" >> gpurun_out/ablate.log
}
for round in 1 2; do
run $GRAFT_REPO_ROOT/d3ga_amd/libd3ga_hip.so
for lib in $(ls $GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_abl*.so 2>/dev/null); do D3GA_ALLOW_ABLATION=1 run $lib; done
done
cat gpurun_out/ablate.log

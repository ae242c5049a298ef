#!/bin/bash
# A/B of library variants (tools/_build/libd3ga_hip_<tag>.so) and knob settings on the VIEW-BATCHED step (k = 4, C3), interleaved in one box
# usage: gpu_views_ab.sh "tag1 tag2" "knobspec1 knobspec2" [rounds]     (knobspec: name=value[,name=value]; "-" = defaults)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
TAGS=${1:-""}; KNOBS=${2:-"-"}; ROUNDS=${3:-2}
: > gpurun_out/views_ab.log
for round in $(seq 1 $ROUNDS); do
 for lib in default $TAGS; do
  for kn in $KNOBS; do
   if [ $lib = default ]; then unset D3GA_LIB_PATH; else export D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_$lib.so; fi
   if [ "$kn" = "-" ]; then unset D3GA_KNOBS; else export D3GA_KNOBS=$kn; fi
   timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-train-step --batch-views ${K:-4} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); b=d['batched_views']
print('$lib', '$kn', $round, 'single', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items() if 'composite' in k}, '| batched', b['ms_per_view'], {k:v['ms_per_view'] for k,v in b['kernels'].items()})" >> gpurun_out/views_ab.log
  done
 done
done
unset D3GA_LIB_PATH D3GA_KNOBS
cat gpurun_out/views_ab.log

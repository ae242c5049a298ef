#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r4_cut.err
for k in 1 2 4; do
  timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --force-cut --views-per-rank $k 2>>gpurun_out/r4_cut.err | grep "^{" > gpurun_out/r4_force_cut_C3_k$k.json
  python -c "
import sys,json
d=json.load(open('gpurun_out/r4_force_cut_C3_k$k.json')); print('k=$k |', d['value'], d['ms_per_step'], {k:v for k,v in (d.get('distributed') or {}).items() if k!='note'}, '|', d.get('grad_exchange_note'))"
done
grep -v "amdgpu.ids\|UserWarning\|run_backward\|socket.cpp" gpurun_out/r4_cut.err | tail -5 | cut -c1-300

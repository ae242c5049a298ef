#!/bin/bash
# round 5: ColorField's per-row input through d3ga_color_rows_* against the encoding call + torch.cat (tools/_build/mlp_old2.py), same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_train_soak.py tests/test_gpu_view_sharded.py tests/test_gpu_bench_contract.py -m gpu -q 2>&1 | tail -15 > gpurun_out/rows_test.log
grep -E "passed|failed|Error|assert" gpurun_out/rows_test.log | cut -c1-300
line() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); t=d['training_step']
print('$1', d['ms_per_step'], {k:v for k,v in t.items() if 'color' in k})"; }
cp d3ga_amd/mlp.py /tmp/mlp_new.py
for r in 1 2; do
  cp /tmp/mlp_new.py d3ga_amd/mlp.py; line new
  cp tools/_build/mlp_old2.py d3ga_amd/mlp.py; line old
done
cp /tmp/mlp_new.py d3ga_amd/mlp.py

#!/bin/bash
# fused vs per-layer field networks: parity tests, then the --field-mlp timing under both settings, then kernel stats.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mlp.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/mlp_ab.log
for r in 1 2; do for f in 1 0; do
  echo "== fused=$f round $r" >> gpurun_out/mlp_ab.log
  D3GA_MLP_FUSED=$f timeout 300 python bench.py --field-mlp 2>>gpurun_out/mlp_ab.err | tail -1 >> gpurun_out/mlp_ab.log
done; done
cd /tmp && D3GA_MLP_FUSED=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_mlp -o mlp -- python $GRAFT_REPO_ROOT/bench.py --field-mlp > /dev/null 2>&1
python - <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/mlp_ab.log
import csv,glob
for f in glob.glob('/tmp/prof_mlp/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:12]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
cat $GRAFT_REPO_ROOT/gpurun_out/mlp_ab.log

#!/bin/bash
# timing ablations of the fused field-network forward: D3GA_CHAIN_ABL bits 1 no stores, 2 no weight DMA, 4 no MFMA, 8 no input load, 16 no sign store
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/mlp_abl.log
for abl in ${ABLS:-0 1 2 4 8 16 17 3 7 31}; do for grid in ${GRIDS:-512}; do export D3GA_CHAIN_GRID=$grid
  rm -rf /tmp/prof_abl
  ( cd /tmp && D3GA_CHAIN_ABL=$abl D3GA_MLP_FUSED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_abl -o kt -- python $R/bench.py --field-mlp --steps 10 --warmup 3 > /tmp/abl.log 2>&1 )
  python - $abl <<'PY' >> $R/gpurun_out/mlp_abl.log
import csv,glob,sys
for f in glob.glob('/tmp/prof_abl/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if "chain_fwd" in r["Name"] and "false" in r["Name"]: print('abl', sys.argv[1], 'grid', __import__('os').environ.get('D3GA_CHAIN_GRID'), r['Calls'], round(float(r['AverageNs'])/1000,1), 'us')
PY
done; done
cat $R/gpurun_out/mlp_abl.log

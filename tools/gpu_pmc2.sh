#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  name=$1; shift
  rm -rf gpurun_out/pmc_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "composite" --output-format csv \
      -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 3 --no-cpu-baseline --no-stage-events \
      > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log" 2>&1 )
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES_EQ_64
python - <<'PY'
import csv, collections
for name in ('sq1','sq2','sq3'):
    try:
        rows=list(csv.DictReader(open(f'gpurun_out/pmc_{name}/pmc_counter_collection.csv')))
    except Exception as e:
        print(name, 'missing', e); continue
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k=r['Kernel_Name'].split('(')[0].replace('void ','')[:40]
        d[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in d.items():
        print(name, k, {c: round(sum(x)/len(x)/1e6,3) for c,x in v.items()})
PY
tail -3 gpurun_out/pmc_sq3.log

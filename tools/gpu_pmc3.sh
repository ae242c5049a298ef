#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  name=$1; shift
  rm -rf gpurun_out/pmc_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "composite" --output-format csv \
      -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 3 --no-cpu-baseline --no-train-step --no-stage-events --no-graph \
      > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log" 2>&1 )
}
run a1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY
run a2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run a3 TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum
run a4 TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
run a5 TCC_BUSY_avr TCC_EA0_WRREQ_ATOMIC_DRAM_sum
python - <<'PY'
import csv, collections
for name in ('a1','a2','a3','a4','a5'):
    try:
        rows=list(csv.DictReader(open(f'gpurun_out/pmc_{name}/pmc_counter_collection.csv')))
    except Exception as e:
        print(name, 'missing', e); continue
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k=r['Kernel_Name'].split('(')[0].replace('void ','')[:34]
        d[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in d.items():
        print(name, k, {c: round(sum(x)/len(x)/1e6,3) for c,x in v.items()})
PY

#!/bin/bash
# round 2 iteration loop: GPU tests, A/B of compositing variants, SQ counters of the compositing kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
VARS=${1:-"63 127"}
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
bash tools/gpu_ab.sh "$VARS" | grep -v "^variant" 
pmc() {  # name, counters...
  name=$1; shift
  rm -rf gpurun_out/pmc_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "composite" --output-format csv \
      -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 3 --no-cpu-baseline --no-train-step --no-stage-events --no-graph \
      > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log" 2>&1 )
}
pmc sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python - <<'PY'
import csv, collections, glob
for name in ("sq","sq2"):
    f=glob.glob(f"gpurun_out/pmc_{name}/**/pmc_counter_collection.csv", recursive=True)
    if not f: print(name,"missing"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0][-50:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(name, k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY

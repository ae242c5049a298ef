#!/bin/bash
# SQ / LDS counters of the compositing backward for the default library and the libraries named in $1 (e.g. "abl1 abl13")
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
pmc() {  # lib tag, pass name, counters...
  lib=$1; name=$2; shift; shift
  rm -rf gpurun_out/pmcb_${lib}_$name
  ( cd /tmp && D3GA_ALLOW_ABLATION=1 D3GA_LIB_PATH=$( [ -n "$lib" ] && echo $GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_$lib.so || echo $GRAFT_REPO_ROOT/d3ga_amd/libd3ga_hip.so ) timeout 600 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "${KREGEX:-composite_bwd}" --output-format csv \
      -d "$GRAFT_REPO_ROOT/gpurun_out/pmcb_${lib}_$name" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 3 --no-cpu-baseline --no-train-step --no-stage-events --no-graph --fixed-camera \
      > "$GRAFT_REPO_ROOT/gpurun_out/pmcb_${lib}_$name.log" 2>&1 )
}
for lib in "" $1; do
  pmc "$lib" sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  pmc "$lib" sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR
done
python - "$1" <<'PY'
import csv, collections, glob, sys
for lib in [""] + sys.argv[1].split():
  for name in ("sq","sq2"):
    f=glob.glob(f"gpurun_out/pmcb_{lib}_{name}/**/pmc_counter_collection.csv", recursive=True)
    if not f: print(lib, name,"missing"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(lib or "default", name, k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY

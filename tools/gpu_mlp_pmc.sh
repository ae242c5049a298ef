#!/bin/bash
# kernel durations and SQ counters of the fused field-network forward (bench.py --field-mlp)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/mlp_prof; mkdir -p $R/gpurun_out/mlp_prof
( cd /tmp && D3GA_MLP_FUSED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mlp_prof/kt -o kt -- python $R/bench.py --field-mlp --steps 20 --warmup 5 > $R/gpurun_out/mlp_prof/kt.log 2>&1 )
pmc() { name=$1; shift
  ( cd /tmp && D3GA_MLP_FUSED=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "${KREGEX:-chain_fwd}" --output-format csv -d $R/gpurun_out/mlp_prof/$name -o pmc -- python $R/bench.py --field-mlp --steps 3 --warmup 2 > $R/gpurun_out/mlp_prof/$name.log 2>&1 )
}
pmc sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
pmc sq3 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_WAIT_INST_LDS
python - <<'PY'
import csv, collections, glob
for f in glob.glob('gpurun_out/mlp_prof/kt/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
for name in ("sq","sq2","sq3"):
    f=glob.glob(f"gpurun_out/mlp_prof/{name}/**/pmc_counter_collection.csv", recursive=True)
    if not f: print(name,"missing"); print(open(f"gpurun_out/mlp_prof/{name}.log").read()[-600:]); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(name, k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY

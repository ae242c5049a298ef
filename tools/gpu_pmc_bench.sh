#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python bench.py --pmc --steps 20 --warmup 5 --no-train-step --no-cpu-baseline > gpurun_out/bench_pmc.log 2> gpurun_out/bench_pmc.err
tail -3 gpurun_out/bench_pmc.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_pmc.log").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:1500])
PY
cp profiles/pmc_C3.json gpurun_out/pmc_C3.json

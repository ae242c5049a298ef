#!/bin/bash
# PMC passes of the headline command with the library in the tree; prints the compositing / cull kernels' counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python bench.py --pmc --steps 10 --warmup 3 --no-cpu-baseline --no-train-step > gpurun_out/pmc_bench.log 2> gpurun_out/pmc_bench.err
cp profiles/pmc_C3.json gpurun_out/pmc_C3_now.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_C3_now.json"))
for k, v in d.items():
    if any(s in k for s in ("cull", "composite", "sort_lds_kernel", "scatter")):
        print(k[:60], json.dumps(v))
PY

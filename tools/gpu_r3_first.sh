#!/bin/bash
# round 3, first call: LDS-atomic microbench (new kinds) + baseline bench on this box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in 8 17 18 19 20 21; do timeout 120 tools/micro/bin/valu_issue $k | grep -v device; done > gpurun_out/r03_lds_atomic.jsonl 2>&1
cat gpurun_out/r03_lds_atomic.jsonl | cut -c1-260
timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step > gpurun_out/r03_base.log 2> gpurun_out/r03_base.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_base.log").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()})
PY

#!/bin/bash
# round 6, first GPU call: the view-batched rasterizer -- its tests, the suites the prune touched, the bench line with `batched_views`
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_views.py tests/test_abi_and_host.py tests/test_gpu_round5.py -x -q 2>&1 | tail -15 ) > gpurun_out/r6a_tests.log
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step 2> gpurun_out/r6a_bench.err | tail -1 ) > gpurun_out/r6a_bench.json
for k in 2 8; do ( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --batch-views $k 2> gpurun_out/r6a_bench_k$k.err | tail -1 ) > gpurun_out/r6a_bench_k$k.json; done
cat gpurun_out/r6a_tests.log
python - <<'PY'
import json
for n in ("r6a_bench", "r6a_bench_k2", "r6a_bench_k8"):
    try:
        j = json.load(open(f"gpurun_out/{n}.json"))
    except Exception as e:
        print(n, "no json", e); continue
    print(n, j["value"], j["ms_per_step"], {k: v["ms"] for k, v in j["kernels"].items()}, j["roofline"]["frac"])
    b = j.get("batched_views")
    if b and "error" not in b:
        print("   batched", b["views"], b["value"], b["ms_per_view"], {k: v["ms_per_view"] for k, v in b["kernels"].items()}, b["roofline"])
    else:
        print("   batched", b)
PY
tail -5 gpurun_out/r6a_bench.err

#!/bin/bash
# Evidence pass of a round: GPU tests, bench lines (headline with live PMC collection, other configs, the view-batched sweep, the
# field networks), rocprofv3 kernel-trace stats of the headline command, counters of the batched kernels.  Everything lands in
# gpurun_out/; tools/collect_profiles.py copies the judged summaries into profiles/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
tail -1 gpurun_out/pytest_gpu.log
timeout 1500 python bench.py --pmc --init-timing > gpurun_out/bench.log 2> gpurun_out/bench.err
cp profiles/pmc_C3.json gpurun_out/pmc_C3.json 2>/dev/null; rm -f profiles/pmc_C3.json
cut -c1-400 gpurun_out/bench.log
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --batch-views 0 > "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.log" 2> "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.err" )
find gpurun_out/prof -name "*_kernel_trace.csv" -size +20M -delete
for wl in C1 C2 C4; do
  timeout 600 python bench.py --workload $wl --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/bench_$wl.log 2> gpurun_out/bench_$wl.err
done
timeout 900 python bench.py --workload C5 --steps 20 --warmup 8 --no-cpu-baseline --no-train-step --init-timing > gpurun_out/bench_C5.log 2> gpurun_out/bench_C5.err
tail -c 300 gpurun_out/bench_C5.log
# the view-batched regime at other k (the default line carries k = 4)
for k in 2 8; do
  timeout 600 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-train-step --batch-views $k 2> gpurun_out/bench_C3_views$k.err | grep "^{" > gpurun_out/bench_C3_views$k.log
done
# sensitivity: larger splats (D/P 10-30, mid / big sort paths hot) and a camera whose subject fills the frame; the Gaussians' index order
for sm in 2 4; do
  timeout 600 python bench.py --scale-mult $sm --steps 20 --warmup 8 --no-cpu-baseline --no-train-step > gpurun_out/bench_C3_s$sm.log 2> gpurun_out/bench_C3_s$sm.err
done
timeout 600 python bench.py --fill 1.7 --steps 20 --warmup 8 --no-cpu-baseline --no-train-step > gpurun_out/bench_C3_fill.log 2> gpurun_out/bench_C3_fill.err
for o in random morton; do
  timeout 600 python bench.py --gaussian-order $o --steps 20 --warmup 8 --no-cpu-baseline --no-train-step 2> gpurun_out/bench_C3_$o.err | grep "^{" > gpurun_out/bench_C3_$o.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_C*.log")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        b = d.get("batched_views") or {}
        print(f, d["value"], d["ms_per_step"], d["config"]["duplicates_D"], {k: v["ms"] for k, v in d["kernels"].items()},
              "| batched", b.get("views"), b.get("value"), b.get("ms_per_view"), (b.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
for wl in C3 C4; do
  timeout 600 python bench.py --field-mlp --workload $wl --steps 50 --warmup 10 > gpurun_out/bench_field_mlp_$wl.log 2> gpurun_out/bench_field_mlp_$wl.err
done
bash tools/prof_step.sh color > gpurun_out/prof_step_color_summary.txt 2>&1
bash tools/prof_mlp.sh > gpurun_out/prof_mlp_summary.txt 2>&1
# BASELINE config 4 entry point at N = 1 (the N > 1 launches are the driver's) and the camera-sharded step's per-rank cost
timeout 900 python bench.py --train-step color --steps 100 --warmup 10 2> gpurun_out/bench_color_train.err | grep "^{" > gpurun_out/bench_color_train.log
for k in 1 2 4; do
  timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --force-cut --views-per-rank $k 2>> gpurun_out/force_cut.err | grep "^{" > gpurun_out/force_cut_C3_k$k.log
done
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --force-cut --views-per-rank 4 --sequential-views 2>> gpurun_out/force_cut.err | grep "^{" > gpurun_out/force_cut_C3_k4_sequential.log
timeout 600 python bench.py --workload C4 --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --force-cut 2>> gpurun_out/force_cut.err | grep "^{" > gpurun_out/force_cut_C4_k1.log
# the batched compositing kernels by rocprofv3: per-kernel durations (k = 4) and their counters (separate --pmc passes)
rm -rf gpurun_out/prof_views && mkdir -p gpurun_out/prof_views
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_views" -o views -- python "$GRAFT_REPO_ROOT/tools/batched_step.py" 4 C3 20 > "$GRAFT_REPO_ROOT/gpurun_out/prof_views/log.txt" 2>&1 )
find gpurun_out/prof_views -name "*_kernel_trace.csv" -delete
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "grbm:GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/pmc_views_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "composite|preprocess|tile_" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_views_$name" -o pmc -- python "$GRAFT_REPO_ROOT/tools/batched_step.py" 4 C3 4 > /dev/null 2>&1 )
done
python - <<'PY'
import collections, csv, glob, json
out = {}
for name in ("fetch", "write", "sq", "grbm"):
    f = glob.glob(f"gpurun_out/pmc_views_{name}/**/pmc_counter_collection.csv", recursive=True)
    if not f:
        continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        out.setdefault(k, {}).update({c: round(sum(x) / len(x), 1) for c, x in v.items()})
out["_meta"] = {"command": "tools/batched_step.py 4 C3 (k = 4 cameras of the C3 pose in one grid per stage), separate rocprofv3 --pmc passes",
                "traffic": "(2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, MI355X_MICROARCH.md"}
json.dump(out, open("gpurun_out/pmc_C3_views4.json", "w"), indent=1)
for k, v in out.items():
    if "composite" in k:
        print(k[:50], {c: v.get(c) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")})
PY
bash tools/gpu_diag_json.sh C3 > gpurun_out/diag_json.log 2>&1

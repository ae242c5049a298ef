#!/bin/bash
# End-of-round evidence: tests, PMC passes (fetch/write/sq), kernel trace, bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
pmc() {  # name, counters...
  name=$1; shift
  rm -rf gpurun_out/pmc_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "d3ga" --output-format csv \
      -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 3 --no-cpu-baseline --no-train-step --no-stage-events --no-graph \
      > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log" 2>&1 )
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-train-step > "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.log" 2> "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.err" )
find gpurun_out/prof -name "*_kernel_trace.csv" -size +20M -delete
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err
cat gpurun_out/pytest_gpu.log | tail -2; cat gpurun_out/bench.log | cut -c1-600
for wl in C1 C2 C4 C5; do
  timeout 600 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-train-step > gpurun_out/bench_$wl.log 2> gpurun_out/bench_$wl.err
done
tail -c 300 gpurun_out/bench_C5.log
for wl in C3 C4; do
  timeout 600 python bench.py --field-mlp --workload $wl --steps 50 --warmup 10 > gpurun_out/bench_field_mlp_$wl.log 2> gpurun_out/bench_field_mlp_$wl.err
done
# per-kernel summaries of the field networks (CanonicalField fwd+bwd) and of the reference's training step with colour MLP
bash tools/prof_mlp.sh > gpurun_out/prof_mlp_summary.txt 2>&1
bash tools/prof_step.sh color > gpurun_out/prof_step_color_summary.txt 2>&1
cut -c1-400 gpurun_out/bench_field_mlp_C3.log

#!/bin/bash
# PMC counter passes (each in its own run, kernel-trace only) for the d3ga kernels of a short bench.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  rm -rf gpurun_out/pmc_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "d3ga" --output-format csv \
      -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 3 --no-cpu-baseline --no-stage-events \
      > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log" 2>&1 )
  find gpurun_out/pmc_$name -name "*.csv" | head -3
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run grbm GRBM_GUI_ACTIVE
ls -la gpurun_out/pmc_*/ | head -30

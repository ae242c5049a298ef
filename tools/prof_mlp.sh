#!/bin/bash
# per-kernel times of CanonicalField fwd+bwd (tools/time_mlp.py) under rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_mlp && mkdir -p gpurun_out/prof_mlp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_mlp" -o mlp -- python "$GRAFT_REPO_ROOT/tools/time_mlp.py" > "$GRAFT_REPO_ROOT/gpurun_out/prof_mlp/log.txt" 2>&1 )
f=$(find gpurun_out/prof_mlp -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
find gpurun_out/prof_mlp -name "*_kernel_trace.csv" -delete

#!/bin/bash
# SSIM: parity (loss tests + fuzz) and kernel-level times of the tiled (D3GA_SSIM_IMPL=0) and marching (1) kernels at 1080p
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( D3GA_LOSS_FUZZ_N=40 timeout 1200 python -m pytest tests -m gpu -q -x -k "loss or ssim" 2>&1 | tail -2 )
for impl in 0 1; do
  rm -rf gpurun_out/ks_ssim$impl
  ( cd /tmp && D3GA_SSIM_IMPL=$impl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/ks_ssim$impl" -o ks -- python "$GRAFT_REPO_ROOT/tools/time_ssim.py" > /dev/null 2>&1 )
  python - $impl <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/ks_ssim{sys.argv[1]}/**/ks_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "ssim" in r["Name"]:
        print("impl", sys.argv[1], r["Name"][:60], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2))
PY
  find gpurun_out/ks_ssim$impl -name "*_kernel_trace.csv" -delete
done

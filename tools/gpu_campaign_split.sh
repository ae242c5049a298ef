#!/bin/bash
# rasterizer fuzz campaigns with the split backward: the default count and EVERY tile split (the R = 2 walk on all lists)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/campaign_split.log
run() { name=$1; shift; echo "== $name: $*" >> gpurun_out/campaign_split.log; ( env "$@" 2>&1 | tail -3 ) >> gpurun_out/campaign_split.log; }
run raster_fuzz_default   D3GA_FUZZ_N=${N_RASTER:-1500} timeout 2400 python -m pytest tests -m gpu -q -k fuzz_ragged
run raster_fuzz_split_all D3GA_BWD_SPLIT=100000 D3GA_FUZZ_N=${N_RASTER:-1500} timeout 2400 python -m pytest tests -m gpu -q -k fuzz_ragged
run raster_fuzz_C1_default D3GA_FUZZ_SCENE=C1 D3GA_FUZZ_N=${N_C1:-150} timeout 1800 python -m pytest tests -m gpu -q -k fuzz_ragged
run raster_fuzz_C1_split_all D3GA_BWD_SPLIT=100000 D3GA_FUZZ_SCENE=C1 D3GA_FUZZ_N=${N_C1:-150} timeout 1800 python -m pytest tests -m gpu -q -k fuzz_ragged
run pair_fuzz_default     D3GA_PAIR_FUZZ_N=${N_PAIR:-300} timeout 1200 python -m pytest tests -m gpu -q -k pair_fuzz
run pair_fuzz_split_all   D3GA_BWD_SPLIT=100000 D3GA_PAIR_FUZZ_N=${N_PAIR:-300} timeout 1200 python -m pytest tests -m gpu -q -k pair_fuzz
run suite_split_all       D3GA_BWD_SPLIT=100000 timeout 1500 python -m pytest tests -m gpu -q
run suite_split_off       D3GA_BWD_SPLIT=0 timeout 1500 python -m pytest tests -m gpu -q
cat gpurun_out/campaign_split.log

#!/bin/bash
# The fuzz campaigns of docs/LOG.md sec. 5 (each is a committed test with a larger seed range); results in gpurun_out/campaigns.log
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/campaigns.log
run() { name=$1; shift; echo "== $name: $*" >> gpurun_out/campaigns.log; ( env "$@" 2>&1 | tail -4 ) >> gpurun_out/campaigns.log; }
run raster_fuzz   D3GA_FUZZ_N=${N_RASTER:-3000} timeout 2400 python -m pytest tests -m gpu -q -k fuzz_ragged
run raster_fuzz_C1 D3GA_FUZZ_SCENE=C1 D3GA_FUZZ_N=${N_C1:-300} timeout 1800 python -m pytest tests -m gpu -q -k fuzz_ragged
run pair_fuzz     D3GA_PAIR_FUZZ_N=${N_PAIR:-400} timeout 1200 python -m pytest tests -m gpu -q -k pair_fuzz
run chain_fuzz    D3GA_CHAIN_FUZZ_N=${N_CHAIN:-3000} timeout 1800 python -m pytest tests -m gpu -q -k chain_fuzz
run fused_fuzz    D3GA_FUSED_FUZZ_N=${N_FUSED:-600} timeout 1800 python -m pytest tests -m gpu -q -k fused_trunk_fuzz
run deform_fuzz   D3GA_DEFORM_FUZZ_N=${N_DEFORM:-2000} timeout 1500 python -m pytest tests -m gpu -q -k deform_fuzz
run shard_fuzz    D3GA_SHARD_FUZZ_N=${N_SHARD:-300} timeout 1800 python -m pytest tests -m gpu -q -k exchange_at_the_cut
run loss_fuzz     D3GA_LOSS_FUZZ_N=${N_LOSS:-400} timeout 1200 python -m pytest tests -m gpu -q -k losses_fuzz
run views_fuzz    D3GA_VIEWS_FUZZ_N=${N_VIEWS:-1000} timeout 1500 python -m pytest tests -m gpu -q -k views_fuzz
run lbs_fuzz      D3GA_LBS_FUZZ_N=${N_LBS:-500} timeout 900 python -m pytest tests -m gpu -q -k lbs_fuzz
run fem_fuzz      D3GA_FEM_FUZZ_N=${N_FEM:-600} timeout 1200 python -m pytest tests -m gpu -q -k fem_energy_fuzz
run init_fuzz     D3GA_INIT_FUZZ_N=${N_INIT:-400} timeout 1200 python -m pytest tests -m gpu -q -k init_helpers_fuzz
cat gpurun_out/campaigns.log

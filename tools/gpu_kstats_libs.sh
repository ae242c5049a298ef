#!/bin/bash
# per-kernel averages of the headline bench for the shipped library and every tools/_build/libd3ga_hip_<tag>.so named in $1
cd "$GRAFT_REPO_ROOT" || exit 1
SPEC="base:"
for t in $1; do SPEC="$SPEC $t:D3GA_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libd3ga_hip_$t.so,D3GA_ALLOW_ABLATION=1"; done
bash tools/gpu_kstats.sh "$SPEC" 2>&1 | grep -E "^==|composite|cull|sort|scan|scatter|preprocess_kernel"

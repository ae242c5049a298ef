#!/bin/bash
# GPU tests selected by a -k expression ($1), -s output filtered to the lines the tests print themselves
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x -s -k "$1" 2>&1 | grep -E "^\[|passed|failed|^E  |Error|assert" | cut -c1-400 | tail -40

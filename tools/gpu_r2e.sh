#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2e
mkdir -p $O
timeout -k 10 600 python tools/mv_debug.py > $O/mv_debug.log 2>&1
tail -60 $O/mv_debug.log

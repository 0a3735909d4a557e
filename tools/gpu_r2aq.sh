#!/bin/bash
# round 2, GPU call AQ: API-level robustness sweep (tools/sweep_api.py)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2aq
mkdir -p $O
timeout -k 10 1000 python tools/sweep_api.py > $O/sweep.log 2> $O/sweep.err; echo "rc=$?"
cat $O/sweep.log | cut -c1-260; grep -i "error" $O/sweep.err | tail -3
echo ALL DONE

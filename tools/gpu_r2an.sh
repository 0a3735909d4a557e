#!/bin/bash
# round 2, GPU call AN: where the host time of Initialize goes at 256^3 (sub-phase timers)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2an
mkdir -p $O
nproc > $O/nproc.txt; cat $O/nproc.txt
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python tools/init_profile.py 256 2 gpu > $O/init256.log 2>&1; rc=$?
grep -v "class:" $O/init256.log | tail -24
echo ALL DONE

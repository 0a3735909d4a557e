#!/bin/bash
# round 2, GPU call AP: Initialize at 256^3 after reserving the per-subdomain pattern arrays
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ap
mkdir -p $O
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python tools/init_profile.py 256 2 gpu > $O/init256.log 2>&1
grep "^Initialize\|classes /" $O/init256.log | cut -c1-120
echo ALL DONE

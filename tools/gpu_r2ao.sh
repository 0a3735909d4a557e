#!/bin/bash
# round 2, GPU call AO: Initialize at 256^3 with 16 / 32 / 64 host threads
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ao
mkdir -p $O
for t in 16 32 64; do
  HYMLS_MI_HOST_THREADS=$t HYMLS_MI_VERBOSE=1 timeout -k 10 600 python tools/init_profile.py 256 2 gpu > $O/init256_t$t.log 2>&1
  echo "threads $t: $(grep '^Initialize' $O/init256_t$t.log)"
  grep "classes / patterns\|pull lists\|contributions\|A12/A21\|partition  " $O/init256_t$t.log | cut -c20-100 | tr '\n' ';'; echo
done
echo ALL DONE

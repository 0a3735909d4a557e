#!/bin/bash
# round 2, GPU call AD: where does the pivot-free last-level factorisation of Darcy problems grow?
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ad
mkdir -p $O
for cfg in "32 16 1" "64 16 1" "96 16 1" "128 16 1" "64 8 1" "128 8 1" "128 8 2" "64 8 2"; do
  set -- $cfg
  HYMLS_MI_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --problem darcy --grid $1 --sx $2 --levels $3 > $O/darcy_$1_$2_$3.json 2> $O/darcy_$1_$2_$3.err
  echo "darcy grid $1 sx $2 levels $3: rc=$?"
  grep -i "growth\|hymls_mi error" $O/darcy_$1_$2_$3.err | sort | uniq -c | head -6
done
echo ALL DONE

#!/bin/bash
# round 2, GPU call AG: rounding-level pressure diagonals count as zeros (Darcy, separator length 16) + regression subset
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ag
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "rounding_level or darcy or compiled or skew or full_size_darcy" > $O/gpu_tests_subset.log 2>&1; rc=$?
grep "GMRES(" $O/gpu_tests_subset.log | tail -8; tail -4 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
for cfg in "96 16 1" "128 16 1"; do
  set -- $cfg
  HYMLS_MI_VERBOSE=1 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --krylov --problem darcy --grid $1 --sx $2 --levels $3 > $O/darcy_$1_$2_$3.json 2> $O/darcy_$1_$2_$3.err
  echo "darcy grid $1 sx $2 levels $3: rc=$?  $(grep -i 'coarse solver: largest' $O/darcy_$1_$2_$3.err | head -1)"
  python -c "
import json; d=json.load(open('$O/darcy_$1_$2_$3.json')); print(d['ms_per_step'], d.get('krylov'))"
done
echo ALL DONE

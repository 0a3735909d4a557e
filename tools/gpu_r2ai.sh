#!/bin/bash
# round 2, GPU call AI: two-pass separator transform for blocks beyond the fused kernel's LDS (separator length 32); sweep leftovers
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ai
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_pass or skew or compiled" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -4 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
run() {
  name=$1; shift
  HYMLS_MI_VERBOSE=1 timeout -k 10 500 python bench.py --no-cpu-baseline --steps 3 --krylov "$@" > $O/$name.json 2> $O/$name.err; rc=$?
  if [ $rc -ne 0 ]; then echo "$name FAILED rc=$rc"; grep -i "error" $O/$name.err | tail -2 | cut -c1-300; return 0; fi
  python -c "
import json; d=json.load(open('$O/$name.json')); c=d['config']; k=d.get('krylov') or {}; print('$name', 'init %.2f compute %.2f recompute %.2f  apply %.2f ms  frac %.2f' % (c['initialize_s'], c['compute_s'], c['recompute_s'], d['ms_per_step'], d['roofline']['frac']), 'its', k.get('iterations'), 'res', k.get('true_relative_residual'), [l[1] for l in c['levels']])"
}
run sx32_128_l1 --grid 128 --sx 32 --levels 1
run nvec3 --grid 128 --levels 2 --nvec 3
run darcy_sx4_64_l2 --problem darcy --grid 64 --sx 4 --levels 2
run cavity_sx16_256 --problem cavity --grid 256 --sx 16 --levels 1 --re 1000
echo ALL DONE

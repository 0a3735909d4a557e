#!/bin/bash
# round 2, GPU call AH: second robustness sweep (non-BASELINE parameter combinations)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ah
mkdir -p $O
run() {
  name=$1; shift
  HYMLS_MI_VERBOSE=1 timeout -k 10 500 python bench.py --no-cpu-baseline --steps 3 --krylov "$@" > $O/$name.json 2> $O/$name.err; rc=$?
  if [ $rc -ne 0 ]; then echo "$name FAILED rc=$rc"; grep -i "error" $O/$name.err | tail -2 | cut -c1-300; return 0; fi
  python -c "
import json; d=json.load(open('$O/$name.json')); c=d['config']; k=d.get('krylov') or {}; print('$name', 'init %.2f compute %.2f recompute %.2f  apply %.2f ms  frac %.2f' % (c['initialize_s'], c['compute_s'], c['recompute_s'], d['ms_per_step'], d['roofline']['frac']), 'its', k.get('iterations'), 'res', k.get('true_relative_residual'), [l[1] for l in c['levels']])"
}
run sx4_128_l2 --grid 128 --sx 4 --levels 2
run sx16_256_l1 --grid 256 --sx 16 --levels 1
run cavity_re5000 --problem cavity --grid 128 --levels 2 --re 5000
run darcy_sx16_256 --problem darcy --grid 256 --sx 16 --levels 1
run nvec3 --grid 128 --levels 2 --nvec 3
run levels0_32 --grid 32 --levels 0
run sx8_64_l1_cavity_re100 --problem cavity --grid 64 --levels 1 --re 100
run sx32_128_l1 --grid 128 --sx 32 --levels 1
echo ALL DONE

#!/bin/bash
# round 2, GPU call AC: robustness sweep over separator lengths / levels that are not BASELINE configurations
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ac
mkdir -p $O
run() {
  name=$1; shift
  HYMLS_MI_VERBOSE=1 timeout -k 10 500 python bench.py --no-cpu-baseline --steps 5 --krylov "$@" > $O/$name.json 2> $O/$name.err; rc=$?
  if [ $rc -ne 0 ]; then echo "$name FAILED rc=$rc"; tail -4 $O/$name.err; return 0; fi
  python -c "
import json; d=json.load(open('$O/$name.json')); c=d['config']; print('$name', 'init %.2f compute %.2f recompute %.2f  apply %.2f ms  frac %.2f' % (c['initialize_s'], c['compute_s'], c['recompute_s'], d['ms_per_step'], d['roofline']['frac']), c.get('krylov'), c['levels'])"
}
run sx16_128_l1 --grid 128 --sx 16 --levels 1
run sx16_128_l2 --grid 128 --sx 16 --levels 2
run sx4_64_l2 --grid 64 --sx 4 --levels 2
run sx4_64_l3 --grid 64 --sx 4 --levels 3
run sx8_192_l2 --grid 192 --sx 8 --levels 2
run darcy_sx16_128 --problem darcy --grid 128 --sx 16 --levels 1
run cavity_sx16_128 --problem cavity --grid 128 --sx 16 --levels 1 --re 500
echo ALL DONE

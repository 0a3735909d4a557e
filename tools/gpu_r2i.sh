#!/bin/bash
# round 2, GPU call I: setup A/B on one box: multi-workgroup factorisation threshold, smaller-LDS k_factor_level, unrolled chains
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2i
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "matches_oracle or compiled or unstable or bordered" > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for wf in 3e6 1e12 1.2e7; do
  HYMLS_MI_VERBOSE=1 HYMLS_MI_WIDE_FACTOR_FLOPS=$wf timeout -k 10 600 python bench.py --no-cpu-baseline --steps 5 > $O/bench_256_wf$wf.json 2> $O/bench_256_wf$wf.err || { tail -5 $O/bench_256_wf$wf.err; exit 11; }
  python -c "
import json; d=json.load(open('$O/bench_256_wf$wf.json')); print('wf $wf', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
  grep "compute: factor" $O/bench_256_wf$wf.err | tail -2
done
HYMLS_MI_FACTOR_PROF=1 timeout -k 10 600 python bench.py --grid 128 --levels 2 --steps 3 --no-cpu-baseline > $O/factor_prof.json 2> $O/factor_prof.err || { tail -5 $O/factor_prof.err; exit 12; }
echo ALL DONE

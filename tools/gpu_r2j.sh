#!/bin/bash
# round 2, GPU call J: same-box A/B of the level-1 factorisation variants at 256^3; reproducibility test
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "reproducible or matches_oracle or compiled" > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
run() {
  tag=$1; shift
  env "$@" HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 5 > $O/bench_256_$tag.json 2> $O/bench_256_$tag.err || { tail -5 $O/bench_256_$tag.err; exit 11; }
  python -c "
import json; d=json.load(open('$O/bench_256_$tag.json')); print('$tag', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
  grep "compute: factor" $O/bench_256_$tag.err | tail -2
}
run wide_small HYMLS_MI_WIDE_FACTOR_FLOPS=3e6
run wide_big HYMLS_MI_WIDE_FACTOR_FLOPS=3e6 HYMLS_MI_FACTOR_LDS=6144
run nowide_big HYMLS_MI_WIDE_FACTOR_FLOPS=1e12 HYMLS_MI_FACTOR_LDS=6144
run wide1e6_small HYMLS_MI_WIDE_FACTOR_FLOPS=1e6
echo ALL DONE

#!/bin/bash
# round 2, GPU call K: same-box A/B of the k_factor_level edits (library built from commit 78dd7b9 against the current one)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 5 > $O/bench_256_$tag.json 2> $O/bench_256_$tag.err || { tail -5 $O/bench_256_$tag.err; exit 11; }
  python -c "
import json; d=json.load(open('$O/bench_256_$tag.json')); print('$tag', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
  grep "compute: factor" $O/bench_256_$tag.err | tail -2
}
run old_nowide HYMLS_MI_LIBRARY=$PWD/hymls_amd/libhymls_mi_ab_old.so HYMLS_MI_WIDE_FACTOR_FLOPS=1e12
run new_nowide HYMLS_MI_WIDE_FACTOR_FLOPS=1e12
run old_wide HYMLS_MI_LIBRARY=$PWD/hymls_amd/libhymls_mi_ab_old.so
run new_wide HYMLS_MI_WIDE_FACTOR_FLOPS=3e6
echo ALL DONE

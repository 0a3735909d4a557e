#!/bin/bash
# round 2, GPU call AM: Schur GEMM inside k_factor_level on the matrix cores (A/B on one box)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2am
mkdir -p $O
HYMLS_MI_FACTOR_MFMA=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "apply_inverse_matches or skew or compiled or reproducible or big_front" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -4 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
for v in 0 1 0 1; do
  HYMLS_MI_FACTOR_MFMA=$v HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 3 > $O/bench_256_m$v.json 2> $O/bench_256_m$v.err || { tail -5 $O/bench_256_m$v.err; exit 11; }
  python -c "
import json; d=json.load(open('$O/bench_256_m$v.json')); print('256 mfma $v', d['config']['compute_s'], d['config']['recompute_s'])"
  grep "level 0 compute: factor" $O/bench_256_m$v.err | tail -1
done
echo ALL DONE

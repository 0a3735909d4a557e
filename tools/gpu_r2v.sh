#!/bin/bash
# round 2, GPU call V: single-vector fused solve, 8 panel entries in flight per thread at six workgroups per CU (A/B on one box)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2v
mkdir -p $O
HYMLS_MI_FUSED_DEPTH=8 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multivector or apply_inverse_matches or skew" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -5 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
for d in 4 8 4 8; do
  HYMLS_MI_FUSED_DEPTH=$d timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_256_d$d.json 2> $O/bench_256_d$d.err || exit 15
  python -c "
import json; d=json.load(open('$O/bench_256_d$d.json')); print('depth $d', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
done
echo ALL DONE

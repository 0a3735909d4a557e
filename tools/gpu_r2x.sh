#!/bin/bash
# round 2, GPU call X: coarser levels: classes balanced over the side streams by work; 4 / 6 / 8 side streams
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2x
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reproducible or skew or compiled or big_front" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -5 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
for v in 4 6 8; do
  HYMLS_MI_SIDE_STREAMS=$v HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 5 > $O/bench_256_ss$v.json 2> $O/bench_256_ss$v.err || { tail -5 $O/bench_256_ss$v.err; exit 11; }
  python -c "
import json; d=json.load(open('$O/bench_256_ss$v.json')); print('256 side streams $v', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'], d['config']['hbm_used_gib_rank0'])"
  grep "level 1 compute: factor\|level 1 compute: pull" $O/bench_256_ss$v.err | tail -2
done
echo ALL DONE

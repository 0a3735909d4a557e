#!/bin/bash
# round 2, GPU call AA: last-level solver: wide fronts of a tree level on the side streams (configs[1]); A/B against one stream
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2aa
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reproducible or skew or compiled or big_front or recompute or apply_inverse_matches or values" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -5 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
for v in spread one; do
  if [ $v = one ]; then export HYMLS_MI_SIDE_STREAMS=1; fi
  HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --grid 128 --levels 1 --no-cpu-baseline --steps 5 > $O/bench_128_l1_$v.json 2> $O/bench_128_l1_$v.err || exit 12
  python -c "
import json; d=json.load(open('$O/bench_128_l1_$v.json')); print('128 L1 $v', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
  grep "coarse solver   \|compute: coarse" $O/bench_128_l1_$v.err | tail -3
done
unset HYMLS_MI_SIDE_STREAMS
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 5 > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 11; }
python -c "
import json; d=json.load(open('$O/bench_256.json')); print('256', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'], d['config']['hbm_used_gib_rank0'])"
grep "compute: factor" $O/bench_256.err | tail -2
echo ALL DONE

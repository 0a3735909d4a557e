#!/bin/bash
# round 2, GPU call S: blocked pivot-piece kernel (harness: against the scalar kernel), GEMM C preload A/B, 128^3 L1 trace
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2s
mkdir -p $O
timeout -k 10 120 ./tools/pivot_check.bin > $O/pivot_check.log 2>&1; rc=$?
cat $O/pivot_check.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "skew or compiled or big_front or reproducible or unstable" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -5 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
for v in blocked scalar; do
  if [ $v = scalar ]; then export HYMLS_MI_PIVOT_BLOCKED=0; fi
  HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --grid 128 --levels 1 --no-cpu-baseline --steps 5 > $O/bench_128_l1_$v.json 2> $O/bench_128_l1_$v.err || exit 12
  python -c "
import json; d=json.load(open('$O/bench_128_l1_$v.json')); print('128 L1 $v', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
done
unset HYMLS_MI_PIVOT_BLOCKED
for v in 0 1; do
  HYMLS_MI_GEMM_PRELOAD=$v HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 5 > $O/bench_256_pre$v.json 2> $O/bench_256_pre$v.err || { tail -5 $O/bench_256_pre$v.err; exit 11; }
  python -c "
import json; d=json.load(open('$O/bench_256_pre$v.json')); print('256 preload $v', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
  grep "compute: factor" $O/bench_256_pre$v.err | tail -2
done
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/trace128 -o run --output-format csv -- python3 bench.py --grid 128 --levels 1 --no-cpu-baseline --steps 10 > $O/trace128.log 2>&1 || { tail -20 $O/trace128.log; exit 17; }
head -14 $O/trace128/run_kernel_stats.csv | cut -c1-200
echo ALL DONE

#!/bin/bash
# round 2, GPU call P: blocked Gauss-Jordan for the large separator blocks, 1024-thread pivot pieces
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2p
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "separator_block" > $O/invert_tests.log 2>&1; rc=$?
tail -15 $O/invert_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "skew or compiled or big_front or reproducible or unstable or full_size_properties_128" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -5 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
HYMLS_MI_VERBOSE=1 HYMLS_MI_BLOCK_STATS=1 timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 11; }
grep "separator blocks" $O/bench_256.err | head -3 | cut -c1-600
python -c "
import json; d=json.load(open('$O/bench_256.json')); print('256', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/trace256 -o run --output-format csv -- python3 bench.py --no-cpu-baseline --steps 10 > $O/trace256.log 2>&1 || { tail -20 $O/trace256.log; exit 17; }
head -12 $O/trace256/run_kernel_stats.csv | cut -c1-200
timeout -k 10 400 python bench.py --grid 128 --levels 2 --sx 8 --no-cpu-baseline > $O/bench_128_l2.json 2> $O/bench_128_l2.err || exit 15
python -c "
import json; d=json.load(open('$O/bench_128_l2.json')); print('128 L2', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
echo ALL DONE

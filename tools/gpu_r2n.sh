#!/bin/bash
# round 2, GPU call N: one barrier per step in the LDS LU / inversions: parity, setup times, k_big_pivot average
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2n
mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 5 > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 11; }
python -c "
import json; d=json.load(open('$O/bench_256.json')); print('256', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
grep "compute: " $O/bench_256.err | tail -9
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/trace128 -o run --output-format csv -- python3 bench.py --no-cpu-baseline --grid 128 --levels 2 --steps 5 > $O/trace128.log 2>&1 || { tail -20 $O/trace128.log; exit 14; }
head -8 $O/trace128/run_kernel_stats.csv
timeout -k 10 400 python bench.py --grid 128 --levels 1 --no-cpu-baseline > $O/bench_128_l1.json 2> $O/bench_128_l1.err || exit 15
python -c "
import json; d=json.load(open('$O/bench_128_l1.json')); print('128 L1', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
echo ALL DONE

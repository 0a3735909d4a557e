#!/bin/bash
# round 2, GPU call R: k_factor_level at four workgroups per CU (was one: 248 VGPRs), division-free index loops
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2r
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "skew or compiled or big_front or reproducible or unstable or apply_inverse_matches" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -5 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --grid 128 --levels 1 --no-cpu-baseline > $O/bench_128_l1.json 2> $O/bench_128_l1.err || exit 12
python -c "
import json; d=json.load(open('$O/bench_128_l1.json')); print('128 L1', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 11; }
python -c "
import json; d=json.load(open('$O/bench_256.json')); print('256', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
grep "compute:" $O/bench_256.err | tail -12
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/trace256 -o run --output-format csv -- python3 bench.py --no-cpu-baseline --steps 10 > $O/trace256.log 2>&1 || { tail -20 $O/trace256.log; exit 17; }
head -12 $O/trace256/run_kernel_stats.csv | cut -c1-200
echo ALL DONE

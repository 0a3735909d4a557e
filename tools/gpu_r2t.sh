#!/bin/bash
# round 2, GPU call T: coarse plan kept across Compute calls (configs[1] recompute), pivot kernel rule, phase ticks at 4 WGs/CU
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2t
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "recompute or skew or compiled or big_front or reproducible or unstable or bordered" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -5 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --grid 128 --levels 1 --no-cpu-baseline > $O/bench_128_l1.json 2> $O/bench_128_l1.err || exit 12
python -c "
import json; d=json.load(open('$O/bench_128_l1.json')); print('128 L1', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
grep "compute:" $O/bench_128_l1.err | tail -6
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 11; }
python -c "
import json; d=json.load(open('$O/bench_256.json')); print('256', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
grep "compute:" $O/bench_256.err | tail -12
HYMLS_MI_FACTOR_PROF=1 HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --grid 128 --levels 1 --no-cpu-baseline --steps 3 > $O/prof_128_l1.json 2> $O/prof_128_l1.err || { tail -5 $O/prof_128_l1.err; exit 13; }
grep -c k_factor_level $O/prof_128_l1.err
echo ALL DONE

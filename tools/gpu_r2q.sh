#!/bin/bash
# round 2, GPU call Q: per-phase ticks of k_factor_level on the level-1 fronts (128^3, sx=8), then the whole GPU suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2q
mkdir -p $O
HYMLS_MI_FACTOR_PROF=1 HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --grid 128 --levels 1 --no-cpu-baseline --steps 3 > $O/prof_128_l1.json 2> $O/prof_128_l1.err || { tail -5 $O/prof_128_l1.err; exit 11; }
grep -c k_factor_level $O/prof_128_l1.err
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --grid 128 --levels 1 --no-cpu-baseline > $O/bench_128_l1.json 2> $O/bench_128_l1.err || exit 12
python -c "
import json; d=json.load(open('$O/bench_128_l1.json')); print('128 L1', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
echo ALL DONE

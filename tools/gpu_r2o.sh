#!/bin/bash
# round 2, GPU call O: sharded coarse ordering = one-rank ordering; growth factor report; pivot kernel edits; final bench lines
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2o
mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --force-sharded --no-cpu-baseline > $O/bench_256_forced_rccl.json 2> $O/bench_256_forced_rccl.err || { tail -12 $O/bench_256_forced_rccl.err; exit 16; }
grep -i "growth" $O/bench_256_forced_rccl.err | tail -6
head -c 250 $O/bench_256_forced_rccl.json; echo
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 11; }
grep -i "growth" $O/bench_256.err | tail -6
python -c "
import json; d=json.load(open('$O/bench_256.json')); print('256', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/trace256 -o run --output-format csv -- python3 bench.py --no-cpu-baseline --steps 10 > $O/trace256.log 2>&1 || { tail -20 $O/trace256.log; exit 17; }
head -6 $O/trace256/run_kernel_stats.csv
timeout -k 10 400 python bench.py --grid 128 --levels 1 --no-cpu-baseline > $O/bench_128_l1.json 2> $O/bench_128_l1.err || exit 15
python -c "
import json; d=json.load(open('$O/bench_128_l1.json')); print('128 L1', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
echo ALL DONE

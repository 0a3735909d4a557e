#!/bin/bash
# round 2, GPU call ZZ (same as Y, after the stream / plan-reuse changes): the whole GPU suite and every bench line / kernel trace that goes into profiles/ (final code)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2zz
mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 11; }
head -c 250 $O/bench_256.json; echo
timeout -k 10 400 python bench.py --grid 128 --levels 1 > $O/bench_stokes128_l1.json 2> $O/bench_stokes128_l1.err || exit 12
timeout -k 10 400 python bench.py --problem cavity --grid 128 --levels 2 > $O/bench_cavity128.json 2> $O/bench_cavity128.err || exit 13
timeout -k 10 600 python bench.py --problem darcy --grid 256 --levels 2 --no-cpu-baseline > $O/bench_darcy256.json 2> $O/bench_darcy256.err || exit 14
timeout -k 10 600 python bench.py --nvec 4 --steps 10 --no-cpu-baseline > $O/bench_256_nvec4.json 2> $O/bench_256_nvec4.err || exit 15
timeout -k 10 600 python bench.py --force-sharded --no-cpu-baseline > $O/bench_256_forced_rccl.json 2> $O/bench_256_forced_rccl.err || exit 16
echo benches done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); c=d['config']; print('$f'.split('/')[-1], c.get('initialize_s'), c.get('compute_s'), c.get('recompute_s'), d['ms_per_step'], d['roofline']['frac'])"; done
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/trace256 -o run --output-format csv -- python3 bench.py --no-cpu-baseline --steps 10 > $O/trace256.log 2>&1 || { tail -20 $O/trace256.log; exit 17; }
head -8 $O/trace256/run_kernel_stats.csv | cut -c1-180
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/trace128 -o run --output-format csv -- python3 bench.py --grid 128 --levels 1 --no-cpu-baseline --steps 10 > $O/trace128.log 2>&1 || { tail -20 $O/trace128.log; exit 18; }
head -8 $O/trace128/run_kernel_stats.csv | cut -c1-180
echo ALL DONE

#!/bin/bash
# round 2, GPU call C: merged task kernels for the coarse direct solver; value parity against the compiled CPU oracle; full bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?
tail -15 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --grid 128 --levels 1 --no-cpu-baseline > $O/bench_stokes128_l1.json 2> $O/bench_stokes128_l1.err || { tail -5 $O/bench_stokes128_l1.err; exit 11; }
timeout -k 10 600 python bench.py > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 12; }
cat $O/bench_stokes128_l1.json | head -c 400; echo; cat $O/bench_256.json | head -c 300; echo
echo ALL DONE

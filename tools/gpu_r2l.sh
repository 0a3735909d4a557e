#!/bin/bash
# round 2, GPU call L: kernel trace of the current code (256^3 default run), full GPU test suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/trace256 -o run --output-format csv -- python3 bench.py --no-cpu-baseline --steps 10 > $O/trace256.log 2>&1 || { tail -20 $O/trace256.log; exit 14; }
head -25 $O/trace256/run_kernel_stats.csv
echo ALL DONE

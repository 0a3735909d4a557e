#!/bin/bash
# round 2, GPU call F: bench lines (default with CPU baseline, nvec 2/4/8, configs[1]) after the host-path and multi-vector changes
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2f
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "matches_oracle or multivector or smoke or lifecycle" > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 11; }
head -c 300 $O/bench_256.json; echo
for nv in 2 4 8; do
  timeout -k 10 600 python bench.py --no-cpu-baseline --nvec $nv --steps 10 > $O/bench_256_nvec$nv.json 2> $O/bench_256_nvec$nv.err || { tail -5 $O/bench_256_nvec$nv.err; exit 12; }
  head -c 330 $O/bench_256_nvec$nv.json; echo
done
timeout -k 10 400 python bench.py --grid 128 --levels 1 --no-cpu-baseline > $O/bench_stokes128_l1.json 2> $O/bench_stokes128_l1.err || { tail -5 $O/bench_stokes128_l1.err; exit 13; }
head -c 330 $O/bench_stokes128_l1.json; echo
echo ALL DONE

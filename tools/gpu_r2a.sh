#!/bin/bash
# round 2, GPU call A: parity tests, bench lines of configs[1]/[3]/[4], PMC traffic of k_interior_fused on the benched config
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --grid 128 --levels 1 --no-cpu-baseline > $O/bench_stokes128_l1.json 2> $O/bench_stokes128_l1.err || exit 11
timeout -k 10 300 python bench.py --problem cavity --grid 128 --levels 2 --no-cpu-baseline > $O/bench_cavity128.json 2> $O/bench_cavity128.err || exit 12
timeout -k 10 400 python bench.py --problem darcy --grid 256 --levels 2 --no-cpu-baseline > $O/bench_darcy256.json 2> $O/bench_darcy256.err || exit 13
echo benches done
for C in FETCH_SIZE WRITE_SIZE; do
  PMC_DRIVER_OUT=$O timeout -k 10 600 rocprofv3 --pmc $C --kernel-include-regex 'k_interior_fused|k_axpby' --kernel-trace -d $O/pmc256_$C -o run -- python3 tools/pmc_driver.py 256 8 2 3 > $O/pmc256_$C.log 2>&1 || { echo "pmc $C failed"; tail -30 $O/pmc256_$C.log; exit 14; }
  tail -3 $O/pmc256_$C.log
done
echo ALL DONE

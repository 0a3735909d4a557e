#!/bin/bash
# round 2, GPU call G: where k_factor_level spends its time; panel column tile 256 vs 1024 for the coarse solver of configs[1]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2g
mkdir -p $O
HYMLS_MI_FACTOR_PROF=1 timeout -k 10 600 python bench.py --grid 128 --levels 2 --steps 3 --no-cpu-baseline > $O/factor_prof.json 2> $O/factor_prof.err || { tail -5 $O/factor_prof.err; exit 11; }
grep -c k_factor_level $O/factor_prof.err
HYMLS_MI_SOLVE_KT=256 timeout -k 10 400 python bench.py --grid 128 --levels 1 --no-cpu-baseline > $O/bench_128_l1_kt256.json 2> $O/bench_128_l1_kt256.err || { tail -5 $O/bench_128_l1_kt256.err; exit 12; }
head -c 300 $O/bench_128_l1_kt256.json; echo
HYMLS_MI_SOLVE_KT=256 timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "matches_oracle or full_size_properties_128 or compiled" > $O/gpu_tests_kt256.log 2>&1; tail -3 $O/gpu_tests_kt256.log
echo ALL DONE

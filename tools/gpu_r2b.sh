#!/bin/bash
# round 2, GPU call B: per-handle contexts, built-in RCCL transport (single rank, forced), default bench + kernel trace
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "rccl or two_live or adapter or capi or smoke or sharded" > $O/gpu_tests.log 2>&1; rc=$?
tail -15 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 11; }
timeout -k 10 400 python bench.py --no-cpu-baseline --force-sharded > $O/bench_256_forced_rccl.json 2> $O/bench_256_forced_rccl.err || { tail -20 $O/bench_256_forced_rccl.err; exit 12; }
timeout -k 10 400 python bench.py --no-cpu-baseline --force-sharded --transport torch > $O/bench_256_forced_torch.json 2> $O/bench_256_forced_torch.err || { tail -20 $O/bench_256_forced_torch.err; exit 13; }
echo benches done
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/trace256 -o run --output-format csv -- python3 bench.py --no-cpu-baseline --steps 10 > $O/trace256.log 2>&1 || { tail -20 $O/trace256.log; exit 14; }
ls $O/trace256
echo ALL DONE

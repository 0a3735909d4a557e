#!/bin/bash
# round 2, GPU call H: multi-workgroup factorisation of the fronts with large Schur updates: parity + setup times
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2h
mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for wf in 3e6 1e6 1e7 1e12; do
  HYMLS_MI_WIDE_FACTOR_FLOPS=$wf timeout -k 10 600 python bench.py --grid 128 --levels 2 --steps 5 --no-cpu-baseline > $O/bench_128_wf$wf.json 2> $O/bench_128_wf$wf.err || { tail -5 $O/bench_128_wf$wf.err; exit 11; }
  python -c "
import json; d=json.load(open('$O/bench_128_wf$wf.json')); print('wf $wf', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
done
HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 12; }
python -c "
import json; d=json.load(open('$O/bench_256.json')); print('256', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'])"
echo ALL DONE

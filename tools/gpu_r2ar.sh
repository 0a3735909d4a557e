#!/bin/bash
# round 2, GPU call AR: the last-level solve recorded as a HIP graph (configs[1]: ~100 launches per solve); A/B on one box
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ar
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "apply_inverse_matches or skew or compiled or big_front or recompute or bordered or multivector or merged" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -4 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
for v in graph direct graph direct; do
  if [ $v = direct ]; then export HYMLS_MI_NO_GRAPH=1; else unset HYMLS_MI_NO_GRAPH; fi
  timeout -k 10 600 python bench.py --grid 128 --levels 1 --no-cpu-baseline > $O/bench_128_l1_$v.json 2> $O/bench_128_l1_$v.err || exit 12
  python -c "
import json; d=json.load(open('$O/bench_128_l1_$v.json')); print('128 L1 $v', d['ms_per_step'], d['phase_ms'])"
done
unset HYMLS_MI_NO_GRAPH
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_256.json 2> $O/bench_256.err || exit 11
python -c "
import json; d=json.load(open('$O/bench_256.json')); print('256', d['ms_per_step'], d['phase_ms'])"
echo ALL DONE

#!/bin/bash
# round 2, GPU call U: multi-vector fused solve with deeper load prefetch (16 / 8 panel entries in flight per thread)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2u
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multivector or separator_block or recompute" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -5 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
for nv in 4 2 8; do
  timeout -k 10 600 python bench.py --nvec $nv --steps 10 --no-cpu-baseline > $O/bench_256_nvec$nv.json 2> $O/bench_256_nvec$nv.err || exit 15
  python -c "
import json; d=json.load(open('$O/bench_256_nvec$nv.json')); print('nvec $nv', d['ms_per_step'], d['ms_per_vector'], d['phase_ms'])"
done
echo ALL DONE

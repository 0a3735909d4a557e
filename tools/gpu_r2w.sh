#!/bin/bash
# round 2, GPU call W: chunks of the finest level alternate over two side streams (A/B: 0 / 2 / 4 streams)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2w
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reproducible or skew or compiled or unstable or recompute or full_size_properties_128" > $O/gpu_tests_subset.log 2>&1; rc=$?
tail -5 $O/gpu_tests_subset.log
[ $rc -eq 0 ] || exit $rc
for v in 0 2 4; do
  HYMLS_MI_CHUNK_STREAMS=$v HYMLS_MI_VERBOSE=1 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 5 > $O/bench_256_cs$v.json 2> $O/bench_256_cs$v.err || { tail -5 $O/bench_256_cs$v.err; exit 11; }
  python -c "
import json; d=json.load(open('$O/bench_256_cs$v.json')); print('256 chunk streams $v', d['config']['initialize_s'], d['config']['compute_s'], d['config']['recompute_s'], d['ms_per_step'], d['config']['hbm_used_gib_rank0'])"
  grep "level 0 compute: factor" $O/bench_256_cs$v.err | tail -2
done
echo ALL DONE

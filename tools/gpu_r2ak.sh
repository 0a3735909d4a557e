#!/bin/bash
# round 2, GPU call AK: final check of the committed tree: smoke(), the whole GPU suite, the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ak
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?
tail -3 $O/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?
tail -4 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/bench_256.json 2> $O/bench_256.err || { tail -5 $O/bench_256.err; exit 11; }
python -c "
import json; d=json.load(open('$O/bench_256.json')); c=d['config']; print('256', c['initialize_s'], c['compute_s'], c['recompute_s'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
echo ALL DONE

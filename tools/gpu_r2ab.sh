#!/bin/bash
# round 2, GPU call AB: blocked inversion for orders above 1024 (panel step in a global scratch copy)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ab
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "separator_block" --durations=5 > $O/invert_tests.log 2>&1; rc=$?
tail -14 $O/invert_tests.log
[ $rc -eq 0 ] || exit $rc
HYMLS_MI_INVERT_BLOCKED_MIN=100000 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "separator_block_inversion_gpu and (1100 or 2050)" --durations=5 > $O/invert_tests_scalar.log 2>&1; rc=$?
tail -8 $O/invert_tests_scalar.log
echo ALL DONE

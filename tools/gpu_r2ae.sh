#!/bin/bash
# round 2, GPU call AE: which kernel makes the last-level factorisation of Darcy 64^3 sx=16 grow (the same order is stable on the CPU)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2ae
mkdir -p $O
run() {
  name=$1; shift
  env "$@" HYMLS_MI_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --problem darcy --grid 64 --sx 16 --levels 1 > $O/$name.json 2> $O/$name.err
  echo "$name: rc=$?  $(grep -i 'coarse solver: largest' $O/$name.err | head -1)"
}
run default A=1
run scalar_pivot HYMLS_MI_PIVOT_BLOCKED=0
run no_wide HYMLS_MI_WIDE_FACTOR_FLOPS=1e30
run no_side HYMLS_MI_NO_SIDE_STREAMS=1
run one_stream HYMLS_MI_SIDE_STREAMS=1
run big_lds HYMLS_MI_FACTOR_LDS=6144
echo ALL DONE

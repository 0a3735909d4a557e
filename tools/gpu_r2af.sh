#!/bin/bash
# round 2, GPU call AF: dump the last-level matrix + elimination order of Darcy 64^3 sx=16 as the GPU run sees it
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2af
mkdir -p $O
HYMLS_MI_DUMP_COARSE=$O/coarse_gpu.bin HYMLS_MI_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --problem darcy --grid 64 --sx 16 --levels 1 > $O/run.json 2> $O/run.err
echo "rc=$?"; grep -i "growth" $O/run.err | head -3; ls -la $O
echo ALL DONE

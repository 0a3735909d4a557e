#!/bin/bash
# round 2, GPU call AL: sharded rehearsal sweep on one GPU (ranks share the device): other separator lengths / problems / rank grids
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2al
mkdir -p $O
run() {
  name=$1; np=$2; shift; shift
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus $np --share-gpu --backend gloo --no-cpu-baseline --steps 3 --krylov "$@" > $O/$name.json 2> $O/$name.err; rc=$?
  if [ $rc -ne 0 ]; then echo "$name FAILED rc=$rc"; grep -i "error" $O/$name.err | tail -3 | cut -c1-300; return 0; fi
  python -c "
import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); c=d['config']; k=d.get('krylov') or {}; print('$name', 'n_gpus', d['n_gpus'], 'init %.2f compute %.2f recompute %.2f  apply %.2f ms' % (c['initialize_s'], c['compute_s'], c['recompute_s'], d['ms_per_step']), 'its', k.get('iterations'), 'res', k.get('true_relative_residual'))"
}
run stokes128_sx16_n2 2 --grid 128 --sx 16 --levels 1
run darcy128_sx16_n4 4 --problem darcy --grid 128 --sx 16 --levels 1
run cavity128_n4 4 --problem cavity --grid 128 --levels 2 --re 1000
run stokes128_sx4_n2 2 --grid 128 --sx 4 --levels 2
run stokes64_n6 6 --grid 96 --levels 1
echo ALL DONE

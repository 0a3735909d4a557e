#!/bin/bash
# round 2, GPU call AJ: rehearsal of the multi-rank bench flow on ONE GPU (ranks share the device; transport through gloo)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2aj
mkdir -p $O
run() {
  name=$1; np=$2; shift; shift
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $np --share-gpu --backend gloo --no-cpu-baseline --steps 5 --krylov "$@" > $O/$name.json 2> $O/$name.err; rc=$?
  if [ $rc -ne 0 ]; then echo "$name FAILED rc=$rc"; grep -i "error" $O/$name.err | tail -3 | cut -c1-300; return 0; fi
  python -c "
import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); c=d['config']; k=d.get('krylov') or {}; print('$name', 'n_gpus', d['n_gpus'], 'init %.2f compute %.2f recompute %.2f  apply %.2f ms' % (c['initialize_s'], c['compute_s'], c['recompute_s'], d['ms_per_step']), 'its', k.get('iterations'), 'res', k.get('true_relative_residual'), c['parallelism'])"
}
run stokes128_n2 2 --grid 128 --levels 2
run stokes128_n4 4 --grid 128 --levels 2
run darcy128_n2 2 --problem darcy --grid 128 --levels 2
run stokes128_l1_n2 2 --grid 128 --levels 1
echo ALL DONE

#!/bin/bash
# per-iteration time of the in-library sharded solve with 1, 2, 4 and 8 ranks SHARING the box's one GPU (8 ranks = 4 processes of
# two rank threads: the box admits six processes on its card) (rehearsal: the device code
# is the one that runs between GPUs; the numbers say what the protocol costs, not what xGMI adds)
set -euo pipefail
cd "$(dirname "$0")/.."
for cfg in C D E; do
  for n in 1 2 4; do
    if [ "$n" = 1 ]; then
      ELPH_FORCE_DEVICE=0 python3 bench.py --mode spatial --config $cfg --steps 2000 --warmup 200 --no-cpu
    else
      ELPH_FORCE_DEVICE=0 ELPH_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --mode spatial --config $cfg --gpus $n --steps 2000 --warmup 200 --no-cpu 2>/dev/null
    fi
  done
  ELPH_FORCE_DEVICE=0 ELPH_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29608 bench.py --mode spatial --config $cfg --gpus 8 --ranks-per-proc 2 --steps 2000 --warmup 200 --no-cpu 2>/dev/null
done

#!/bin/bash
# Two ranks sharing one GPU over gloo: rehearses the sharded train step of bench.py on real kernels
# (RCCL refuses two ranks on one device, so the collectives backend is gloo here).
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  echo "== $*"
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 2 --row-cap 2000000 --backend gloo --share-gpu --no-cpu-baseline \
    --watchdog ${WATCHDOG:-150} --progress --global-batch ${GB:-16384} "$@"
}
if [ -n "$ONLY" ]; then run $ONLY; exit; fi
run
run --eager
run --sharding hybrid --row-wise-min-rows 1000000
run --sharding hybrid --row-wise-min-rows 1000000 --eager

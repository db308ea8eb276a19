#!/bin/bash
# Ranks sharing one GPU: rehearses the sharded train step of bench.py on real kernels.  RCCL refuses two ranks on
# one device, so torch.distributed runs over gloo here and the C host's exchanges over dlrm_comm's rehearsal
# transport (shared host memory).  NP=3 tools/rehearse_2rank.sh for three ranks.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export DLRM_COMM_REHEARSAL_MB=${DLRM_COMM_REHEARSAL_MB:-256}   # pooled vectors of a 16384-sample batch: 109 MB per rank
run() {
  echo "== $*"
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NP:-2} --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus ${NP:-2} --steps 5 --warmup 2 --row-cap 2000000 --backend gloo --share-gpu --no-cpu-baseline \
    --watchdog ${WATCHDOG:-150} --progress --global-batch ${GB:-16384} "$@"
}
if [ -n "$ONLY" ]; then run $ONLY; exit; fi
run                                                         # C host (default)
run --sharding hybrid --row-wise-min-rows 1000000
run --host python
run --host python --sharding hybrid --row-wise-min-rows 1000000

#!/bin/bash
# Parity of every workgroup tile shape of the bf16x3 GEMM (forced through DLRM_GEMM_TILE), then timings of the
# MLP layer shapes per tile at a given batch.   tools/gpu_gemm_tiles.sh [M ...]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for t in 0 1 2 3; do
  echo "== parity, tile $t"
  DLRM_GEMM_TILE=$t timeout -k 10 300 python -u -m pytest tests/test_dense_gpu.py -m gpu -x -q -k "bf16x3 and (linear or backward_weight)" --timeout 120 --timeout-method thread > gpurun_out/gemm_tile$t.log 2>&1 || { tail -30 gpurun_out/gemm_tile$t.log; exit 1; }
  tail -1 gpurun_out/gemm_tile$t.log
done
for M in ${*:-8192 2048}; do
  for t in auto 0 1 2 3; do
    echo "== M=$M tile $t"
    if [ $t = auto ]; then GEMM_PADK=1 GEMM_M=$M timeout -k 10 120 python tools/microbench.py gemm > gpurun_out/gemm_M${M}_$t.log 2>&1
    else DLRM_GEMM_TILE=$t GEMM_PADK=1 GEMM_M=$M timeout -k 10 120 python tools/microbench.py gemm > gpurun_out/gemm_M${M}_$t.log 2>&1; fi
    python3 - gpurun_out/gemm_M${M}_$t.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r=json.loads(l); print("  N%5d K%5d  fwd %6.1f us %5.0f TF | bwd_data %6.1f us %5.0f TF | wgrad %6.1f us %5.0f TF"%(r["N"],r["K"],r["fwd_us"],r["fwd_TF"],r["bwd_data_us"],r["bwd_data_TF"],r["bwd_weight_us"],r["bwd_weight_TF"]))
PY
  done
done

#!/bin/bash
# Everything that profiles/ keeps from one build: bench lines (un-profiled), rocprofv3 kernel traces and PMC passes,
# the kernel sweeps and the cache sweeps.  Each program runs under its own time limit; stops at the first failure.
#   tools/gpu_final.sh <tag> [step|cache]     (two halves, each within one GPU call's time limit)
set -e
cd "$(dirname "$0")/.."
TAG=${1:-final}
HALF=${2:-step}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$HALF" = cache ]; then
  bash tools/gpu_profiles.sh $TAG cache
  echo "== cache sweeps"
  timeout -k 10 900 python3 -u tools/cache_sweep.py $OUT/cache_sweep.json > $OUT/cache_sweep.log 2>&1
  timeout -k 10 600 python3 -u tools/cache_sweep.py $OUT/vec_cache_sweep.json --policy vec > $OUT/vec_cache_sweep.log 2>&1
  du -sh gpurun_out
  echo all done
  exit 0
fi
echo "== bench lines"
timeout -k 10 400 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
timeout -k 10 200 python3 bench.py --global-batch 8192 --no-cpu-baseline > $OUT/b8192_bench_line.json 2> $OUT/b8192.err
timeout -k 10 200 python3 bench.py --config kaggle --no-cpu-baseline > $OUT/kaggle_bench_line.json 2> $OUT/kaggle.err
timeout -k 10 300 python3 bench.py --dist zipf --no-cpu-baseline > $OUT/zipf_bench_line.json 2> $OUT/zipf.err
timeout -k 10 200 python3 bench.py --global-batch 8192 --force-exchange --no-cpu-baseline > $OUT/b8192_exchange_bench_line.json 2> $OUT/b8192x.err
timeout -k 10 200 python3 bench.py --config kaggle --graph --no-cpu-baseline > $OUT/kaggle_graph_bench_line.json 2> $OUT/kaggle_graph.err
timeout -k 10 300 python3 -u tools/gemm_modes.py 65536 $OUT/gemm_modes.json > $OUT/gemm_modes.log 2>&1
echo "== profiles"
bash tools/gpu_profiles.sh $TAG bench b8192 kaggle mfma
echo "== sweeps"
timeout -k 10 500 python3 -u tools/microbench.py sparse $OUT/sparse_update_sweep.json > $OUT/sparse.log 2>&1
timeout -k 10 500 python3 -u tools/microbench.py embbag $OUT/embbag_fwd_sweep.json > $OUT/embbag.log 2>&1
timeout -k 10 200 python3 -u tools/microbench.py sparse_mlperf $OUT/sparse_mlperf.json > $OUT/sparse_mlperf.log 2>&1
timeout -k 10 200 python3 -u tools/microbench.py gemm $OUT/gemm_65536.json > $OUT/gemm_65536.log 2>&1
GEMM_M=8192 timeout -k 10 200 python3 -u tools/microbench.py gemm $OUT/gemm_8192.json > $OUT/gemm_8192.log 2>&1
du -sh gpurun_out
echo all done

#!/bin/bash
# Collects the rocprofv3 summaries kept under profiles/: kernel traces (per-kernel average duration) and
# FETCH_SIZE / WRITE_SIZE passes (HBM-side bytes) of the cache sweeps and of bench.py's other configurations.
# Every profiled program runs under its own time limit; the script stops at the first failure.
#   tools/gpu_profiles.sh <tag> [cache|kaggle|b8192|bench|mfma ...]
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=${1:-r}
shift || true
WHAT=${*:-cache kaggle b8192}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp

trace() {  # name, program args...
  local name=$1; shift
  echo "== kernel trace: $name"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/$name.log" 2>&1
  python3 $ROOT/tools/prof_summary.py "$OUT/$name" "$OUT/$name" && rm -rf "$OUT/$name"
}
pmc() {    # name, "counter [counter ...]" (one pass), program args...; files are tagged with the first counter
  local name=$1 ctrs=$2; shift 2
  local tag=${ctrs%% *}
  echo "== pmc $ctrs: $name"
  timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/${name}_$tag" -- "$@" > "$OUT/${name}_$tag.log" 2>&1
  python3 $ROOT/tools/prof_summary.py "$OUT/${name}_$tag" "$OUT/${name}_$tag" && rm -rf "$OUT/${name}_$tag"
}

for w in $WHAT; do
  case $w in
    cache)
      for rpp in 1 8; do
        A="python3 $ROOT/tools/cache_sweep.py --fracs 0.125 --rpp $rpp --steps 10"
        trace cache_rpp$rpp $A
        pmc cache_rpp$rpp FETCH_SIZE $A
        pmc cache_rpp$rpp WRITE_SIZE $A
      done
      A="python3 $ROOT/tools/cache_sweep.py --policy vec --fracs 0.0625 --steps 10"
      trace cache_vec $A
      pmc cache_vec FETCH_SIZE $A
      pmc cache_vec WRITE_SIZE $A
      ;;
    kaggle)
      trace kaggle python3 $ROOT/bench.py --config kaggle --steps 20 --warmup 3 --no-cpu-baseline --no-exact-reference
      ;;
    b8192)
      trace b8192 python3 $ROOT/bench.py --global-batch 8192 --steps 20 --warmup 3 --no-cpu-baseline --no-exact-reference
      ;;
    bench)
      trace bench python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-reference
      pmc bench FETCH_SIZE python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-reference --no-inference
      pmc bench WRITE_SIZE python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-reference --no-inference
      # (bench.py also launches operators outside the train steps -- the EmbeddingBag kernel on its own when the
      # step reads the rows inside the interaction kernel: per-launch averages only)
      python3 $ROOT/tools/traffic_from_pmc.py "$OUT/bench_FETCH_SIZE_pmc.json" "$OUT/bench_WRITE_SIZE_pmc.json" 0 \
        "$OUT/hbm_traffic.json" mlperf:65536
      ;;
    mfma)   # matrix-pipe busy cycles of the GEMM kernels in the step, at both batch sizes, and where their waves wait
      M="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"
      W="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      pmc mfma_b65536 "$M" python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-reference
      pmc mfma_b8192 "$M" python3 $ROOT/bench.py --global-batch 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-exact-reference
      pmc wait_b65536 "$W" python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-reference
      ;;
  esac
done
du -sh "$OUT"
echo done

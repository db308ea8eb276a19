#!/bin/bash
# The clock the chip holds (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the matrix pipe's busy share under the ablated
# variants of gemm_f16p_kernel (tools/build_f16p_ablate.sh, tools/f16p_ablate.py): one rocprofv3 --pmc pass per variant.
#   tools/gpu_clock_probe.sh <tag> <mask> [<mask> ...]
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=$1; shift
OUT=$ROOT/gpurun_out/clock_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for m in "$@"; do
  echo "== mask $m"
  DLRM_F16P_ABLATE=$m timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
      --output-format csv -d "$OUT/m$m" -- python3 $ROOT/tools/f16p_ablate.py > "$OUT/m$m.log" 2>&1
  python3 $ROOT/tools/prof_summary.py "$OUT/m$m" "$OUT/m$m" && rm -rf "$OUT/m$m"
  python3 - "$OUT/m${m}_pmc.json" <<'PY'
import json, sys
for k, c in json.load(open(sys.argv[1])).items():
    if "gemm_f16p" in k:
        n = c["GRBM_GUI_ACTIVE"]["dispatches"]
        cyc = c["GRBM_GUI_ACTIVE"]["sum"] / 8.0 / n
        us = c["GRBM_GUI_ACTIVE"]["avg_ns_under_pmc"] / 1e3
        print("%-70s %4d launches  %.1f us  clock %.2f GHz  mfma busy %.3f" % (
            k[:70], n, us, cyc / (us * 1e3), c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / 1024.0 / n / cyc))
PY
done

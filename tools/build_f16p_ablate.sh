#!/bin/bash
# Diagnostic build: the library with gemm_f32.hip compiled with -DDLRM_F16P_ABLATE (gemm_f16p_kernel variants that leave
# out one part of a step each -- global loads, split arithmetic, barrier, most B fragment reads; their RESULTS ARE WRONG,
# only their timings mean something).  Output: dlrm_amd/lib_ablate/libdlrm_amd.so, used through DLRM_AMD_LIB by
# tools/f16p_ablate.py; never loaded by tests, bench.py or smoke().
set -e
cd "$(dirname "$0")/.."
make -C dlrm_amd/csrc -j8 > /dev/null
OUT=dlrm_amd/lib_ablate
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on \
    -Xclang -target-feature -Xclang -packed-fp32-ops -DDLRM_F16P_ABLATE \
    -c dlrm_amd/csrc/gemm_f32.hip -o $OUT/gemm_f32.o 2> $OUT/err.txt || { cat $OUT/err.txt; exit 1; }
grep -v "packed-fp32-ops' is not a recognized feature" $OUT/err.txt || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdlrm_amd.so \
    $(ls dlrm_amd/lib/obj/*.o | grep -v gemm_f32.o) $OUT/gemm_f32.o -ldl
rm -f $OUT/gemm_f32.o $OUT/err.txt
echo built $OUT/libdlrm_amd.so

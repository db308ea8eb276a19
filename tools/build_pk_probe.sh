#!/bin/bash
# Diagnostic build for the packed-float32 observation (DESIGN: "kernels of different streams"): the library with
# embbag_bwd.hip compiled WITH packed float32 VALU instructions and with the in-kernel probe (DLRM_PK_PROBE), every
# other object as in the shipped build.  Output: dlrm_amd/lib_probe/libdlrm_amd.so (used through DLRM_AMD_LIB by
# tools/pk_probe.py; never loaded by tests, bench.py or smoke()).
set -e
cd "$(dirname "$0")/.."
make -C dlrm_amd/csrc -j8 > /dev/null
OUT=dlrm_amd/lib_probe
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -DDLRM_PK_PROBE \
    -c dlrm_amd/csrc/embbag_bwd.hip -o $OUT/embbag_bwd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdlrm_amd.so \
    $(ls dlrm_amd/lib/obj/*.o | grep -v embbag_bwd.o) $OUT/embbag_bwd.o -ldl
echo built $OUT/libdlrm_amd.so

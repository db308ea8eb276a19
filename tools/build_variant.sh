#!/bin/bash
# A/B builds: the library with ONE source file compiled with extra flags, every other object as in the shipped build.
#   tools/build_variant.sh <name> <file.hip> <flags...>   ->  dlrm_amd/lib_<name>/libdlrm_amd.so   (use: DLRM_AMD_LIB=<that path>)
# Never loaded by tests, bench.py or smoke() unless DLRM_AMD_LIB says so.
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
make -C dlrm_amd/csrc -j8 > /dev/null
OUT=dlrm_amd/lib_$NAME
mkdir -p $OUT
BASE=$(basename $SRC .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on \
    -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c dlrm_amd/csrc/$BASE.hip -o $OUT/$BASE.o 2> $OUT/err.txt || { cat $OUT/err.txt; exit 1; }
grep -v "packed-fp32-ops' is not a recognized feature" $OUT/err.txt || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdlrm_amd.so \
    $(ls dlrm_amd/lib/obj/*.o | grep -v "/$BASE.o") $OUT/$BASE.o -ldl
rm -f $OUT/$BASE.o $OUT/err.txt
echo built $OUT/libdlrm_amd.so

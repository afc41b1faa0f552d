#!/bin/bash
# dev: a development build of the library with extra macros for ONE source file, for A/B runs on the GPU box through MSCOMP_AMD_LIB
# (hipcc cross-compiles here; build/ travels with the snapshot).   usage: tools/dev/build_variant.sh <name> <source stem> "<-D flags>"
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/ms_compress_amd/csrc
mkdir -p $R/build
make -C $C -j8 > /dev/null
OBJS="api.o lznt1.o lznt1_sa.o util.o xpress_match.o xpress_lazy.o xpress_emit.o xhuff.o decompress.o lzglobal.o stream.o hostbatch.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-exceptions -Wno-unused-function $3 -c $C/$2.hip -o $R/build/$2_$1.o
(cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/libmscomp_amd_$1.so ${OBJS/$2.o/$R/build/$2_$1.o})
echo built build/libmscomp_amd_$1.so

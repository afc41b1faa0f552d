#!/bin/bash
# dev: the LZNT1 variant with a 12-bit hash whose bucket-end table leaves LDS after the sort (-DLZ_TBL_GLOBAL changes LZNT1_SLOT: three files)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/ms_compress_amd/csrc; mkdir -p $R/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-exceptions -Wno-unused-function -DLZ_TBL_GLOBAL $1"
for f in api lznt1 lznt1_sa; do /opt/rocm/bin/hipcc $F -c $C/$f.hip -o $R/build/${f}_lztbl.o & done; wait
(cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/libmscomp_amd_lztbl.so $R/build/api_lztbl.o $R/build/lznt1_lztbl.o $R/build/lznt1_sa_lztbl.o util.o xpress_match.o xpress_lazy.o xpress_emit.o xhuff.o decompress.o lzglobal.o stream.o hostbatch.o)
echo built build/libmscomp_amd_lztbl.so

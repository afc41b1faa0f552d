#!/bin/bash
# Additive probes (profiles/r04_additive_probes.json): development builds of the library with ONE kind of instruction added to a kernel's hot
# loop (csrc: -DXF_PROBE=1|2|3 in xpress_match.hip, -DLZ_PROBE=1..4 in lznt1.hip), loaded through MSCOMP_AMD_LIB, timed on the single-GPU legs.
# How much slower a kernel gets per added instruction says what its time is made of. Run on the GPU box (hipcc is there): bash tools/dev/run_probe.sh
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/ms_compress_amd/csrc
mkdir -p $R/build
make -C $C -j8 > /dev/null
OBJS="api.o lznt1.o lznt1_sa.o util.o xpress_match.o xpress_lazy.o xpress_emit.o xhuff.o decompress.o lzglobal.o stream.o hostbatch.o"
build() {   # build <source stem> <macro> <value> -> build/libmscomp_amd_<stem>_<value>.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-exceptions -Wno-unused-function -D$2=$3 -c $C/$1.hip -o $R/build/$1_$3.o
  (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/libmscomp_amd_$1_$3.so ${OBJS/$1.o/$R/build/$1_$3.o})
}
cd $R
echo "xp_find_kernel (Xpress+Huffman, 12 files): base, then +1 global gather / +10 VALU / +5 LDS dwords per chain step"
python tools/gpu_ab.py xpress_huff 2 single 3 2>&1 | grep codec
for v in 1 2 3; do build xpress_match XF_PROBE $v; MSCOMP_AMD_LIB=$R/build/libmscomp_amd_xpress_match_$v.so python tools/gpu_ab.py xpress_huff 2 single 3 2>&1 | grep codec; done
echo "lznt1_chunk4_kernel (mozilla): base, then +40 VALU / +4 LDS reads per window, +20 VALU / +1 LDS read per finishing step"
python tools/gpu_ab.py lznt1 1 single 5 2>&1 | grep codec
for v in 1 2 3 4; do build lznt1 LZ_PROBE $v; MSCOMP_AMD_LIB=$R/build/libmscomp_amd_lznt1_$v.so python tools/gpu_ab.py lznt1 1 single 5 2>&1 | grep codec; done

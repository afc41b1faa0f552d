# additive probes (dev builds under build/, see the *_PROBE macros in csrc): how much slower does a kernel get per instruction of a kind added to its hot loop?
echo base; python tools/gpu_ab.py lznt1 1 single 5 2>&1 | grep codec
for v in 1 2 3 4; do echo probe $v; MSCOMP_AMD_LIB=$PWD/build/libmscomp_amd_lz$v.so python tools/gpu_ab.py lznt1 1 single 5 2>&1 | grep codec; done

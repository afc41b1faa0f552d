"""Phase cycles of xh_lazy_kernel (library built with EXTRA=-DXZ_PROFILE): spec parse / chain scans / repair+completion / count."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus, sharding
import bench
ctx = m.Context()
cor = bench.Corpus(corpus, torch.device("cuda", 0))
b, o, l, d = bench.single_gpu_workload(cor, "silesia_files")
lib = ctx.lib
buf = (C.c_ulonglong * 16)()
for fi in list(range(12)) + [-1]:
    if fi >= 0:
        oo, ll = o[fi:fi + 1], l[fi:fi + 1]
    else:
        oo, ll = o, l
    j = bench.Job(m, ctx, 4, b, oo, ll)
    j.step(); torch.cuda.synchronize()
    lib.mscomp_amd_debug_xz_prof(buf)
    j.step(); torch.cuda.synchronize()
    lib.mscomp_amd_debug_xz_prof(buf)
    v = list(buf); ch = max(1, v[10])
    print("%-8s chunks %5d  kcycles/chunk: spec %7.1f scan %7.1f repair %7.1f count %7.1f | rounds/chunk %.2f requests/chunk %.1f" % (
        corpus.NAMES[fi] if fi >= 0 else "ALL", ch, v[0] / ch / 1e3, v[1] / ch / 1e3, v[2] / ch / 1e3, v[3] / ch / 1e3, v[8] / ch, v[9] / ch))
    j.close()

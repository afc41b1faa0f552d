"""Phase cycles of lznt1_sa_chunk_kernel (library built with `make -C ms_compress_amd/csrc EXTRA=-DSA_PROFILE`; rebuild without it afterwards)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
ctx = m.Context(); lib = ctx.lib
lib.mscomp_amd_set_lznt1_sa_dict(1)
buf = (C.c_ulonglong * 8)()
names = ["sort", "keys+ranks", "sa/inv/lcp", "trees", "find", "parse+emit"]
for i in (1, 0, 3, 11):
    f = corpus.file_bytes(i).tobytes()
    m.compress_units(2, [f[:1 << 20]], ctx=ctx)
    lib.mscomp_amd_debug_sa_prof(buf)
    t0 = time.perf_counter(); out, st = m.compress_units(2, [f], ctx=ctx); dt = time.perf_counter() - t0
    lib.mscomp_amd_debug_sa_prof(buf)
    v = list(buf); n = max(1, v[7])
    print(corpus.NAMES[i], "chunks", n, "rounds/chunk %.2f" % (v[6] / n), " cycles/chunk:", "  ".join("%s %.0f" % (names[k], v[k] / n) for k in range(6)), " total %.0f" % (sum(v[:6]) / n))

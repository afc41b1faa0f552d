import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
ctx = m.Context(); lib = ctx.lib
buf = (C.c_ulonglong * 8)()
for i in (1, 3, 9, 2):
    f = corpus.file_bytes(i).tobytes()
    comp, st = m.compress_units(4, [f], ctx=ctx)
    lib.mscomp_amd_debug_lzb_prof(buf)
    back, st2 = m.decompress_units(4, comp, [len(f)], ctx=ctx)
    lib.mscomp_amd_debug_lzb_prof(buf)
    v = list(buf); n = max(1, v[6])
    print(corpus.NAMES[i], "ok" if back[0] == f else "BAD", "tiles", n, "cycles/tile: place %.0f scan %.0f bytes %.0f jump %.0f store %.0f | rounds/tile %.2f" % (v[0]/n, v[1]/n, v[2]/n, v[3]/n, v[4]/n, v[7]/n))

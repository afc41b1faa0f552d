"""Counters of the Huffman walk (library built with `make -C ms_compress_amd/csrc EXTRA=-DXHC_PROFILE`): steps, symbols per step, symbols taken one at a time."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ms_compress_amd as m
from ms_compress_amd import corpus
ctx = m.Context(); lib = ctx.lib
buf = (C.c_ulonglong * 8)()
for i in range(12):
    f = corpus.file_bytes(i).tobytes()
    units = [f[o:o + 65536] for o in range(0, len(f), 65536)]
    comp, st = m.compress_units(4, units, ctx=ctx)
    lib.mscomp_amd_debug_xhc_prof(buf)
    back, st2 = m.decompress_units(4, comp, [len(u) for u in units], ctx=ctx)
    lib.mscomp_amd_debug_xhc_prof(buf)
    v = list(buf); n = max(1, v[7])
    print("%-8s chunks %5d  per chunk: steps %6.0f  symbols/step %5.2f  one at a time %6.0f | worst chunk: steps %6d  one at a time %6d" % (corpus.NAMES[i], n, v[0] / n, v[1] / max(1, v[0]), v[2] / n, v[4], v[5]))

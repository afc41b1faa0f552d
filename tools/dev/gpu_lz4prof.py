"""dev (GPU box): phase shares of lznt1_chunk4_kernel from a -DLZ4_PROFILE build (thread 0's cycles per phase, summed over blocks).
    tools/dev/build_variant.sh lz4prof lznt1 -DLZ4_PROFILE;  MSCOMP_AMD_LIB=build/libmscomp_amd_lz4prof.so python tools/dev/gpu_lz4prof.py [corpus member]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import ms_compress_amd as m
from ms_compress_amd import corpus
name = sys.argv[1] if len(sys.argv) > 1 else "mozilla"
ctx = m.Context()
buf = corpus.by_name(name, 51220480).tobytes()
out, st = m.compress_units(2, [buf])          # warm-up (tables, first-touch)
z = (ctypes.c_ulonglong * 8)()
ctx.lib.mscomp_amd_debug_lz4_prof(z)
out, st = m.compress_units(2, [buf])
ctx.lib.mscomp_amd_debug_lz4_prof(z)
v = list(z); tot = float(sum(v)) or 1.0
names = ["stage", "sort (histogram, scan, scatter)", "parse of my segment + seam repair", "-", "wait for the other waves", "cascade check + token scan", "emit"]
print(name, {names[i]: round(v[i] / tot, 3) for i in range(7)}, "cycles per chunk (wave 0): %.0f" % (tot / ((len(buf) + 4095) // 4096)))

"""The 12 corpus files as 12 buffers of one codec, decompressed in one batch: time per kernel (HIP events).  python tools/dev/gpu_files_dec.py <xpress|xpress_huff|lznt1>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ms_compress_amd as m
from ms_compress_amd import corpus
ctx = m.Context()
fmt = m.FORMATS[sys.argv[1] if len(sys.argv) > 1 else "xpress_huff"]
files = [corpus.file_bytes(i).tobytes() for i in range(12)]
comp, st = m.compress_units(fmt, files, ctx=ctx)
assert all(s == 0 for s in st)
m.decompress_units(fmt, comp, [len(f) for f in files], ctx=ctx)
ctx.profile_read(); ctx.profile_enable(True)
t0 = time.perf_counter(); back, st2 = m.decompress_units(fmt, comp, [len(f) for f in files], ctx=ctx); dt = time.perf_counter() - t0
p = ctx.profile_read(); ctx.profile_enable(False)
tot = sum(v[0] for v in p.values())
print(sys.argv[1:], "ok" if all(s == 0 for s in st2) and all(b == f for b, f in zip(back, files)) else "BAD", "%d -> %d B" % (sum(map(len, comp)), sum(map(len, files))),
      "kernels %.1f ms = %.2f GB/s" % (tot, sum(map(len, files)) / tot / 1e6), {k: round(v[0], 3) for k, v in p.items()})
import ctypes as C
buf = (C.c_uint32 * 33)()
ctx.lib.mscomp_amd_debug_lzg_open.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
words = sum(len(f) + 64 for f in files if len(f) >= (1 << 20))
print("open words after each pass:", ctx.lib.mscomp_amd_debug_lzg_open(ctx._h, words, buf), list(buf))

"""ONE Xpress stream per file back to bytes: time per phase (HIP events) for the files named on the command line (default mozilla)."""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")   # (the kernel switches: csrc/api.hip test_hooks_on)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
ctx = m.Context()
for name in (sys.argv[1:] or ["mozilla"]):
    i = corpus.NAMES.index(name)
    f = corpus.file_bytes(i).tobytes()
    comp, st = m.compress_units(3, [f], ctx=ctx)
    assert st[0] == 0
    for mode in (0,):
        ctx.lib.mscomp_amd_debug_set_xpress_decoder(mode)
        m.decompress_units(3, comp, [len(f)], ctx=ctx)
        ctx.profile_read(); ctx.profile_enable(True)
        t0 = time.perf_counter(); back, st2 = m.decompress_units(3, comp, [len(f)], ctx=ctx); dt = time.perf_counter() - t0
        p = ctx.profile_read(); ctx.profile_enable(False)
        print(name, "mode", mode, "ok" if (st2[0] == 0 and back[0] == f) else "BAD", "%d -> %d B" % (len(comp[0]), len(f)),
              "wall %.1f ms" % (dt * 1e3), {k: round(v[0], 3) for k, v in p.items()})
    ctx.lib.mscomp_amd_debug_set_xpress_decoder(0)

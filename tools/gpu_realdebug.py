"""Dev helper (GPU box): Xpress over the real-file corpus in sub-batches of N units under every (finder, emit kernel) combination, each unit's
bytes against the reference's; the first units that differ are dumped to gpurun_out/bad_units/ (input + both outputs) for analysis on the CPU."""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")   # (the kernel switches: csrc/api.hip test_hooks_on)
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ms_compress_amd as m
from oracle import loader
from tools import real_corpus
import test_gpu_realdata as T

fmt = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sub = int(sys.argv[2]) if len(sys.argv) > 2 else 512
t0 = time.time()
c = real_corpus.RealCorpus(int(os.environ.get("MSCOMP_AMD_REAL_MB", "1280")) << 20)
print("corpus: %d files, %d B, read in %.1f s" % (len(c.paths), c.total, time.time() - t0), flush=True)
uoff, ulen, idx = c.units(65536 if fmt == 3 else None)
want, woff, wlen, kind = T._reference_units(loader, fmt, c.blob, uoff, ulen)
ctx = m.Context()
lib = ctx.lib
d_blob = torch.from_numpy(c.blob).to("cuda")
outdir = os.path.join(ROOT, "gpurun_out", "bad_units")
os.makedirs(outdir, exist_ok=True)
dumped = 0
for finder in (1, 2):
    for emit in (0, 1, 2, 3, 4):
        lib.mscomp_amd_debug_set_finder(finder); lib.mscomp_amd_debug_set_xpress_emit(emit)
        bad = []
        for a in range(0, len(ulen), sub):
            b = min(a + sub, len(ulen))
            got, goff, _ = T._gpu_units(m, ctx, fmt, d_blob, uoff[a:b], ulen[a:b])
            for i in range(a, b):
                g = got[int(goff[i - a]):int(goff[i - a + 1])]; w = want[int(woff[i]):int(woff[i]) + int(wlen[i])]
                if len(g) != len(w) or not np.array_equal(g, w):
                    bad.append(i)
                    if dumped < 6:
                        dumped += 1
                        c.blob[int(uoff[i]):int(uoff[i]) + int(ulen[i])].tofile(os.path.join(outdir, "unit%d_f%d_e%d.in" % (i, finder, emit)))
                        g.tofile(os.path.join(outdir, "unit%d_f%d_e%d.gpu" % (i, finder, emit))); w.tofile(os.path.join(outdir, "unit%d_f%d_e%d.ref" % (i, finder, emit)))
        print("finder %d emit %d sub-batches of %d: %d bad of %d%s" % (finder, emit, sub, len(bad), len(ulen),
              ("  first: " + ", ".join("%d (%s +%d, %d B)" % (i, c.paths[int(idx[i])][-40:], int(uoff[i] - c.off[int(idx[i])]), int(ulen[i])) for i in bad[:4])) if bad else ""), flush=True)
lib.mscomp_amd_debug_set_finder(1); lib.mscomp_amd_debug_set_xpress_emit(0)

"""Dev (GPU box, -DXE_PROFILE build of xpress_emit.hip): wave cycles of xpress_emit_kernel per phase on BASELINE configs[4] (or `single`).
   [0] inputs + candidate ballot, [1] serial walk, [2] token stores, [3] flag words + carry, [5] loop head / burst staging"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus, sharding
import bench
ctx = m.Context(); lib = ctx.lib
cor = bench.Corpus(corpus, torch.device("cuda", 0))
fmt = m.FORMATS["xpress"]
kind = sys.argv[1] if len(sys.argv) > 1 else "config5"
if kind == "config5":
    off, ln, _ = bench.config5_units(cor, fmt); d = cor.device_range(0, int(off[-1] + ln[-1]))
else:
    d, off, ln, _ = bench.single_gpu_workload(cor, "silesia_units64k")
j = bench.Job(m, ctx, fmt, d, off, ln)
t, p = bench.timed(j, 1, 1, sharding)
buf = (C.c_ulonglong * 8)(); lib.mscomp_amd_debug_xe_prof(buf)
t, p = bench.timed(j, 2, 0, sharding)
lib.mscomp_amd_debug_xe_prof(buf)
tot = sum(buf[i] for i in range(6)) or 1
nwin = sum((int(x) + 63) // 64 for x in ln) * 2
print({k: round(v[0] / 2, 3) for k, v in p.items()})
print("wave cycles per window: %.0f | inputs %.2f walk %.2f stores %.2f flags/carry %.2f loop/burst %.2f" % (tot / nwin, buf[0] / tot, buf[1] / tot, buf[2] / tot, buf[3] / tot, buf[5] / tot))

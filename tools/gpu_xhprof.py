"""Per-kernel times of one codec on the 12-file corpus (file mode for lznt1 / xpress_huff, 64 KiB units for xpress).
usage: python tools/gpu_xhprof.py [lznt1|xpress|xpress_huff] [steps]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus, sharding
import bench
codec = sys.argv[1] if len(sys.argv) > 1 else "xpress_huff"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = m.Context()
cor = bench.Corpus(corpus, torch.device("cuda", 0))
wl = {"lznt1": "mozilla", "xpress": "silesia_units64k", "xpress_huff": "silesia_files"}[codec]
b, o, l, d = bench.single_gpu_workload(cor, wl)
j = bench.Job(m, ctx, m.FORMATS[codec], b, o, l)
dt, prof = bench.timed(j, steps, 2, sharding)
print(codec, "%.3f ms/step  %.1f GB/s" % (dt / steps * 1e3, j.in_bytes * steps / dt / 1e9), {k: round(v[0] / steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])})

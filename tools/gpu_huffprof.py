"""Latency of the Huffman builder kernel for chosen histograms (run under rocprofv3 --kernel-trace to read the launch durations)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, time
import ms_compress_amd as m
import cases
ctx = m.Context()
hs = cases.huff_histograms()
flat = hs[2]; geo = hs[1]; fib = hs[7]; zipf = hs[4]
for name, h, n in (("flat x1", flat, 1), ("geo x1", geo, 1), ("fib x1", fib, 1), ("zipf x1", zipf, 1), ("flat x3328", flat, 3328), ("geo x3328", geo, 3328), ("zipf x3328", zipf, 3328), ("zipf x1024", zipf, 1024)):
    c = np.tile(np.asarray(h, dtype=np.uint32), (n, 1)); lens = np.zeros((n, 512), dtype=np.uint8)
    for rep in range(2):
        t = time.perf_counter()
        assert ctx.lib.mscomp_amd_debug_huff_lengths(ctx._h, c.ctypes.data, n, lens.ctypes.data) == 0
        dt = time.perf_counter() - t
    print(name, "maxlen", lens.max(), "%.3f ms wall" % (dt * 1e3))

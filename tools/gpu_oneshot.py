"""Dev: PCIe-inclusive rate of the drop-in host-pointer path (ms_compress on host buffers), all three codecs."""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ms_compress_amd as m
from ms_compress_amd import corpus
lib = m.load_library()
data = np.ascontiguousarray(corpus.by_name("mozilla")); n = len(data)
for fmt, name in ((2, "lznt1"), (3, "xpress"), (4, "xpress_huff")):
    cap = lib.ms_max_compressed_size(fmt, n) + 2
    out = np.empty(cap, dtype=np.uint8)
    ol = C.c_size_t(cap)
    st = lib.ms_compress(fmt, data.ctypes.data, n, out.ctypes.data, C.byref(ol))      # warm-up (context, scratch)
    ts = []
    for _ in range(3):
        ol = C.c_size_t(cap); t0 = time.perf_counter()
        st = lib.ms_compress(fmt, data.ctypes.data, n, out.ctypes.data, C.byref(ol)); ts.append(time.perf_counter() - t0)
    print("%-12s ms_compress(host buffers, %d B as one buffer): status %d, out %d, %.1f ms -> %.2f GB/s PCIe-inclusive" % (name, n, st, ol.value, min(ts) * 1e3, n / min(ts) / 1e9))

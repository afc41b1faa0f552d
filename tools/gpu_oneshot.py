"""PCIe-inclusive rate of the drop-in host-pointer path (ms_compress / ms_decompress on host buffers), all three codecs: the same buffers
every call (their page-locking is cached by the runtime) and a fresh pair of buffers per call (what a caller with new data pays), plus a
check against the oracle."""
import sys, os, ctypes as C, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ms_compress_amd as m
from ms_compress_amd import corpus
from oracle import loader
lib = m.load_library()
data = np.ascontiguousarray(corpus.by_name("mozilla")); n = len(data)
res = {}
for fmt, name in ((2, "lznt1"), (3, "xpress"), (4, "xpress_huff")):
    cap = lib.ms_max_compressed_size(fmt, n) + 2
    out = np.empty(cap, dtype=np.uint8)
    ol = C.c_size_t(cap)
    st = lib.ms_compress(fmt, data.ctypes.data, n, out.ctypes.data, C.byref(ol))      # warm-up (context, scratch)
    ts = []
    for _ in range(5):
        ol = C.c_size_t(cap); t0 = time.perf_counter()
        st = lib.ms_compress(fmt, data.ctypes.data, n, out.ctypes.data, C.byref(ol)); ts.append(time.perf_counter() - t0)
    fresh = []
    for _ in range(3):
        d2 = data.copy(); o2 = np.empty(cap, dtype=np.uint8); ol2 = C.c_size_t(cap); t0 = time.perf_counter()
        lib.ms_compress(fmt, d2.ctypes.data, n, o2.ctypes.data, C.byref(ol2)); fresh.append(time.perf_counter() - t0)
    comp = out[: ol.value].tobytes()
    back = np.empty(n, dtype=np.uint8); bl = C.c_size_t(n)
    lib.ms_decompress(fmt, comp, len(comp), back.ctypes.data, C.byref(bl))
    td = []
    for _ in range(3):
        bl = C.c_size_t(n); t0 = time.perf_counter(); sd = lib.ms_decompress(fmt, comp, len(comp), back.ctypes.data, C.byref(bl)); td.append(time.perf_counter() - t0)
    ok = sd == 0 and bl.value == n and bytes(back) == data.tobytes()
    res[name] = {"compress_ms": round(min(ts) * 1e3, 2), "compress_GBps": round(n / min(ts) / 1e9, 2), "fresh_buffers_ms": round(min(fresh) * 1e3, 2),
                 "decompress_ms": round(min(td) * 1e3, 2), "decompress_GBps": round(n / min(td) / 1e9, 2), "round_trip_ok": ok, "status": st, "out": ol.value}
    print("%-12s ms_compress(host buffers, %d B as one buffer): status %d, out %d, %.2f ms -> %.2f GB/s PCIe-inclusive (fresh buffers %.2f ms); ms_decompress %.2f ms, round trip %s"
          % (name, n, st, ol.value, min(ts) * 1e3, n / min(ts) / 1e9, min(fresh) * 1e3, min(td) * 1e3, ok))
    if fmt == 2:
        want = loader.oracle_compress(2, data.tobytes())[1] if n < 60_000_000 else None
        print("   oracle bytes equal:", want == comp, "EOB:", bytes(out[ol.value: ol.value + 2]) == b"\0\0")
        # short capacities: exact fit, one short
        for c2 in (len(comp), len(comp) + 1, len(comp) + 2, len(comp) - 1, len(comp) // 2):
            o3 = np.full(len(comp) + 8, 0xAA, dtype=np.uint8); l3 = C.c_size_t(c2)
            s3 = lib.ms_compress(2, data.ctypes.data, n, o3.ctypes.data, C.byref(l3))
            print("   capacity %d: status %d len %s, bytes behind the capacity untouched: %s" % (c2, s3, l3.value if s3 == 0 else "-", bool((o3[max(c2, 0) + (2 if False else 0):] == 0xAA).all()) if s3 != 0 or c2 - l3.value < 2 else bool((o3[l3.value + 2:] == 0xAA).all())))
import json; print(json.dumps(res))

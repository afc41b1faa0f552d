"""Host-pointer ms_compress(LZNT1) of large buffers (mapped caller buffers, one launch) from several host threads at once; digests of the reference."""
import sys, os, json, hashlib, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ms_compress_amd as m
from ms_compress_amd import corpus
lib = m.load_library()
gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "corpus_full.json")))
names = ["mozilla", "webster", "nci", "samba"]
bad = []
def work(name):
    data = np.ascontiguousarray(corpus.by_name(name)); n = len(data)
    cap = lib.ms_max_compressed_size(2, n) + 2
    out = np.empty(cap, dtype=np.uint8)
    for it in range(4):
        ol = C.c_size_t(cap)
        st = lib.ms_compress(2, data.ctypes.data, n, out.ctypes.data, C.byref(ol))
        ok = st == 0 and ol.value == gold[name]["lznt1"]["len"] and hashlib.sha256(out[: ol.value].tobytes()).hexdigest() == gold[name]["lznt1"]["sha256"]
        if not ok: bad.append((name, it, st, ol.value))
ts = [threading.Thread(target=work, args=(n,)) for n in names]
[t.start() for t in ts]; [t.join() for t in ts]
print("threads", len(ts), "calls", 4 * len(ts), "bad", bad)
sys.exit(1 if bad else 0)

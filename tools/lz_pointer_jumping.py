"""CPU check of the whole-output resolution planned for ONE large buffer: every output byte points at its source (a literal
at itself); ptr = ptr[ptr] is repeated until nothing changes; the bytes are then literal[ptr]. Counts the passes and compares with the
decoder's output."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import loader
from ms_compress_amd import corpus
lib = loader.load_oracle()
lib.orc_xh_sources.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
lib.orc_xh_sources.restype = C.c_longlong
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
for i, name in enumerate(corpus.NAMES):
    data = corpus.file_bytes(i, N)
    comp = loader.oracle_compress(4, data.tobytes())[1]
    src = np.zeros(len(data) + 8, dtype=np.uint32); out = np.zeros(len(data) + 64, dtype=np.uint8)
    n = lib.orc_xh_sources(comp, len(comp), len(data), src.ctypes.data, out.ctypes.data)
    assert n == len(data) and np.array_equal(out[:n], data)
    ptr = src[:n].astype(np.int64)
    lit = ptr == np.arange(n)
    vals = np.where(lit, data, 0).astype(np.uint8)                    # only the literal bytes are known at the start
    passes = 0
    while True:
        nxt = ptr[ptr]
        if np.array_equal(nxt, ptr):
            break
        ptr = nxt; passes += 1
    assert bool(lit[ptr].all()) and np.array_equal(vals[ptr], data)
    print("%-8s %9d B  literals %5.1f %%  pointer-doubling passes %2d  -> output identical" % (name, n, 100.0 * lit.mean(), passes))

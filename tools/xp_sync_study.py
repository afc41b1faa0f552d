"""CPU study: would ONE Xpress stream parse in parallel?  A parse started at an arbitrary offset (as if a flag word began there) reads
garbage tokens, but whenever it reaches a flag word of the true parse in the true state (same pending-nibble state) it is synchronised for
good. How many bytes does that take?  Random starts per corpus member (one stream per file)."""
import ctypes as C, random, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import loader
from ms_compress_amd import corpus
lib = loader.load_oracle()
lib.orc_xp_flag_starts.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]; lib.orc_xp_flag_starts.restype = C.c_longlong
lib.orc_xp_sync.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]; lib.orc_xp_sync.restype = C.c_size_t
lib.orc_xp_sync2.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]; lib.orc_xp_sync2.restype = C.c_size_t
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
TWO = len(sys.argv) > 2 and sys.argv[2] == "two"
rnd = random.Random(3)
allv = []
for i, name in enumerate(corpus.NAMES):
    data = corpus.file_bytes(i, N).tobytes()
    comp = loader.oracle_compress(3, data)[1]
    mark = np.zeros(len(comp) + 8, dtype=np.uint8); hp = np.zeros(len(comp) + 8, dtype=np.uint32)
    assert lib.orc_xp_flag_starts(comp, len(comp), len(data), mark.ctypes.data, hp.ctypes.data) == len(data)
    dist, never = [], 0
    for _ in range(400):
        s = rnd.randrange(0, max(1, len(comp) - 200000))
        e = lib.orc_xp_sync(comp, len(comp), s, mark.ctypes.data, hp.ctypes.data)
        if TWO: e = min(e, lib.orc_xp_sync2(comp, len(comp), s, mark.ctypes.data, hp.ctypes.data, 1))   # the better of the two hypotheses about a pending nibble
        if e >= len(comp): never += 1
        else: dist.append(e - s)
    d = np.array(dist) if dist else np.array([0])
    allv += dist
    print("%-8s stream %9d B  synchronised %3d/400  bytes until then: median %6d  p90 %7d  p99 %7d  max %8d" %
          (name, len(comp), len(dist), int(np.median(d)), int(np.percentile(d, 90)), int(np.percentile(d, 99)), int(d.max())))
a = np.array(allv)
print("all: median %d  p90 %d  p99 %d  max %d" % (int(np.median(a)), int(np.percentile(a, 90)), int(np.percentile(a, 99)), int(a.max())))

"""CPU study for DESIGN 4.5: copy-chain depth of Xpress+Huffman streams (how many pointer-jumping passes a whole-output resolution
of the tokens would need: ceil(log2(max depth + 1)))."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import loader
from ms_compress_amd import corpus
lib = loader.load_oracle()
lib.orc_xh_copy_depths.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
lib.orc_xh_copy_depths.restype = C.c_longlong
N = int(sys.argv[1]) if len(sys.argv) > 1 else None
for i, name in enumerate(corpus.NAMES):
    data = corpus.file_bytes(i, N).tobytes()
    comp = loader.oracle_compress(4, data)[1]
    line = "%-8s %9d B" % (name, len(data))
    for mode in (0, 1):                          # 0: byte i copies byte i - off; 1: a match's byte i copies byte (i mod off) of its first period
        lib.orc_xh_depth_first_period(mode)
        depth = np.zeros(len(data) + 8, dtype=np.uint32)
        n = lib.orc_xh_copy_depths(comp, len(comp), len(data), depth.ctypes.data)
        assert n == len(data)
        d = depth[:n]
        line += "  | %s: mean %7.1f  p99 %6d  max %7d -> %2d passes" % ("byte-wise" if mode == 0 else "first period", float(d.mean()), int(np.percentile(d, 99)), int(d.max()), int(np.ceil(np.log2(int(d.max()) + 1))))
    print(line)

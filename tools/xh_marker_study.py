"""CPU study for DESIGN 4.5 (chunk-parallel Xpress+Huffman decoding): how well does "the 512 nibbles of a 256-byte window satisfy
the Kraft equality" mark the chunk starts of real streams?  For every corpus file: the stream of the oracle's encoder (== the
reference's), the true chunk starts (oracle decoder), and every offset whose window sums to exactly 2^15."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import loader
from ms_compress_amd import corpus

lib = loader.load_oracle()
lib.orc_xh_chunk_starts.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]
lib.orc_xh_chunk_starts.restype = C.c_long
K = np.zeros(16, dtype=np.uint32); K[1:] = 1 << (15 - np.arange(1, 16))
BYTE = (K[np.arange(256) & 15] + K[np.arange(256) >> 4]).astype(np.uint64)          # contribution of one table byte
tot_true = tot_cand = tot_missed = 0
for i, name in enumerate(corpus.NAMES):
    data = corpus.file_bytes(i, int(sys.argv[1]) if len(sys.argv) > 1 else None).tobytes()
    comp = loader.oracle_compress(4, data)[1]
    starts = np.zeros(len(data) // 65536 + 8, dtype=np.uint64)
    k = lib.orc_xh_chunk_starts(comp, len(comp), len(data), starts.ctypes.data, len(starts))
    assert k > 0
    true = set(int(x) for x in starts[:k])
    c = np.concatenate([[0], np.cumsum(BYTE[np.frombuffer(comp, dtype=np.uint8)])])
    win = c[256:] - c[:-256]                                                       # window starting at p: sum over comp[p .. p+255]
    cand = set(int(p) for p in np.nonzero(win == 32768)[0])
    missed = true - cand
    print("%-8s %9d B -> %9d B  chunks %4d  candidates %5d  false %4d  missed %d" % (name, len(data), len(comp), k, len(cand), len(cand - true), len(missed)))
    tot_true += k; tot_cand += len(cand); tot_missed += len(missed)
print("total: chunks %d, candidates %d (false %d), missed %d" % (tot_true, tot_cand, tot_cand - (tot_true - tot_missed), tot_missed))

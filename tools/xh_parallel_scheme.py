"""CPU prototype of chunk-parallel Xpress+Huffman decoding (plan for the next round), end to end on the corpus:
  1. candidates: every offset whose 256-byte window satisfies the Kraft equality (plus offset 0);
  2. each candidate is parsed INDEPENDENTLY as one chunk (oracle helper orc_xh_parse_chunk: end offset, bytes produced, how far its
     matches reach in front of the chunk, stream end) -- the work of one speculative wave;
  3. the chain from offset 0 is followed through the candidates (end(k) must be a candidate), the output offsets are the prefix sums
     of the produced bytes, a chunk may not reach in front of the buffer;
  4. the result must be the decoder's: same chunk starts, same total length.
Reports per file how many candidates were parsed in vain and whether the chain closed without the serial fallback."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import loader
from ms_compress_amd import corpus

lib = loader.load_oracle()
lib.orc_xh_parse_chunk.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
lib.orc_xh_parse_chunk.restype = C.c_int
lib.orc_xh_chunk_starts.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]
lib.orc_xh_chunk_starts.restype = C.c_long
K = np.zeros(16, dtype=np.uint32); K[1:] = 1 << (15 - np.arange(1, 16))
BYTE = (K[np.arange(256) & 15] + K[np.arange(256) >> 4]).astype(np.uint64)
N = int(sys.argv[1]) if len(sys.argv) > 1 else None
ok_all = True
for i, name in enumerate(corpus.NAMES):
    data = corpus.file_bytes(i, N).tobytes()
    comp = loader.oracle_compress(4, data)[1]
    c = np.concatenate([[0], np.cumsum(BYTE[np.frombuffer(comp, dtype=np.uint8)])])
    cand = sorted(set(int(p) for p in np.nonzero(c[256:] - c[:-256] == 32768)[0]) | {0})
    parsed = {}
    res = (C.c_uint64 * 4)()
    for p in cand:                                                     # independent: one wave each on the GPU
        st = lib.orc_xh_parse_chunk(comp, len(comp), p, res)
        parsed[p] = (st, int(res[0]), int(res[1]), int(res[2]), int(res[3]))
    pos, out, chain, closed = 0, 0, [], False
    while pos in parsed:
        st, end, produced, reach, ntok = parsed[pos]
        if st < 0 or reach > out:
            break
        chain.append(pos); out += produced
        if st == 1:
            closed = end == len(comp)
            break
        pos = end
    starts = np.zeros(len(data) // 65536 + 8, dtype=np.uint64)
    k = lib.orc_xh_chunk_starts(comp, len(comp), len(data), starts.ctypes.data, len(starts))
    same = closed and chain == [int(x) for x in starts[:k]] and out == len(data)
    ok_all &= same
    print("%-8s chunks %4d  candidates %4d (parsed in vain %3d)  chain closed %s  offsets and length right %s" %
          (name, k, len(cand), len(cand) - len(chain), closed, same))
print("ALL FILES DECODE THROUGH THE SPECULATIVE CHAIN" if ok_all else "SOME FILE NEEDS THE SERIAL FALLBACK")

"""dev (GPU box): the most repetitive 64 KiB pieces (Xpress) / whole small files (LZNT1, Xpress+Huffman) of the real-file corpus as batches of many
CONCURRENT COPIES, several passes, every copy against the reference's bytes -- the regime in which the long-match-cache race showed (3-10 % of 512
copies) while one copy per batch ran clean.   python tools/dev/gpu_copies_stress.py [MB of corpus] [units per codec] [copies] [passes]"""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ms_compress_amd as m
from oracle import loader
from tools import real_corpus
import test_gpu_realdata as T
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nun = int(sys.argv[2]) if len(sys.argv) > 2 else 24
copies = int(sys.argv[3]) if len(sys.argv) > 3 else 96
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 3
c = real_corpus.RealCorpus(mb << 20, deadline_s=300)
print("corpus: %d files, %d B" % (len(c.paths), c.total), flush=True)
ctx = m.Context()
for fmt in (3, 2, 4):
    if fmt == 3:
        uoff, ulen, idx = c.units(65536)
        keep = np.nonzero(ulen == 65536)[0]
    else:
        uoff, ulen, idx = c.units(None)
        keep = np.nonzero((ulen >= (128 << 10)) & (ulen <= (1 << 20)))[0]
    uoff, ulen, idx = uoff[keep], ulen[keep], idx[keep]
    want, woff, wlen, kind = T._reference_units(loader, fmt, c.blob, uoff, ulen)
    ratio = wlen.astype(np.float64) / ulen.astype(np.float64)
    order = np.argsort(ratio)                                        # most compressible first: long matches, deep chains, full buckets
    pick = list(order[:nun // 2]) + list(order[len(order) // 2: len(order) // 2 + nun - nun // 2])      # ... and some ordinary ones
    bad_total = 0
    for i in pick:
        u = c.blob[int(uoff[i]):int(uoff[i]) + int(ulen[i])].tobytes()
        exp = want[int(woff[i]):int(woff[i]) + int(wlen[i])].tobytes()
        n_copies = copies if fmt == 3 else max(8, min(copies, (96 << 20) // len(u)))
        for p in range(passes):
            got, st = m.compress_units(fmt, [u] * n_copies, ctx=ctx)
            bad = sum(1 for g in got if g != exp)
            if bad:
                bad_total += bad
                print("fmt %d unit %d (%s +%d, %d B, CR %.3f) pass %d: %d of %d copies differ" % (fmt, i, c.paths[int(idx[i])][-40:], int(uoff[i] - c.off[int(idx[i])]), len(u), ratio[i], p, bad, n_copies), flush=True)
    print("fmt %d: %d units x <= %d copies x %d passes: %d differing copies (checker: %s)" % (fmt, len(pick), copies, passes, bad_total, kind), flush=True)

"""The same soak for LARGE units (segment walk of one Xpress stream, chunk-parallel Xpress+Huffman buffers, byte stage by all CUs): units of 0.6-4 MB,
their streams cut / corrupted, capacities around the output size.   python tools/dev/fuzz_big.py [seed] [rounds]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ms_compress_amd as m
from ms_compress_amd import corpus
from oracle import loader
loader.build(); loader.load_oracle()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rnd = random.Random(seed)
ctx = m.Context()


def gen(n):
    kind = rnd.randrange(5)
    if kind == 0: return rnd.randbytes(n)
    if kind == 1:
        i = rnd.randrange(12); f = corpus.file_bytes(i); o = rnd.randrange(0, max(1, len(f) - n)); return f[o:o + n].tobytes()
    if kind == 2: return bytes(rnd.choice(b"abc") for _ in range(min(n, 700000))) * (n // 700000 + 1)
    out = bytearray(rnd.randbytes(rnd.randint(1, 5000)))
    while len(out) < n:
        if rnd.random() < 0.3: out += rnd.randbytes(rnd.randint(1, 40))
        ln = rnd.choice((3, 9, 10, 24, 25, 279, 280, 3000, 70000, 200000))
        off = rnd.randint(1, min(len(out), 8192 if kind == 3 else 65535))
        start = len(out) - off
        for k in range(ln): out.append(out[start + k])
    return bytes(out[:n])


bad = 0
for r in range(rounds):
    units = [gen(rnd.randint(600_000, 4_000_000)) for _ in range(10)]
    for fmt in (3, 4):
        comp, st = m.compress_units(fmt, units, ctx=ctx)
        assert all(s == 0 for s in st)
        streams = []
        for u, c in zip(units, comp):
            streams.append((c, len(u)))
            streams.append((c, len(u) + rnd.choice((1, 4096, 1 << 20))))
            streams.append((c, len(u) - rnd.choice((1, 4096, len(u) // 3))))
            streams.append((c[: rnd.randrange(len(c) // 2, len(c))], len(u)))
            b = bytearray(c)
            for _ in range(rnd.randint(1, 3)): b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
            streams.append((bytes(b), len(u) + rnd.choice((0, 0, 100000))))
        outs, sts = m.decompress_units(fmt, [s for s, _ in streams], [c for _, c in streams], ctx=ctx)
        n_ok = n_err = 0
        for (s_, cap), o, st_ in zip(streams, outs, sts):
            so, oo, undefined = loader.oracle_decompress_ex(fmt, s_, cap)
            if undefined: continue
            if st_ != so or (so == 0 and o != oo):
                bad += 1; print("MISMATCH fmt %d stream %d B cap %d: gpu %d / %d B, checker %d / %d B" % (fmt, len(s_), cap, st_, len(o), so, len(oo)))
            n_ok += so == 0; n_err += so != 0
        print("round %d fmt %d: %d ok, %d errors agreed (streams of %d - %d KB)" % (r, fmt, n_ok, n_err, min(len(c) for c in comp) >> 10, max(len(c) for c in comp) >> 10), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

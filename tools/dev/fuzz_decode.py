"""Differential soak of the decoders against the checker (oracle): random structured inputs -> GPU compress (bytes = oracle) -> GPU decompress;
then truncated / corrupted streams with random capacities: status and bytes must be the checker's.   python tools/dev/fuzz_decode.py [seed] [rounds]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import ms_compress_amd as m
from oracle import loader
loader.build(); loader.load_oracle()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rnd = random.Random(seed)
ctx = m.Context()


def gen(n):
    kind = rnd.randrange(6)
    if kind == 0: return rnd.randbytes(n)
    if kind == 1: return bytes(rnd.choice(b"ab") for _ in range(n))
    if kind == 2: return bytes(n)
    out = bytearray(rnd.randbytes(rnd.randint(1, 64)))
    while len(out) < n:
        if rnd.random() < (0.5 if kind == 3 else 0.15): out += rnd.randbytes(rnd.randint(1, 6))
        ln = rnd.choice((3, 5, 9, 10, 17, 24, 25, 40, 279, 280, 300, 2000, 70000 if kind == 5 else 33))
        off = rnd.randint(1, min(len(out), 65535 if kind == 4 else 9000))
        for _ in range(ln): out.append(out[-off])
    return bytes(out[:n])


bad_total = 0
for r in range(rounds):
    units = [gen(rnd.choice((0, 1, 5, 300, 4096, 4097, 65536, 65537, 70000, 200000, rnd.randint(1, 400000)))) for _ in range(60)] + [gen(1_300_000)]
    for fmt in (2, 3, 4):
        comp, st = m.compress_units(fmt, units, ctx=ctx)
        for u, c, s in zip(units, comp, st):
            es, ec = loader.oracle_compress(fmt, u)
            assert s == 0 and es == 0 and c == ec, ("compress", fmt, len(u))
        streams = []
        for u, c in zip(units, comp):
            streams.append((c, len(u)))
            if len(c) > 8:
                streams.append((c[: rnd.randrange(1, len(c))], len(u)))
                b = bytearray(c)
                for _ in range(rnd.randint(1, 3)): b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
                streams.append((bytes(b), len(u) + rnd.choice((0, 0, 7, 5000))))
                streams.append((c, max(0, len(u) - rnd.choice((1, 2, 100, len(u) // 2 + 1)))))
        outs, sts = m.decompress_units(fmt, [s for s, _ in streams], [c for _, c in streams], ctx=ctx)
        n_ok = n_err = n_undef = 0
        for (s_, cap), o, st_ in zip(streams, outs, sts):
            so, oo, undefined = loader.oracle_decompress_ex(fmt, s_, cap)
            if undefined: n_undef += 1; continue
            if st_ != so or (so == 0 and o != oo):
                bad_total += 1
                print("MISMATCH fmt %d stream %d B cap %d: gpu %d / %d B, checker %d / %d B" % (fmt, len(s_), cap, st_, len(o), so, len(oo)))
            n_ok += so == 0; n_err += so != 0
        print("round %d fmt %d: %d streams ok, %d errors agreed, %d undefined in the reference (skipped)" % (r, fmt, n_ok, n_err, n_undef), flush=True)
print("mismatches:", bad_total)
sys.exit(1 if bad_total else 0)

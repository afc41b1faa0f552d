"""Differential soak of the LZNT1 suffix-array flavour (csrc/lznt1_sa.hip) against the oracle's restatement.   python tools/dev/fuzz_sa.py [seed] [units]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ms_compress_amd as m
from oracle import loader
loader.build(); loader.load_oracle()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rnd = random.Random(seed)
ctx = m.Context()


def gen(n):
    kind = rnd.randrange(6)
    if kind == 0: return rnd.randbytes(n)
    if kind == 1: return bytes(rnd.choice(b"ab") for _ in range(n))
    if kind == 2: return bytes(rnd.choice(b"abc") for _ in range(n))
    if kind == 3: return (rnd.randbytes(rnd.randint(1, 70)) * (n // 1 + 1))[:n]
    out = bytearray(rnd.randbytes(rnd.randint(1, 64)))
    while len(out) < n:
        if rnd.random() < 0.4: out += rnd.randbytes(rnd.randint(1, 6))
        ln = rnd.choice((3, 4, 5, 9, 17, 33, 100, 1000, 4000)); off = rnd.randint(1, min(len(out), 4095))
        for _ in range(ln): out.append(out[-off])
    return bytes(out[:n])


units = [gen(rnd.choice((0, 1, 2, 3, 4, 5, 100, 4095, 4096, 4097, 8191, 8192, 12289, rnd.randint(1, 40000)))) for _ in range(count)]
ctx.lib.mscomp_amd_set_lznt1_sa_dict(1)
try:
    got, st = m.compress_units(2, units, ctx=ctx)
finally:
    ctx.lib.mscomp_amd_set_lznt1_sa_dict(0)
bad = 0
for u, g, s in zip(units, got, st):
    es, exp = loader.oracle_compress_sa(u)
    if s != 0 or es != 0 or g != exp: bad += 1; print("MISMATCH", len(u), s, es, len(g), len(exp))
back, st2 = m.decompress_units(2, got, [len(u) for u in units], ctx=ctx)
bad += sum(1 for u, b, s in zip(units, back, st2) if s != 0 or b != u)
print("units", len(units), "mismatches", bad)
sys.exit(1 if bad else 0)

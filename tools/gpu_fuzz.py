"""Dev helper (GPU box): differential fuzz of the HIP path against the oracle. usage: gpu_fuzz.py <seed> <n_units>"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ms_compress_amd as m
from oracle import loader

def gen(rnd):
    kind = rnd.randrange(9)
    n = rnd.choice([rnd.randrange(0, 300), rnd.randrange(300, 9000), rnd.randrange(9000, 70000), rnd.randrange(60000, 200000)])
    if kind == 0:
        return bytes(rnd.getrandbits(8) for _ in range(n))
    if kind == 1:                      # small alphabet
        k = rnd.randrange(1, 12); al = bytes(rnd.getrandbits(8) for _ in range(k))
        return bytes(rnd.choice(al) for _ in range(n))
    if kind == 2:                      # periodic with noise
        per = rnd.randrange(1, 400); base = bytes(rnd.getrandbits(8) for _ in range(per))
        out = bytearray((base * (n // per + 1))[:n])
        for _ in range(rnd.randrange(0, max(1, n // 200))):
            if n: out[rnd.randrange(n)] = rnd.getrandbits(8)
        return bytes(out)
    if kind == 3:                      # synthetic LZ, long range
        out = bytearray()
        while len(out) < n:
            if len(out) > 4 and rnd.random() < 0.6:
                off = rnd.randrange(1, min(len(out), rnd.choice([16, 300, 9000, 70000])) + 1)
                for _ in range(rnd.choice([3, 4, 7, 20, 50, 300, 5000])):
                    out.append(out[-off])
            else:
                out.append(rnd.getrandbits(8) & rnd.choice([0xFF, 0x0F, 0x03]))
        return bytes(out[:n])
    if kind == 4:                      # zero runs + stuff
        out = bytearray()
        while len(out) < n:
            if rnd.random() < 0.5: out += bytes(rnd.choice([1, 5, 100, 5000, 70000]))
            else: out += bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 200)))
        return bytes(out[:n])
    if kind == 5:                      # words
        ws = [bytes(rnd.choice(b"abcdefghij ") for _ in range(rnd.randrange(1, 9))) for _ in range(rnd.randrange(2, 60))]
        return b"".join(rnd.choice(ws) for _ in range(n // 3 + 1))[:n]
    if kind == 6:                      # records
        rec = rnd.randrange(8, 300); out = bytearray(); ctr = 0
        tmpl = bytes(rnd.getrandbits(8) for _ in range(rec))
        while len(out) < n:
            r = bytearray(tmpl); r[0:2] = (ctr & 0xFFFF).to_bytes(2, "little"); ctr += 1
            if rnd.random() < 0.3: r[rnd.randrange(rec)] = rnd.getrandbits(8)
            out += r
        return bytes(out[:n])
    if kind == 7:                      # 16-bit random walk
        v = 1000; out = bytearray()
        while len(out) < n:
            v = (v + rnd.randrange(-3, 4)) & 0xFFFF; out += v.to_bytes(2, "little")
        return bytes(out[:n])
    return (bytes([rnd.getrandbits(8)]) * rnd.randrange(1, 5000) + bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(0, 50)))) * (n // 100 + 1)

seed, count = int(sys.argv[1]), int(sys.argv[2])
rnd = random.Random(seed)
units = [gen(rnd)[:200000] for _ in range(count)]
ctx = m.Context()
bad = 0
for fmt in (2, 3, 4):
    got, st = m.compress_units(fmt, units, ctx=ctx)
    for i, (u, g, s) in enumerate(zip(units, got, st)):
        es, exp = loader.oracle_compress(fmt, u)
        if s != 0 or g != exp:
            bad += 1
            if bad <= 5:
                print("MISMATCH fmt %d unit %d len %d status %d got %s exp %d" % (fmt, i, len(u), s, None if g is None else len(g), len(exp)))
                open("gpurun_out/fuzz_fail_%d_%d_%d.bin" % (seed, fmt, i), "wb").write(u)
print("seed", seed, "units", len(units), "bytes", sum(map(len, units)), "bad", bad)

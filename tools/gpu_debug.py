"""Dev helper (GPU box): run a codec over the edge-case families and print the first mismatches in detail."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cases
import ms_compress_amd as m
from ms_compress_amd import corpus
from oracle import loader

fmt = int(sys.argv[1]) if len(sys.argv) > 1 else 2
units = cases.edge_cases() + [corpus.file_bytes(i, 300_000).tobytes() for i in range(12)] + [cases.mixed_buffer()]
ctx = m.Context()
got, st = m.compress_units(fmt, units, ctx=ctx)
bad = 0
for i, (u, g, s) in enumerate(zip(units, got, st)):
    es, exp = loader.oracle_compress(fmt, u)
    if s != 0 or g != exp:
        bad += 1
        if bad <= 8:
            g = g or b""
            k = next((j for j in range(min(len(g), len(exp))) if g[j] != exp[j]), min(len(g), len(exp)))
            print("MISMATCH unit %d len %d status %d: got %d B exp %d B first diff @%d" % (i, len(u), s, len(g), len(exp), k))
            print("   in :", u[:32].hex())
            print("   got:", g[max(0, k - 8): k + 24].hex())
            print("   exp:", exp[max(0, k - 8): k + 24].hex())
print("units", len(units), "bad", bad)

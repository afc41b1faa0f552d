"""tests/golden/lznt1_sa.json: what the REAL reference built with -DMSCOMP_WITH_LZNT1_SA_DICT (oracle/_ref/libMSCompression_sa.so,
oracle/Makefile) writes for LZNT1: the 1 MB corpus slices, the mixed buffer, one digest over the edge families, and three short
known answers in hex. Dev container only.   python tools/make_golden_sa.py"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from oracle import loader
from ms_compress_amd import corpus

assert loader.load_ref_sa() is not None, "oracle/_ref/libMSCompression_sa.so missing: run make -C oracle"
sha = lambda b: hashlib.sha256(b).hexdigest()
g = {"kat": {}, "corpus": {}}
for name, data in {"abc*100": b"abc" * 100, "abracadabra*6": b"abracadabra" * 6, "zeros4097": bytes(4097),
                   "banana-bandana": b"banana bandana banana bandana cabana banana",
                   "two-candidates": b"aabacbbacbbbcabacacaaabbcbcbbaccbb"}.items():
    st, out = loader.ref_compress_sa(data); assert st == 0
    g["kat"][name] = {"input_hex": data.hex() if len(data) <= 128 else None, "input_len": len(data), "hex": out.hex(),
                      "differs_from_default": out != loader.ref_compress(2, data)[1]}
N = 1_000_000
for i, name in enumerate(corpus.NAMES):
    data = corpus.file_bytes(i, N).tobytes()
    st, out = loader.ref_compress_sa(data); assert st == 0
    g["corpus"][name] = {"input_len": N, "input_sha256": sha(data), "len": len(out), "sha256": sha(out)}
mixed = cases.mixed_buffer()
st, out = loader.ref_compress_sa(mixed); assert st == 0
g["corpus"]["mixed_buffer"] = {"input_len": len(mixed), "input_sha256": sha(mixed), "len": len(out), "sha256": sha(out)}
h = hashlib.sha256(); tot = 0
units = cases.edge_cases()
for u in units:
    st, out = loader.ref_compress_sa(u); assert st == 0
    h.update(len(out).to_bytes(8, "little")); h.update(out); tot += len(out)
g["edge_families"] = {"units": len(units), "total_len": tot, "sha256": h.hexdigest()}
json.dump(g, open(os.path.join(ROOT, "tests/golden/lznt1_sa.json"), "w"), indent=1)
print(json.dumps(g["kat"], indent=1)[:600]); print(g["edge_families"])

"""Generate tests/golden/*.json from the REAL reference compiled here (oracle/_ref, built by oracle/Makefile from
/root/reference/src). Run in the dev container only; the fixtures (data: inputs are seeded/generated, outputs are
sizes + SHA-256 or short hex strings) travel to the GPU box, the reference does not.

    python tools/make_golden.py
"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from oracle import loader
from ms_compress_amd import corpus

ref = loader.load_ref()
assert ref is not None, "oracle/_ref missing: run make -C oracle"
FMTS = {"lznt1": 2, "xpress": 3, "xpress_huff": 4}
sha = lambda b: hashlib.sha256(b).hexdigest()

# 1. known-answer vectors: small inputs, full output hex (XH: size + sha256)
kat_inputs = {"empty": b"", "a": b"a", "abc": b"abc", "a-z": bytes(range(97, 123)), "abc*100": b"abc" * 100,
              "zeros4096": bytes(4096), "zeros4097": bytes(4097), "zeros70000": bytes(70000)}
kat = {}
for name, data in kat_inputs.items():
    kat[name] = {"input_sha256": sha(data), "input_len": len(data)}
    for fn, f in FMTS.items():
        st, out = loader.ref_compress(f, data)
        assert st == 0
        kat[name][fn] = {"len": len(out), "sha256": sha(out), "hex": out.hex() if len(out) <= 64 else None}
sizes = {fn: [ref.ms_max_compressed_size(f, n) for n in (0, 1, 4096, 65536, 1 << 20)] for fn, f in FMTS.items()}
json.dump({"kat": kat, "max_compressed_size": {"n": [0, 1, 4096, 65536, 1 << 20], **sizes}},
          open(os.path.join(ROOT, "tests/golden/kat.json"), "w"), indent=1)

# 2. corpus slices: sha256 of the reference's output for every codec (inputs come from the deterministic generator)
N = 1_000_000
cor = {}
for i, name in enumerate(corpus.NAMES):
    data = corpus.file_bytes(i, N).tobytes()
    e = {"input_sha256": sha(data), "input_len": N}
    for fn, f in FMTS.items():
        st, out = loader.ref_compress(f, data)
        assert st == 0
        e[fn] = {"len": len(out), "sha256": sha(out)}
    # Xpress as independent 64 KiB units (BASELINE config 3): concatenation of the unit outputs
    cat = b"".join(loader.ref_compress(3, data[o:o + 65536])[1] for o in range(0, N, 65536))
    e["xpress_units64k"] = {"len": len(cat), "sha256": sha(cat)}
    cor[name] = e
mixed = cases.mixed_buffer()
e = {"input_sha256": sha(mixed), "input_len": len(mixed)}
for fn, f in FMTS.items():
    st, out = loader.ref_compress(f, mixed); assert st == 0
    e[fn] = {"len": len(out), "sha256": sha(out)}
cor["mixed_buffer"] = e
json.dump(cor, open(os.path.join(ROOT, "tests/golden/corpus_1mb.json"), "w"), indent=1)

# 3. edge-case families: one digest per codec over all outputs (523-ish seeded inputs)
fam = {}
units = cases.edge_cases()
for fn, f in FMTS.items():
    h = hashlib.sha256(); tot = 0
    for u in units:
        st, out = loader.ref_compress(f, u); assert st == 0
        h.update(len(out).to_bytes(8, "little")); h.update(out); tot += len(out)
    fam[fn] = {"units": len(units), "total_len": tot, "sha256": h.hexdigest()}
json.dump(fam, open(os.path.join(ROOT, "tests/golden/edge_families.json"), "w"), indent=1)
print("golden fixtures written:", {k: v["len"] for k, v in cor["mozilla"].items() if isinstance(v, dict)})

# 4. decoders: status and output of the REFERENCE's ms_decompress for the stream families of tests/cases.py::decode_streams
# (valid / terminated / truncated / concatenated / corrupted, several capacities): one digest per codec. Streams on which the
# reference is undefined (flagged by the restated decoder, oracle/mscomp_oracle.c) are left out and counted.
dec = {}
for fn, f in FMTS.items():
    streams = cases.decode_streams(f, lambda d: loader.ref_compress(f, d)[1])
    h = hashlib.sha256(); asked = 0; by_status = {}
    for stream, cap in streams:
        if loader.oracle_decompress_ex(f, stream, cap)[2]:
            continue
        st, out = loader.ref_decompress(f, stream, cap)
        h.update(st.to_bytes(4, "little", signed=True)); h.update(len(out).to_bytes(8, "little")); h.update(out)
        asked += 1; by_status[str(st)] = by_status.get(str(st), 0) + 1
    dec[fn] = {"streams": len(streams), "asked": asked, "by_status": by_status, "sha256": h.hexdigest()}
json.dump(dec, open(os.path.join(ROOT, "tests/golden/decode_streams.json"), "w"), indent=1)
print("decoder fixture:", {k: (v["asked"], v["by_status"]) for k, v in dec.items()})

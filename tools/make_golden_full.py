"""Generate tests/golden/corpus_full.json: size + SHA-256 of what the REAL reference (oracle/_ref, compiled from
/root/reference/src by oracle/Makefile) writes for every FULL member of the Silesia-shaped corpus (211 938 580 B):
  lznt1            ms_compress(LZNT1) of the whole file (= BASELINE configs[1] for mozilla, configs[4] per file)
  xpress_huff      ms_compress(XPRESS_HUFF) of the whole file (file mode, configs[3])
  xpress_units64k  the file cut into independent 64 KiB units, ms_compress(XPRESS) each, outputs concatenated (configs[2]);
                   also the per-unit lengths' digest so that a wrong unit is localised
  xpress           ms_compress(XPRESS) of the whole file as ONE stream
  lznt1_sa         ms_compress(LZNT1) of the whole file by the reference built with -DMSCOMP_WITH_LZNT1_SA_DICT (SURVEY 8f-4;
                   oracle/_ref/libMSCompression_sa.so). `python tools/make_golden_full.py sa` adds only these to the existing file.
Dev container only (the reference does not travel); the JSON is data and does.   python tools/make_golden_full.py
"""
import hashlib, json, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loader
from ms_compress_amd import corpus

ref = loader.load_ref()
assert ref is not None, "oracle/_ref missing: run make -C oracle"
sha = lambda b: hashlib.sha256(b).hexdigest()


SA_ONLY = len(sys.argv) > 1 and sys.argv[1] == "sa"
OUT = os.path.join(ROOT, "tests/golden/corpus_full.json")
OLD = json.load(open(OUT)) if SA_ONLY else None
assert loader.load_ref_sa() is not None, "oracle/_ref/libMSCompression_sa.so missing: run make -C oracle"


def one(i):
    data = corpus.file_bytes(i).tobytes()
    st, out = loader.ref_compress_sa(data)
    assert st == 0
    sa = {"len": len(out), "sha256": sha(out)}
    if SA_ONLY:
        e = OLD[corpus.NAMES[i]]
        assert e["input_sha256"] == sha(data) and sa["len"] == e["lznt1"]["len"]      # same token lengths, other offsets
        e["lznt1_sa"] = sa
        print(corpus.NAMES[i], sa["len"], sa["sha256"] != e["lznt1"]["sha256"], flush=True)
        return corpus.NAMES[i], e
    e = {"input_sha256": sha(data), "input_len": len(data), "lznt1_sa": sa}
    for name, f in (("lznt1", 2), ("xpress_huff", 4), ("xpress", 3)):
        st, out = loader.ref_compress(f, data)
        assert st == 0
        e[name] = {"len": len(out), "sha256": sha(out)}
    outs = [loader.ref_compress(3, data[o:o + 65536])[1] for o in range(0, len(data), 65536)]
    e["xpress_units64k"] = {"units": len(outs), "len": sum(len(o) for o in outs), "sha256": sha(b"".join(outs)),
                            "unit_lens_sha256": sha(np.array([len(o) for o in outs], dtype=np.uint32).tobytes())}
    print(corpus.NAMES[i], {k: v["len"] for k, v in e.items() if isinstance(v, dict)}, flush=True)
    return corpus.NAMES[i], e


with ThreadPoolExecutor(8) as ex:          # ctypes releases the GIL inside the reference's calls
    doc = dict(ex.map(one, range(12)))
json.dump(doc, open(OUT, "w"), indent=1)

"""tests/golden/huff_lengths.json: SHA-256 over the code lengths the REAL reference's HuffmanEncoder<15,512>::CreateCodes gives
(oracle/_ref/huff_ref = the reference's header instantiated in place) for the seeded histograms of tests/cases.py::huff_histograms,
plus how many of them run the > 15-bit rescale loop. Dev container only.   python tools/make_golden_huff.py"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
HUFF = os.path.join(ROOT, "oracle", "_ref", "huff_ref")
hs = cases.huff_histograms()
h = hashlib.sha256(); per = []
for c in hs:
    out = subprocess.run([HUFF, "fast"], input=" ".join(map(str, c)), capture_output=True, text=True, check=True).stdout.split()
    lens = bytes(int(x) for x in out)
    assert len(lens) == 512
    h.update(lens); per.append(hashlib.sha256(lens).hexdigest()[:16])
json.dump({"cases": len(hs), "sha256_all": h.hexdigest(), "sha256_16_per_case": per},
          open(os.path.join(ROOT, "tests/golden/huff_lengths.json"), "w"), indent=1)
print(len(hs), h.hexdigest())

"""CPU: the oracle (our C restatement) against the committed golden vectors generated from the real reference
(tools/make_golden.py) and the SURVEY.md 8c known-answer table; decoders round-trip."""
import hashlib
import json
import os

import pytest

import cases

G = os.path.join(os.path.dirname(__file__), "golden")
FMTS = {"lznt1": 2, "xpress": 3, "xpress_huff": 4}
sha = lambda b: hashlib.sha256(b).hexdigest()

KAT_INPUTS = {"empty": b"", "a": b"a", "abc": b"abc", "a-z": bytes(range(97, 123)), "abc*100": b"abc" * 100,
              "zeros4096": bytes(4096), "zeros4097": bytes(4097), "zeros70000": bytes(70000)}
# SURVEY.md 8c, hex = full output of the reference
SURVEY_KAT = {("empty", "xpress"): "ffffffff", ("a", "lznt1"): "003061", ("a", "xpress"): "ffffff7f61",
              ("abc", "lznt1"): "0230616263", ("abc", "xpress"): "ffffff1f616263",
              ("abc*100", "lznt1"): "05b0086162632621", ("abc*100", "xpress"): "ffffff1761626317000fff250163",
              ("zeros4096", "lznt1"): "03b00200fc0f", ("zeros4096", "xpress"): "ffffff5f0007000ffffb0f00",
              ("zeros4097", "lznt1"): "03b00200fc0f003000", ("zeros4097", "xpress"): "ffffff5f0007000ffffc0f00",
              ("zeros70000", "xpress"): "ffffff5f0007000fff00006b11010000"}


@pytest.mark.parametrize("fmt", list(FMTS))
def test_kat(oracle, fmt):
    kat = json.load(open(os.path.join(G, "kat.json")))["kat"]
    for name, data in KAT_INPUTS.items():
        st, out = oracle.oracle_compress(FMTS[fmt], data)
        e = kat[name][fmt]
        assert st == 0 and len(out) == e["len"] and sha(out) == e["sha256"], (name, fmt)
        if e["hex"] is not None:
            assert out.hex() == e["hex"]
        if (name, fmt) in SURVEY_KAT:
            assert out.hex() == SURVEY_KAT[(name, fmt)]


def test_max_compressed_size(oracle):
    g = json.load(open(os.path.join(G, "kat.json")))["max_compressed_size"]
    lib = oracle.load_oracle()
    for fmt, f in FMTS.items():
        assert [lib.orc_max_compressed_size(f, n) for n in g["n"]] == g[fmt]
    assert g["lznt1"] == [3, 6, 4101, 65571, 1049091] and g["xpress"] == [4, 5, 4612, 73732, 1179652]
    assert g["xpress_huff"] == [292, 293, 4388, 66086, 1052996]
    assert lib.orc_max_compressed_size(1, 10) == 2 ** 64 - 1 and lib.orc_max_compressed_size(5, 10) == 2 ** 64 - 1


@pytest.mark.parametrize("fmt", list(FMTS))
def test_corpus_golden(oracle, fmt):
    from ms_compress_amd import corpus
    g = json.load(open(os.path.join(G, "corpus_1mb.json")))
    for i, name in enumerate(corpus.NAMES):
        data = corpus.file_bytes(i, g[name]["input_len"]).tobytes()
        assert sha(data) == g[name]["input_sha256"], "corpus generator drifted: " + name
        st, out = oracle.oracle_compress(FMTS[fmt], data)
        assert st == 0 and len(out) == g[name][fmt]["len"] and sha(out) == g[name][fmt]["sha256"], (name, fmt)
        st, back = oracle.oracle_decompress(FMTS[fmt], out, len(data))
        assert st == 0 and back == data
    data = cases.mixed_buffer()
    st, out = oracle.oracle_compress(FMTS[fmt], data)
    assert st == 0 and sha(out) == g["mixed_buffer"][fmt]["sha256"]


@pytest.mark.parametrize("fmt", list(FMTS))
def test_edge_families_golden(oracle, fmt):
    g = json.load(open(os.path.join(G, "edge_families.json")))[fmt]
    h = hashlib.sha256()
    units = cases.edge_cases()
    assert len(units) == g["units"]
    tot = 0
    for u in units:
        st, out = oracle.oracle_compress(FMTS[fmt], u)
        assert st == 0
        h.update(len(out).to_bytes(8, "little")); h.update(out); tot += len(out)
        st, back = oracle.oracle_decompress(FMTS[fmt], out, len(u))
        if fmt == "xpress" and len(u) == 0:      # the reference's Xpress decoder rejects the 4-byte stream its encoder writes for no input
            assert st == -3                      # (xpress_decompress.cpp:414-418)
        else:
            assert st == 0 and back == u, (fmt, len(u))
    assert tot == g["total_len"] and h.hexdigest() == g["sha256"]


def test_buf_error_and_capacity(oracle):
    data = cases.mixed_buffer()[:30000]
    for f in FMTS.values():
        st, out = oracle.oracle_compress(f, data)
        assert st == 0
        assert oracle.oracle_compress(f, data, cap=len(out))[1] == out
        assert oracle.oracle_compress(f, data, cap=len(out) - 1)[0] == -5      # MSCOMP_BUF_ERROR


def test_units_driver_threads(oracle):
    import ctypes as C
    import numpy as np
    lib = oracle.load_oracle()
    units = cases.edge_cases(sizes=[0, 1, 100, 5000, 70000], kinds=["words", "lz"])
    blob = b"".join(units)
    in_off = np.cumsum([0] + [len(u) for u in units]).astype(np.uint64)
    caps = [lib.orc_max_compressed_size(4, len(u)) for u in units]
    out_off = np.cumsum([0] + caps).astype(np.uint64)
    out = np.zeros(int(out_off[-1]) + 8, dtype=np.uint8)
    out_len = np.zeros(len(units), dtype=np.uint64)
    status = np.zeros(len(units), dtype=np.int32)
    for fmt in (2, 3, 4):
        lib.orc_compress_units(fmt, blob, in_off.ctypes.data, len(units), out.ctypes.data, out_off.ctypes.data,
                               out_len.ctypes.data, status.ctypes.data, 4)
        for i, u in enumerate(units):
            exp = oracle.oracle_compress(fmt, u)[1]
            assert status[i] == 0 and bytes(out[int(out_off[i]): int(out_off[i]) + int(out_len[i])]) == exp


@pytest.mark.parametrize("fmt", list(FMTS))
def test_decoders_golden(oracle, fmt):
    """The restated decoders against the committed digest of the reference's ms_decompress over the decode stream families
    (status + bytes of ~900 valid / truncated / concatenated / corrupted streams per codec)."""
    g = json.load(open(os.path.join(G, "decode_streams.json")))[fmt]
    f = FMTS[fmt]
    streams = cases.decode_streams(f, lambda d: oracle.oracle_compress(f, d)[1])
    assert len(streams) == g["streams"]
    h = hashlib.sha256(); asked = 0
    for stream, cap in streams:
        st, out, undefined = oracle.oracle_decompress_ex(f, stream, cap)
        if undefined:
            continue
        h.update(st.to_bytes(4, "little", signed=True)); h.update(len(out).to_bytes(8, "little")); h.update(out)
        asked += 1
    assert asked == g["asked"] and h.hexdigest() == g["sha256"]


def test_oracle_huffman_lengths_match_the_reference_fixture(oracle):
    """orc_huff_lengths (the restated CreateCodes) against the digest oracle/_ref/huff_ref (the reference's own header) left in
    tests/golden/huff_lengths.json for the seeded histograms -- the rescale loop and heavy ties included."""
    import hashlib
    import numpy as np
    import cases
    gold = json.load(open(os.path.join(G, "huff_lengths.json")))
    lib = oracle.load_oracle()
    hs = cases.huff_histograms()
    assert len(hs) == gold["cases"]
    allh = hashlib.sha256()
    for i, c in enumerate(hs):
        a = np.asarray(c, dtype=np.uint32); lens = np.zeros(512, dtype=np.uint8)
        lib.orc_huff_lengths(a.ctypes.data, lens.ctypes.data)
        assert hashlib.sha256(lens.tobytes()).hexdigest()[:16] == gold["sha256_16_per_case"][i], i
        allh.update(lens.tobytes())
    assert allh.hexdigest() == gold["sha256_all"]


def test_lznt1_sa_dictionary_flavour_golden(oracle):
    """the oracle's suffix-array dictionary (LZNT1Dictionary_SA.h:404-476 restated) against what the reference BUILT WITH
    MSCOMP_WITH_LZNT1_SA_DICT wrote (tests/golden/lznt1_sa.json, tools/make_golden_sa.py); the default decoder reads it back."""
    from ms_compress_amd import corpus
    g = json.load(open(os.path.join(G, "lznt1_sa.json")))
    assert any(k["differs_from_default"] for k in g["kat"].values())
    for name, k in g["kat"].items():
        data = bytes.fromhex(k["input_hex"]) if k["input_hex"] is not None else {"abc*100": b"abc" * 100, "zeros4097": bytes(4097)}[name]
        st, out = oracle.oracle_compress_sa(data)
        assert st == 0 and out.hex() == k["hex"], name
        assert (out != oracle.oracle_compress(2, data)[1]) == k["differs_from_default"]
    for i, name in enumerate(corpus.NAMES):
        e = g["corpus"][name]
        data = corpus.file_bytes(i, e["input_len"]).tobytes()
        assert sha(data) == e["input_sha256"]
        st, out = oracle.oracle_compress_sa(data)
        assert st == 0 and len(out) == e["len"] and sha(out) == e["sha256"], name
        st, back = oracle.oracle_decompress(2, out, len(data))
        assert st == 0 and back == data
    h = hashlib.sha256(); tot = 0
    for u in cases.edge_cases():
        st, out = oracle.oracle_compress_sa(u)
        assert st == 0
        h.update(len(out).to_bytes(8, "little")); h.update(out); tot += len(out)
    assert tot == g["edge_families"]["total_len"] and h.hexdigest() == g["edge_families"]["sha256"]

"""-m gpu: BASELINE configs[4] at full size on one GPU -- the 12-file corpus replicated 16x (192 files, 3 391 017 280 B) for all
three codecs, the batch bench.py shards over the ranks. Every file of replica 0 is compared with the digest the REAL reference gave
for it (tests/golden/corpus_full.json, written by tools/make_golden_full.py from oracle/_ref); every other replica must hold exactly
replica 0's bytes (compared on the device after mscomp_amd_compact_batch packed the outputs back to back)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPS = 16
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "corpus_full.json")))
KEY = {2: "lznt1", 3: "xpress_units64k", 4: "xpress_huff"}


def corpus_units(fmt, flen, foff):
    if fmt != 3:
        return foff, flen
    offs, lens = [], []
    for o, l in zip(foff, flen):
        s = np.arange(0, int(l), 65536, dtype=np.uint64)
        offs.append(s + o); lens.append(np.minimum(65536, int(l) - s).astype(np.uint64))
    return np.concatenate(offs), np.concatenate(lens)


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_config5_replicated_corpus(gpu_ctx, fmt):
    import torch
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    files = [corpus.file_bytes(i) for i in range(12)]
    for f, name in zip(files, corpus.NAMES):
        assert hashlib.sha256(f.tobytes()).hexdigest() == GOLD[name]["input_sha256"], "the corpus generator changed: regenerate the fixtures"
    flen = np.array([len(f) for f in files], dtype=np.uint64)
    foff = np.zeros(12, dtype=np.uint64); foff[1:] = np.cumsum(flen)[:-1]
    total = int(flen.sum())
    uoff, ulen = corpus_units(fmt, flen, foff)
    nu = len(ulen)
    in_off = np.concatenate([uoff + np.uint64(r * total) for r in range(REPS)])
    in_len = np.tile(ulen, REPS)
    assert int(in_len.sum()) == 3391017280 and len(in_len) == {2: 192, 3: 51824, 4: 192}[fmt]
    dev = torch.device("cuda", gpu_ctx.device)
    one = torch.from_numpy(np.concatenate(files)).to(dev)
    d_in = torch.cat([one] * REPS + [torch.zeros(16, dtype=torch.uint8, device=dev)])
    del one
    caps = np.array([m.max_compressed_size(fmt, int(x)) + 2 for x in in_len], dtype=np.uint64)
    out_off, out_total = m.pack_offsets(caps)
    d_out = torch.empty(out_total + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(len(in_len), dtype=torch.int64, device=dev)
    d_st = torch.full((len(in_len),), -9, dtype=torch.int32, device=dev)
    plan = m.Plan(gpu_ctx, fmt, in_off, in_len, out_off, caps)
    plan.execute(d_in, d_out, d_len, d_st)
    torch.cuda.synchronize()
    plan.close()
    assert bool((d_st == 0).all().item())
    d_packed, d_poff = m.compact_batch(gpu_ctx, out_off, caps, d_out, d_len)
    torch.cuda.synchronize()
    poff = d_poff.cpu().numpy()
    lens = d_len.cpu().numpy()
    # replica 0, file by file, against the reference's digests
    first = d_packed[: int(poff[nu])].cpu().numpy()
    u = 0
    for i, name in enumerate(corpus.NAMES):
        k = 1 if fmt != 3 else (int(flen[i]) + 65535) // 65536
        a, b = int(poff[u]), int(poff[u + k])
        g = GOLD[name][KEY[fmt]]
        assert b - a == g["len"], (name, b - a, g["len"])
        assert hashlib.sha256(first[a:b].tobytes()).hexdigest() == g["sha256"], name
        if fmt == 3:
            assert hashlib.sha256(lens[u:u + k].astype(np.uint32).tobytes()).hexdigest() == g["unit_lens_sha256"], name
        u += k
    assert u == nu
    # replicas 1..15: the same lengths and the same bytes as replica 0
    span = int(poff[nu])
    for r in range(1, REPS):
        assert np.array_equal(lens[r * nu:(r + 1) * nu], lens[:nu]), r
        assert int(poff[(r + 1) * nu]) - int(poff[r * nu]) == span
        assert bool(torch.equal(d_packed[int(poff[r * nu]): int(poff[r * nu]) + span], d_packed[:span])), r


def test_lznt1_sa_dictionary_flavour_full_files(gpu_ctx):
    """SURVEY.md 8f-4 at full size: all 12 files (211 938 580 B, 51 747 chunks) through csrc/lznt1_sa.hip against what the reference built
    with -DMSCOMP_WITH_LZNT1_SA_DICT wrote (corpus_full.json "lznt1_sa"): same lengths as the default flavour, other bytes."""
    import time
    import torch
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    files = [corpus.file_bytes(i) for i in range(12)]
    flen = np.array([len(f) for f in files], dtype=np.uint64)
    foff = np.zeros(12, dtype=np.uint64); foff[1:] = np.cumsum(flen)[:-1]
    dev = torch.device("cuda", gpu_ctx.device)
    d_in = torch.cat([torch.from_numpy(np.concatenate(files)).to(dev), torch.zeros(16, dtype=torch.uint8, device=dev)])
    caps = np.array([m.max_compressed_size(2, int(x)) + 2 for x in flen], dtype=np.uint64)
    out_off, out_total = m.pack_offsets(caps)
    d_out = torch.empty(out_total + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(12, dtype=torch.int64, device=dev)
    d_st = torch.full((12,), -9, dtype=torch.int32, device=dev)
    gpu_ctx.lib.mscomp_amd_set_lznt1_sa_dict(1)
    try:
        plan = m.Plan(gpu_ctx, 2, foff, flen, out_off, caps)
        plan.execute(d_in, d_out, d_len, d_st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.execute(d_in, d_out, d_len, d_st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        plan.close()
    finally:
        gpu_ctx.lib.mscomp_amd_set_lznt1_sa_dict(0)
    print("lznt1 suffix-array flavour: %.1f ms for %d B = %.2f GB/s" % (dt * 1e3, int(flen.sum()), int(flen.sum()) / dt / 1e9))
    assert bool((d_st == 0).all().item())
    lens = d_len.cpu().numpy()
    out = d_out.cpu().numpy()
    for i, name in enumerate(corpus.NAMES):
        g = GOLD[name]["lznt1_sa"]
        assert int(lens[i]) == g["len"] and g["sha256"] != GOLD[name]["lznt1"]["sha256"]
        assert hashlib.sha256(out[int(out_off[i]): int(out_off[i]) + int(lens[i])].tobytes()).hexdigest() == g["sha256"], name

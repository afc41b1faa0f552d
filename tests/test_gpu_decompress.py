"""GPU: the HIP decompressors against the checker (oracle restatement of the reference's one-shot
semantics, itself pinned against the compiled reference in test_oracle_vs_ref.py) and against the compiled reference
where it travelled (oracle/_ref)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
FMTS = {"lznt1": 2, "xpress": 3, "xpress_huff": 4}


@pytest.mark.parametrize("fmt", list(FMTS))
def test_decode_streams_match_checker(oracle, gpu_ctx, fmt):
    """status and bytes of every stream family: valid / terminated / truncated / concatenated / corrupted, all capacities"""
    import ms_compress_amd as m
    f = FMTS[fmt]
    streams = cases.decode_streams(f, lambda d: oracle.oracle_compress(f, d)[1])
    outs, sts = m.decompress_units(f, [s for s, _ in streams], [c for _, c in streams], ctx=gpu_ctx)
    ref = oracle.load_ref()
    for (stream, cap), out, st in zip(streams, outs, sts):
        so, oo, undefined = oracle.oracle_decompress_ex(f, stream, cap)
        assert st == so, (len(stream), cap, st, so)
        if so == 0:
            assert out == oo, (len(stream), cap)
        if ref is not None and not undefined:
            sr, orf = oracle.ref_decompress(f, stream, cap)
            assert (st, out if st == 0 else b"") == (sr, orf), (len(stream), cap)


@pytest.mark.parametrize("fmt", list(FMTS))
def test_round_trip_on_device(oracle, gpu_ctx, fmt):
    """compress on the GPU, decompress on the GPU: the edge units of the compressor tests, in one batch"""
    import ms_compress_amd as m
    f = FMTS[fmt]
    units = cases.edge_cases() + [cases.mixed_buffer()]
    comp, st = m.compress_units(f, units, ctx=gpu_ctx)
    assert all(s == 0 for s in st)
    back, st2 = m.decompress_units(f, comp, [len(u) for u in units], ctx=gpu_ctx)
    for u, b, s2 in zip(units, back, st2):
        if f == 3 and len(u) == 0:                      # the reference's Xpress decoder rejects what its encoder writes for no input
            assert s2 == -3
        else:
            assert s2 == 0 and b == u, len(u)


@pytest.mark.parametrize("fmt", list(FMTS))
def test_one_shot_decompress_host_pointers(oracle, gpu_ctx, fmt):
    import ms_compress_amd as m
    f = FMTS[fmt]
    data = cases.mixed_buffer()
    comp = oracle.oracle_compress(f, data)[1]
    assert m.decompress(f, comp, len(data)) == data
    assert m.decompress(f, comp, len(data) + 77) == data
    with pytest.raises(m.MSCompError) as e:
        m.decompress(f, comp, len(data) - 1)
    assert e.value.status == m.MSCOMP_BUF_ERROR
    bad = comp[:1] + bytes([comp[1] ^ 0x40]) + comp[2:] if f == 2 else comp[:-1]    # LZNT1: wrong signature in the first header; Xpress: cut short
    want = oracle.oracle_decompress_ex(f, bad, len(data))[0]
    assert want != 0
    with pytest.raises(m.MSCompError) as e:
        m.decompress(f, bad, len(data))
    assert e.value.status == want
    assert m.decompress(f, b"", 10) == b""
    assert e.value.status == oracle.ref_decompress(f, bad, len(data))[0] if oracle.load_ref() else True


def test_round_trip_full_size_lznt1(oracle, gpu_ctx):
    """BASELINE configs[1] (mozilla-like, 51 220 480 B as one unit): GPU compress -> GPU decompress gives the input back"""
    import torch
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    data = corpus.file_bytes(corpus.NAMES.index("mozilla"), 51_220_480)
    n = len(data)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    cap = m.max_compressed_size(2, n) + 2
    d_c = torch.zeros(cap + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(1, dtype=torch.int64, device=dev); d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    p = m.Plan(gpu_ctx, 2, [0], [n], [0], [cap]); p.execute(d_in, d_c, d_len, d_st); torch.cuda.synchronize(); p.close()
    clen = int(d_len[0]); assert int(d_st[0]) == 0
    d_back = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
    q = m.Plan(gpu_ctx, 2, [0], [clen], [0], [n], decompress=True)
    for _ in range(2):
        d_len.zero_(); d_st.fill_(-9)
        q.execute(d_c, d_back, d_len, d_st); torch.cuda.synchronize()
        assert int(d_st[0]) == 0 and int(d_len[0]) == n
        assert torch.equal(d_back[:n], d_in)
    q.close()


def test_lznt1_header_walk_falls_back_when_speculation_cannot_decide(oracle, gpu_ctx):
    """Stored chunks full of 0x33 bytes: every offset of such a stream looks like a chunk header (0x3333: signature 011, 822
    bytes), so the speculated chains of a segment do not agree on one landing and the verify kernel walks those segments itself.
    The result must not change; a mixed stream (compressed text, then the hostile part, then text) exercises the hand-over."""
    import ms_compress_amd as m
    hostile = b"".join(b"\xff\x3f" + b"\x33" * 4096 for _ in range(60))           # 60 stored chunks = 245 880 B of input
    text = cases.mixed_buffer()[:200000]
    ctext = oracle.oracle_compress(2, text)[1]
    streams = [hostile, ctext + hostile + ctext, hostile + ctext]
    plain = [b"\x33" * (4096 * 60), text + b"\x33" * (4096 * 60) + text, b"\x33" * (4096 * 60) + text]
    gpu_ctx.lib.mscomp_amd_debug_lzd_walked(gpu_ctx._h)
    outs, sts = m.decompress_units(2, streams, [len(p) for p in plain], ctx=gpu_ctx)
    walked = gpu_ctx.lib.mscomp_amd_debug_lzd_walked(gpu_ctx._h)
    assert sts == [0, 0, 0] and outs == plain
    assert walked >= 3, walked
    for s, p in zip(streams, plain):
        assert oracle.oracle_decompress_ex(2, s, len(p))[:2] == (0, p)
    # and the ordinary corpus needs no fallback
    comp, _ = m.compress_units(2, [cases.mixed_buffer()], ctx=gpu_ctx)
    m.decompress_units(2, comp, [len(cases.mixed_buffer())], ctx=gpu_ctx)
    assert gpu_ctx.lib.mscomp_amd_debug_lzd_walked(gpu_ctx._h) == 0


@pytest.mark.parametrize("fmt", ["xpress", "xpress_huff"])
def test_round_trip_full_size_units(oracle, gpu_ctx, fmt):
    """BASELINE configs[2]/[3] sizes: the 212 MB corpus as 3 239 units of 64 KiB, GPU compress -> GPU decompress gives the
    input back (one wave per stream decodes; every status MSCOMP_OK, every length right)."""
    import torch
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    f = FMTS[fmt]
    files = [corpus.file_bytes(i) for i in range(12)]
    blob = np.concatenate(files)
    offs, lens = [], []
    pos = 0
    for fl in files:
        s = np.arange(0, len(fl), 65536, dtype=np.uint64)
        offs.append(s + np.uint64(pos)); lens.append(np.minimum(65536, len(fl) - s).astype(np.uint64)); pos += len(fl)
    in_off, in_len = np.concatenate(offs), np.concatenate(lens)
    caps = [m.max_compressed_size(f, int(x)) + 2 for x in in_len]
    c_off, c_total = m.pack_offsets(caps)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(blob).to(dev); d_c = torch.zeros(c_total + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(len(in_len), dtype=torch.int64, device=dev); d_st = torch.zeros(len(in_len), dtype=torch.int32, device=dev)
    p = m.Plan(gpu_ctx, f, in_off, in_len, c_off, caps); p.execute(d_in, d_c, d_len, d_st); torch.cuda.synchronize(); p.close()
    assert bool((d_st == 0).all())
    clen = d_len.cpu().numpy().astype(np.uint64)
    d_back = torch.zeros(len(blob) + 16, dtype=torch.uint8, device=dev)
    d_len2 = torch.zeros_like(d_len); d_st2 = torch.full_like(d_st, -9)
    q = m.Plan(gpu_ctx, f, c_off, clen, in_off, in_len, decompress=True); q.execute(d_c, d_back, d_len2, d_st2); torch.cuda.synchronize(); q.close()
    assert bool((d_st2 == 0).all()) and torch.equal(d_len2.cpu(), torch.from_numpy(in_len.astype(np.int64)))
    assert torch.equal(d_back[: len(blob)], d_in)


def test_one_shot_lznt1_with_a_very_generous_capacity(oracle, gpu_ctx):
    """*out_len may be far larger than anything the stream can produce: the device staging is sized by the input, not by the capacity"""
    import ctypes as C
    import ms_compress_amd as m
    data = cases.mixed_buffer()[:50000]
    comp = oracle.oracle_compress(2, data)[1]
    lib = m.load_library()
    out = C.create_string_buffer(len(data) + 16)
    n = C.c_size_t(1 << 40)                                          # "a terabyte of room" (the buffer behind it is only as large as needed)
    assert lib.ms_decompress(2, comp, len(comp), out, C.byref(n)) == 0 and n.value == len(data) and out.raw[: len(data)] == data


def _dense_matches(rnd, n):
    """bytes whose Xpress stream is packed with matches of every length form: 3-9 (in the symbol), 10-24 (shared nibble), 25-279 (+ a byte),
    280-65557 (+ 16 bits) and longer (+ 32 bits), a few literals between them"""
    out = bytearray(rnd.randbytes(40))
    while len(out) < n:
        if rnd.random() < 0.25:
            out += rnd.randbytes(rnd.randint(1, 4))
        form = 4 if (n >= 100000 and rnd.random() < 0.02) else rnd.choice((0, 0, 1, 1, 1, 2, 2, 3))
        ln = (rnd.randint(3, 9), rnd.randint(10, 24), rnd.randint(25, 279), rnd.randint(280, 3000), rnd.randint(65558, 70000))[form]
        off = rnd.randint(1, min(len(out), 8192))
        for _ in range(ln):
            out.append(out[-off])
    return bytes(out)


@pytest.mark.parametrize("mode", [0, 1])
def test_xpress_length_forms_truncations_and_corruptions(oracle, gpu_ctx, mode):
    """The token-parallel Xpress parser (mode 0: 32 tokens of a flag word at a time, the positions behind nibble-bearing matches by iteration)
    and the token-at-a-time kernel (mode 1) on streams dense with long matches: valid (exact, short and generous capacities), EVERY prefix of
    one stream (each way the input can end inside a token), and random corruptions -- status and bytes against the checker."""
    import random
    import ms_compress_amd as m
    rnd = random.Random(77)
    datas = [_dense_matches(rnd, rnd.choice((300, 2000, 20000))) for _ in range(24)] + [_dense_matches(rnd, 200000)]
    comps = [oracle.oracle_compress(3, d)[1] for d in datas]
    streams = []
    for d, c in zip(datas, comps):
        streams += [(c, len(d)), (c, len(d) + 7), (c, len(d) - 1), (c, len(d) // 2)]
    d0, c0 = datas[2], comps[2]
    streams += [(c0[:k], len(d0)) for k in range(len(c0))]
    for d, c in zip(datas[:12], comps[:12]):
        for _ in range(25):
            b = bytearray(c)
            for _ in range(rnd.randint(1, 3)):
                b[rnd.randrange(len(b))] = rnd.randrange(256)
            streams.append((bytes(b), len(d) + rnd.choice((0, 0, 100, 70000))))
    gpu_ctx.lib.mscomp_amd_debug_set_xpress_decoder(mode)
    try:
        outs, sts = m.decompress_units(3, [s for s, _ in streams], [c for _, c in streams], ctx=gpu_ctx)
    finally:
        gpu_ctx.lib.mscomp_amd_debug_set_xpress_decoder(0)
    n_ok = n_err = 0
    for (stream, cap), out, st in zip(streams, outs, sts):
        so, oo, undefined = oracle.oracle_decompress_ex(3, stream, cap)
        assert not undefined
        assert st == so, (len(stream), cap, st, so)
        if so == 0:
            assert out == oo, (len(stream), cap)
            n_ok += 1
        else:
            n_err += 1
    assert n_ok > 100 and n_err > 300


@pytest.mark.parametrize("fmt", ["xpress", "xpress_huff"])
def test_large_units_get_their_bytes_from_all_cus(oracle, gpu_ctx, fmt):
    """csrc/lzglobal.hip (units with room for 1 MiB or more: tile directory, per-tile expansion, pointer passes) next to the block-per-unit
    and wave-per-unit kernels in ONE batch: deep copy chains (5 MB of one byte: every byte copies its neighbour; short periods), data whose
    sources lie far back, a generous capacity (tiles behind the end of the output), a unit that ends in an error, small units in between."""
    import ctypes as C
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    f = FMTS[fmt]
    rng = np.random.default_rng(31)
    per7 = np.tile(rng.integers(0, 256, 7, dtype=np.uint8), 3_000_000 // 7 + 1)[:3_000_000].tobytes()
    far = bytearray(rng.integers(0, 256, 40000, dtype=np.uint8).tobytes())
    while len(far) < 2_500_000:                                   # copies of 100 - 4000 bytes from up to 8000 back, over and over: chains through hundreds of tiles
        o = int(rng.integers(2000, 8000)); n = int(rng.integers(100, 4000)); far += far[-o:-o + n]
    units = [bytes(5_000_000), per7, bytes(far), corpus.file_bytes(1, 4_000_000).tobytes(), corpus.file_bytes(3, 1_500_000).tobytes(),
             b"small unit " * 50, corpus.file_bytes(0, 300_000).tobytes(), rng.integers(0, 256, 2_000_000, dtype=np.uint8).tobytes()]
    comp, st = m.compress_units(f, units, ctx=gpu_ctx)
    assert all(s == 0 for s in st)
    caps = [len(u) for u in units]
    caps[1] += 3 << 20                                            # room for 3 MiB more than comes out
    streams = list(comp) + [comp[3][: len(comp[3]) // 2], comp[3], comp[4]]   # half a stream: an error (or a shorter output) next to the good ones;
    caps += [len(units[3]), len(units[3]) - 1, len(units[4]) // 3]            # and two whose output does not fit
    back, st2 = m.decompress_units(f, streams, caps, ctx=gpu_ctx)
    # (the counters of THIS batch: read before the next decompression overwrites them)
    opened = (C.c_uint32 * 33)()
    gpu_ctx.lib.mscomp_amd_debug_lzg_open.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    assert gpu_ctx.lib.mscomp_amd_debug_lzg_open(gpu_ctx._h, sum(c + 64 for c in caps if c >= (1 << 20)), opened) == 0
    opened = list(opened)
    assert opened[0] > 1_000_000 and 0 in opened and all(x == 0 for x in opened[opened.index(0):])   # the path ran, took several passes, and ended
    for i, u in enumerate(units):
        assert st2[i] == 0 and back[i] == u, (fmt, i, st2[i], len(back[i]), len(u))
    for j in (-3, -2, -1):
        so, oo, _ = oracle.oracle_decompress_ex(f, streams[j], caps[j])
        assert st2[j] == so and (so != 0 or back[j] == oo), (fmt, j, st2[j], so)
    assert st2[-2] == -5 and st2[-1] == -5
    # large streams with a few bytes changed, anywhere: whatever the segment walks make of them, status and bytes are those of the checker
    import random
    rnd = random.Random(5)
    bad = []
    for src_i in (3, 4, 2):
        for _ in range(6):
            b = bytearray(comp[src_i])
            for _ in range(rnd.randint(1, 4)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
            bad.append((bytes(b), len(units[src_i]) + rnd.choice((0, 0, 4096))))
    outs, sts = m.decompress_units(f, [b for b, _ in bad], [c for _, c in bad], ctx=gpu_ctx)
    n_judged = 0
    for (stream, cap), out, st in zip(bad, outs, sts):
        so, oo, undefined = oracle.oracle_decompress_ex(f, stream, cap)
        if not undefined:
            assert st == so and (so != 0 or out == oo), (fmt, len(stream), cap, st, so)
            n_judged += 1
    assert n_judged >= 12


def test_segment_and_tile_edges_of_large_streams(oracle, gpu_ctx):
    """Xpress streams whose length sits on the edges of the segment walk (csrc/decompress.hip xps_*: streams from 512 KiB, 16 KiB segments) and
    outputs on the edges of the byte stage (csrc/lzglobal.hip: capacities from 1 MiB, 8 KiB tiles): prefixes of valid streams cut at those
    lengths, 600 KB of random bytes as a "stream", capacities one byte around 1 MiB -- status and bytes of the checker in every case."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    rng = np.random.default_rng(41)
    base = corpus.file_bytes(1, 3_000_000).tobytes()                       # mozilla: ~1.6 MB of stream
    noise = rng.integers(0, 256, 1_400_000, dtype=np.uint8).tobytes()      # incompressible: the stream is longer than the data
    comp, st = m.compress_units(3, [base, noise], ctx=gpu_ctx)
    assert all(s == 0 for s in st) and len(comp[0]) > 1_200_000 and len(comp[1]) > 1_400_000
    cases_ = []
    for c, n in ((comp[0], len(base)), (comp[1], len(noise))):
        for cut in (524287, 524288, 524289, 540672, 540673, 16384 * 40 - 1, 16384 * 40, len(c) - 1, len(c)):
            cases_.append((c[:cut], n))
    cases_ += [(rng.integers(0, 256, 600_000, dtype=np.uint8).tobytes(), 2_000_000), (bytes(700_000), 1 << 20), (b"\xff" * 600_000, 3_000_000)]
    for cap in ((1 << 20) - 1, 1 << 20, (1 << 20) + 1, (1 << 20) + 8192, len(base)):
        cases_.append((comp[0], cap))
    outs, sts = m.decompress_units(3, [s for s, _ in cases_], [c for _, c in cases_], ctx=gpu_ctx)
    n_ok = 0
    for (stream, cap), out, st in zip(cases_, outs, sts):
        so, oo, undefined = oracle.oracle_decompress_ex(3, stream, cap)
        assert not undefined
        assert st == so and (so != 0 or out == oo), (len(stream), cap, st, so)
        n_ok += so == 0
    assert n_ok >= 3
    # the same capacities for Xpress+Huffman (chunk-parallel walk + byte stage)
    comp4, st = m.compress_units(4, [base], ctx=gpu_ctx)
    caps = [(1 << 20) - 1, 1 << 20, (1 << 20) + 1, len(base), len(base) + (1 << 20)]
    outs, sts = m.decompress_units(4, [comp4[0]] * len(caps), caps, ctx=gpu_ctx)
    for cap, out, st in zip(caps, outs, sts):
        so, oo, _ = oracle.oracle_decompress_ex(4, comp4[0], cap)
        assert st == so and (so != 0 or out == oo), (cap, st, so)


@pytest.mark.parametrize("fmt", ["xpress", "xpress_huff"])
def test_a_unit_with_room_for_more_than_4_gib_next_to_large_units(oracle, gpu_ctx, fmt):
    """the byte stage of csrc/lzglobal.hip indexes a unit's output with 32 bits: a unit with a capacity of 4 GiB or more keeps the whole plan on the
    block-per-unit kernel (it used to be left without bytes when another large unit took the all-CU stage)"""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    f = FMTS[fmt]
    units = [corpus.file_bytes(1, 2_500_000).tobytes(), corpus.file_bytes(3, 400_000).tobytes(), b"abc" * 1000]
    comp, st = m.compress_units(f, units, ctx=gpu_ctx)
    assert all(s == 0 for s in st)
    caps = [len(units[0]), 5 << 30, len(units[2])]
    back, st2 = m.decompress_units(f, comp, caps, ctx=gpu_ctx)
    assert list(st2) == [0, 0, 0] and all(b == u for b, u in zip(back, units))

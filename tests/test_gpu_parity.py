"""-m gpu: the HIP path, called through the C-ABI, against the oracle (bit-exact) and the committed golden fixtures."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
FMTS = {"lznt1": 2, "xpress": 3, "xpress_huff": 4}
sha = lambda b: hashlib.sha256(b).hexdigest()


def _check_units(m, oracle, fmt, units, ctx):
    got, st = m.compress_units(fmt, units, ctx=ctx)
    for i, (u, g, s) in enumerate(zip(units, got, st)):
        es, exp = oracle.oracle_compress(fmt, u)
        assert es == 0 and s == 0, (i, len(u), s)
        assert g == exp, "fmt %d unit %d (len %d): GPU bytes differ from the oracle (%d vs %d B)" % (fmt, i, len(u), len(g), len(exp))


@pytest.mark.parametrize("fmt", list(FMTS))
def test_edge_families(oracle, gpu_ctx, fmt):
    """empty / tiny / ragged / chunk-boundary sizes x {random, 2-symbol, run, words, synthetic LZ} (SURVEY.md 8c fuzz classes)."""
    import ms_compress_amd as m
    units = cases.edge_cases()
    _check_units(m, oracle, FMTS[fmt], units, gpu_ctx)
    # and against the committed digest generated from the real reference
    g = json.load(open(os.path.join(G, "edge_families.json")))[fmt]
    got, _ = m.compress_units(FMTS[fmt], units, ctx=gpu_ctx)
    h = hashlib.sha256()
    for o in got:
        h.update(len(o).to_bytes(8, "little")); h.update(o)
    assert h.hexdigest() == g["sha256"]


@pytest.mark.parametrize("fmt", list(FMTS))
def test_corpus_golden(gpu_ctx, fmt):
    """1 MB of every corpus member + the mixed buffer (100 000 zeros: Xpress lagging fill, >64 KiB match; 70 000 random
    bytes: XH fallback) against SHA-256 of the REFERENCE's output (tests/golden, tools/make_golden.py)."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    g = json.load(open(os.path.join(G, "corpus_1mb.json")))
    names = corpus.NAMES + ["mixed_buffer"]
    units = [corpus.file_bytes(i, g[n]["input_len"]).tobytes() for i, n in enumerate(corpus.NAMES)] + [cases.mixed_buffer()]
    got, st = m.compress_units(FMTS[fmt], units, ctx=gpu_ctx)
    for n, o, s in zip(names, got, st):
        assert s == 0 and len(o) == g[n][fmt]["len"] and sha(o) == g[n][fmt]["sha256"], (n, fmt)


def test_xpress_units_64k_golden(gpu_ctx):
    """BASELINE config 3: every 64 KiB slice an independent Xpress stream."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    g = json.load(open(os.path.join(G, "corpus_1mb.json")))
    for i, n in enumerate(corpus.NAMES[:4]):
        data = corpus.file_bytes(i, g[n]["input_len"]).tobytes()
        got, st = m.compress_units(3, [data[o:o + 65536] for o in range(0, len(data), 65536)], ctx=gpu_ctx)
        cat = b"".join(got)
        assert all(s == 0 for s in st) and len(cat) == g[n]["xpress_units64k"]["len"] and sha(cat) == g[n]["xpress_units64k"]["sha256"]


@pytest.mark.parametrize("mode", [(0x2000, 0), (0xFFFF, 1)])
def test_match_finder_stage(oracle, gpu_ctx, mode):
    """Stage-level parity: per-position (length capped at 48, offset) of the HIP hash-chain finder == XpressDictionary::Find."""
    import torch
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    lib, orc = m.load_library(), oracle.load_oracle()
    max_off, clip = mode
    for data in [corpus.file_bytes(1, 300_000).tobytes(), cases.mixed_buffer(), cases.family("lz", 131073, __import__("random").Random(9))]:
        n = len(data)
        d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        gl = np.zeros(n, dtype=np.uint16); go = np.zeros(n, dtype=np.uint16)
        assert lib.mscomp_amd_debug_xpress_matches(gpu_ctx._h, C.c_void_p(d.data_ptr()), n, max_off, clip, gl.ctypes.data, go.ctypes.data) == 0
        ol = np.zeros(n, dtype=np.uint32); oo = np.zeros(n, dtype=np.uint32)
        orc.orc_xpress_match_table(data, n, max_off, ol.ctypes.data, oo.ctypes.data)
        exp_l = np.where(ol >= 3, np.minimum(ol, 48) - 3, 0).astype(np.uint16)
        exp_o = np.where(ol >= 3, oo, 0).astype(np.uint16)
        if clip:
            pos = np.arange(n); rem = np.minimum((pos // 65536 + 1) * 65536, n) - pos
            exp_l[rem < 3] = 0; exp_o[rem < 3] = 0
        assert np.array_equal(gl, exp_l) and np.array_equal(go, exp_o)


@pytest.mark.parametrize("fmt", list(FMTS))
def test_one_shot_abi_and_capacity(oracle, fmt):
    """ms_compress with host pointers (the drop-in path): exact-fit capacity, BUF_ERROR one byte short (reference semantics:
    lznt1_compress.cpp:251,267; xpress_compress.cpp:255-342; xpress_huff_compress.cpp:283,319), empty input."""
    import ms_compress_amd as m
    f = FMTS[fmt]
    data = cases.mixed_buffer()[:150_000]
    exp = oracle.oracle_compress(f, data)[1]
    assert m.compress(f, data) == exp
    assert m.compress(f, data, out_capacity=len(exp)) == exp
    with pytest.raises(m.MSCompError) as e:
        m.compress(f, data, out_capacity=len(exp) - 1)
    assert e.value.status == m.MSCOMP_BUF_ERROR
    assert m.compress(f, b"") == oracle.oracle_compress(f, b"")[1]
    # per-codec entry points are the same code path
    lib = m.load_library()
    out = C.create_string_buffer(len(exp) + 8); n = C.c_size_t(len(exp) + 8)
    assert getattr(lib, fmt + "_compress")(data, len(data), out, C.byref(n)) == 0 and out.raw[: n.value] == exp


def test_lznt1_end_of_buffer_terminal(gpu_ctx):
    """00 00 is written past *out_len when >= 2 bytes of capacity remain, and not counted (lznt1_compress.cpp:270-271)."""
    import ms_compress_amd as m
    lib = m.load_library()
    data = b"hello hello hello hello hello hello"
    out = C.create_string_buffer(b"\xAA" * 128, 128); n = C.c_size_t(128)
    assert lib.ms_compress(2, data, len(data), out, C.byref(n)) == 0
    assert out.raw[n.value: n.value + 2] == b"\x00\x00" and out.raw[n.value + 2] == 0xAA


def test_batch_exact_capacities(oracle, gpu_ctx):
    """batch interface: per-unit BUF_ERROR does not disturb neighbours; unaligned output offsets are accepted."""
    import ms_compress_amd as m
    units = cases.edge_cases(sizes=[0, 1, 100, 4097, 70000], kinds=["words", "lz", "random"])
    for f in (2, 3, 4):
        exp = [oracle.oracle_compress(f, u)[1] for u in units]
        caps = [len(e) - (1 if (i % 3 == 1 and len(e) > 0) else 0) for i, e in enumerate(exp)]
        got, st = m.compress_units(f, units, ctx=gpu_ctx, capacities=caps)
        for i, (g, s, e) in enumerate(zip(got, st, exp)):
            if caps[i] < len(e):
                assert s == m.MSCOMP_BUF_ERROR and g is None
            else:
                assert s == 0 and g == e, (f, i)


@pytest.mark.parametrize("fmt", list(FMTS))
def test_unaligned_unit_offsets(oracle, gpu_ctx, fmt):
    """Units packed back to back with NO alignment (odd input and output offsets): exercises the byte-granular loaders."""
    import torch
    import ms_compress_amd as m
    f = FMTS[fmt]
    units = cases.edge_cases(sizes=[1, 7, 100, 4097, 8191, 20001, 65537, 70001], kinds=["words", "lz", "run"])
    lens = [len(u) for u in units]
    in_off, in_total = m.pack_offsets(lens, align=1)
    in_off = in_off + np.uint64(3)                                # first unit at an odd address too
    caps = [m.max_compressed_size(f, n) + 2 for n in lens]
    out_off, out_total = m.pack_offsets(caps, align=1)
    out_off = out_off + np.uint64(5)
    blob = np.zeros(in_total + 32, dtype=np.uint8)
    for u, o in zip(units, in_off):
        blob[int(o): int(o) + len(u)] = np.frombuffer(u, dtype=np.uint8)
    dev = torch.device("cuda", gpu_ctx.device)
    d_in = torch.from_numpy(blob).to(dev)
    d_out = torch.zeros(out_total + 32, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(len(units), dtype=torch.int64, device=dev)
    d_st = torch.zeros(len(units), dtype=torch.int32, device=dev)
    plan = m.Plan(gpu_ctx, f, in_off, lens, out_off, caps)
    plan.execute(d_in, d_out, d_len, d_st)
    torch.cuda.synchronize()
    plan.close()
    h_out, h_len, h_st = d_out.cpu().numpy(), d_len.cpu().numpy(), d_st.cpu().numpy()
    for i, u in enumerate(units):
        exp = oracle.oracle_compress(f, u)[1]
        got = bytes(h_out[int(out_off[i]): int(out_off[i]) + int(h_len[i])])
        assert h_st[i] == 0 and got == exp, (fmt, i, len(u))


@pytest.mark.parametrize("nkeys", [1, 2, 3, 7, 33, 257, 2048])
def test_lds_atomics_are_served_in_lane_order(gpu_ctx, nkeys):
    """The LZNT1 bucket sort (rank = value returned by the histogram atomic) and the Xpress chain links
    (link = exchange(head[hash], position)) rely on gfx950 serving the same-address LDS atomics of ONE wave
    instruction in lane order; the parity tests would catch a violation only indirectly (a tie decided differently)."""
    bad = gpu_ctx.lib.mscomp_amd_debug_lds_lane_order(gpu_ctx._h, 1234 + nkeys, 512, 64, nkeys)
    assert bad == 0, "%d lanes were served out of lane order" % bad


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_xpress_emit_kernels_agree(oracle, gpu_ctx, mode):
    """The Xpress parse/emit stage has two kernels (one wave per unit / four waves per unit with speculative segments and
    seam repair, 4 or 16 waves); the library picks by batch size. Both must produce the reference's bytes: edge families, a multi-
    super-block stream (> 64 KiB, incl. a run longer than 64 KiB and an incompressible stretch) and exact capacities."""
    import ms_compress_amd as m
    gpu_ctx.lib.mscomp_amd_debug_set_xpress_emit(mode)
    try:
        units = cases.edge_cases()
        rng = np.random.default_rng(77)
        big = np.concatenate([np.frombuffer(b"the quick brown fox jumps over the lazy dog. " * 3000, dtype=np.uint8),
                              np.zeros(150000, dtype=np.uint8), rng.integers(0, 256, 90000, dtype=np.uint8),
                              np.frombuffer(b"ab" * 50000, dtype=np.uint8), rng.integers(0, 4, 120000, dtype=np.uint8)])
        units = list(units) + [big, big[:200001], big[131000:400000]]
        _check_units(m, oracle, FMTS["xpress"], units, gpu_ctx)
        # exact and one-short capacities go through the capacity-checked store path
        sub = [u for u in units if 0 < len(u) <= 70000][:60]
        exp = [oracle.oracle_compress(FMTS["xpress"], u)[1] for u in sub]
        got, st = m.compress_units(FMTS["xpress"], sub, ctx=gpu_ctx, capacities=[len(e) for e in exp])
        assert all(s == 0 for s in st) and all(g == e for g, e in zip(got, exp))
        got, st = m.compress_units(FMTS["xpress"], sub, ctx=gpu_ctx, capacities=[len(e) - 1 for e in exp])
        assert all(s == -5 for s in st)
    finally:
        gpu_ctx.lib.mscomp_amd_debug_set_xpress_emit(0)


@pytest.mark.parametrize("mode", [1, 2])
def test_lznt1_chunk_kernels_agree(oracle, gpu_ctx, mode):
    """The LZNT1 chunk stage has two kernels (one wave per chunk; four waves per chunk with speculative segments, seam
    repair and a cascade check). Both must produce the reference's bytes."""
    import ms_compress_amd as m
    gpu_ctx.lib.mscomp_amd_debug_set_lznt1(mode)
    try:
        units = list(cases.edge_cases())
        rng = np.random.default_rng(5)
        # chunks whose parse never re-synchronises quickly: period-3 / period-5 runs with phase shifts, long runs, mixtures
        for period in (3, 5, 7, 64, 65):
            base = rng.integers(0, 256, period, dtype=np.uint8)
            units.append(np.tile(base, 20000 // period + 1)[:20000])
        units.append(np.concatenate([np.zeros(5000, np.uint8), rng.integers(0, 2, 6000, dtype=np.uint8), np.frombuffer(b"abcabcabd" * 900, dtype=np.uint8)]))
        _check_units(m, oracle, FMTS["lznt1"], units, gpu_ctx)
    finally:
        gpu_ctx.lib.mscomp_amd_debug_set_lznt1(0)


@pytest.mark.parametrize("fmt", list(FMTS))
def test_plan_reuse_replays_a_graph(oracle, gpu_ctx, fmt):
    """A plan executed repeatedly replays its launches as a hipGraph from the second execution on; the graph must be
    re-captured when the buffers change (different input tensor, different output tensor) and when another plan grows the
    context's scratch. Every execution must give the reference's bytes."""
    import torch
    import ms_compress_amd as m
    rng = np.random.default_rng(11)
    def batch(seed):
        r = np.random.default_rng(seed)
        return [np.frombuffer((b"lorem ipsum dolor sit amet %d " % seed) * 900, dtype=np.uint8), r.integers(0, 3, 40000, dtype=np.uint8),
                np.zeros(70000, dtype=np.uint8), r.integers(0, 256, 9000, dtype=np.uint8)]
    dev = torch.device("cuda", 0)
    def run(plan, units, d_in, d_out, caps, out_off):
        d_len = torch.zeros(len(units), dtype=torch.int64, device=dev); d_st = torch.full((len(units),), -99, dtype=torch.int32, device=dev)
        plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
        out = d_out.cpu().numpy(); ln = d_len.cpu().numpy()
        for i, u in enumerate(units):
            es, exp = oracle.oracle_compress(FMTS[fmt], u)
            assert int(d_st[i]) == 0 and bytes(out[int(out_off[i]): int(out_off[i]) + int(ln[i])]) == exp
    units = batch(1)
    lens = [len(u) for u in units]
    in_off, in_total = m.pack_offsets(lens)
    caps = [m.max_compressed_size(FMTS[fmt], n) + 2 for n in lens]
    out_off, out_total = m.pack_offsets(caps)
    plan = m.Plan(gpu_ctx, FMTS[fmt], in_off, lens, out_off, caps)
    def upload(us):
        buf = np.zeros(in_total + 16, dtype=np.uint8)
        for o, u in zip(in_off, us): buf[int(o): int(o) + len(u)] = u
        return torch.from_numpy(buf).to(dev)
    d_in1 = upload(units); d_out1 = torch.zeros(out_total + 16, dtype=torch.uint8, device=dev)
    for _ in range(3): run(plan, units, d_in1, d_out1, caps, out_off)          # eager, capture, replay
    units2 = [rng.permutation(u) if i == 1 else u for i, u in enumerate(units)]  # same lengths, other bytes, other tensors
    d_in2 = upload(units2); d_out2 = torch.zeros(out_total + 16, dtype=torch.uint8, device=dev)
    for _ in range(2): run(plan, units2, d_in2, d_out2, caps, out_off)         # re-capture, replay
    big = [np.tile(units[0], 40)]                                               # a larger plan grows (moves) the shared scratch
    m.compress_units(FMTS[fmt], big, ctx=gpu_ctx)
    for _ in range(2): run(plan, units, d_in1, d_out1, caps, out_off)          # re-capture after the scratch moved, replay
    plan.close()


def test_compact_batch_packs_outputs_back_to_back(oracle, gpu_ctx):
    """mscomp_amd_compact_batch: the outputs of a batch, which sit at out_off[i] with gaps, packed on the device"""
    import ctypes as C
    import torch
    import ms_compress_amd as m
    units = [cases.mixed_buffer()[:300000], b"", b"abc" * 1000, cases.mixed_buffer()[100000:170001], bytes(5)]
    lens = np.array([len(u) for u in units], dtype=np.uint64)
    lib = gpu_ctx.lib
    dev = torch.device("cuda", 0)
    for fmt in (2, 3, 4):
        out_off = np.zeros(len(units), dtype=np.uint64); out_cap = np.zeros(len(units), dtype=np.uint64)
        total = lib.mscomp_amd_plan_layout(fmt, len(units), lens.ctypes.data, 16, out_off.ctypes.data, out_cap.ctypes.data)
        in_off, in_total = m.pack_offsets([int(x) for x in lens])
        blob = np.zeros(in_total + 16, dtype=np.uint8)
        for o, u in zip(in_off, units): blob[int(o): int(o) + len(u)] = np.frombuffer(u, dtype=np.uint8)
        d_in = torch.from_numpy(blob).to(dev); d_out = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
        d_len = torch.zeros(len(units), dtype=torch.int64, device=dev); d_st = torch.zeros(len(units), dtype=torch.int32, device=dev)
        p = m.Plan(gpu_ctx, fmt, in_off, lens, out_off, out_cap); p.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize(); p.close()
        d_packed = torch.zeros(total + 16, dtype=torch.uint8, device=dev); d_poff = torch.zeros(len(units) + 1, dtype=torch.int64, device=dev)
        st = lib.mscomp_amd_compact_batch(gpu_ctx._h, len(units), C.c_void_p(d_out.data_ptr()), out_off.ctypes.data, out_cap.ctypes.data,
                                          C.c_void_p(d_len.data_ptr()), C.c_void_p(d_packed.data_ptr()), C.c_void_p(d_poff.data_ptr()))
        assert st == 0
        torch.cuda.synchronize()
        want = b"".join(oracle.oracle_compress(fmt, u)[1] for u in units)
        poff = d_poff.cpu().numpy()
        assert int(poff[-1]) == len(want) and bytes(d_packed[: len(want)].cpu().numpy()) == want
        acc = 0
        for i, u in enumerate(units):
            assert int(poff[i]) == acc; acc += len(oracle.oracle_compress(fmt, u)[1])


def test_one_shot_calls_from_several_host_threads(oracle, gpu_ctx):
    """The reference's one-shot calls are reentrant; here every calling host thread gets its own GPU context. Four threads compress
    and decompress different buffers of all three formats at the same time through the host-pointer entry points."""
    import threading
    import ms_compress_amd as m
    bufs = [cases.mixed_buffer()[k * 50000: k * 50000 + 120000 + 1000 * k] for k in range(4)]
    errors = []

    def work(k):
        try:
            for rep in range(3):
                for fmt in (2, 3, 4):
                    c = m.compress(fmt, bufs[k])
                    assert c == oracle.oracle_compress(fmt, bufs[k])[1], ("compress", k, fmt)
                    assert m.decompress(fmt, c, len(bufs[k])) == bufs[k], ("decompress", k, fmt)
        except Exception as e:                     # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))
    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_huffman_builder_stage(oracle, gpu_ctx):
    """Stage-level parity of the Huffman builder (xh_huff_kernel's heap, HuffmanEncoder.h:58-107): code lengths for histograms given
    directly -- against the oracle for every case and against the digests the reference's own header left in the fixture. A tie-break
    or rescale regression shows up here as "histogram i, symbol s", not as a differing chunk digest."""
    import ctypes as C
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "huff_lengths.json")))
    hs = cases.huff_histograms()
    rnd = __import__("random").Random(77)
    more = [[rnd.randint(0, 2) * int(2 ** rnd.uniform(0, 14)) for _ in range(512)] for _ in range(400)]      # oracle-only cases
    allc = np.asarray(hs + more, dtype=np.uint32)
    lens = np.zeros((len(allc), 512), dtype=np.uint8)
    st = gpu_ctx.lib.mscomp_amd_debug_huff_lengths(gpu_ctx._h, allc.ctypes.data, len(allc), lens.ctypes.data)
    assert st == 0
    lib = oracle.load_oracle()
    deep = 0
    for i in range(len(allc)):
        want = np.zeros(512, dtype=np.uint8)
        lib.orc_huff_lengths(allc[i].ctypes.data, want.ctypes.data)
        if not np.array_equal(lens[i], want):
            s = int(np.nonzero(lens[i] != want)[0][0])
            raise AssertionError("histogram %d: symbol %d got length %d, CreateCodes gives %d" % (i, s, lens[i][s], want[s]))
        deep += int(want.max() == 15)
    assert deep > 20                                                  # the > 15-bit rescale loop was exercised
    for i in range(len(hs)):
        assert hashlib.sha256(lens[i].tobytes()).hexdigest()[:16] == gold["sha256_16_per_case"][i], i


@pytest.mark.parametrize("fmt", [3, 4])
def test_both_find_kernels_give_the_same_bytes(oracle, gpu_ctx, fmt):
    """Find for every position (xp_find_kernel, mscomp_amd_debug_set_finder(2)) against the oracle on the edge families, the mixed
    buffer (100 000 zeros: matches longer than 8 KiB and the lagging-Fill resume points of Xpress) and corpus slices; the default
    (the lazy finder of csrc/xpress_lazy.hip for Xpress units up to 64 KiB) runs in every other test."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    lib = gpu_ctx.lib
    units = cases.edge_cases()[::3] + [cases.mixed_buffer()[i:i + 65536] for i in range(0, 300000, 65536)]
    units += [corpus.file_bytes(i, 300_000).tobytes()[o:o + 65536] for i in (1, 3, 9) for o in (0, 65536, 200000)]
    if fmt == 4:
        units += [cases.mixed_buffer(), corpus.file_bytes(1, 400_000).tobytes()]          # multi-chunk units: windows reach into the previous chunk
    lib.mscomp_amd_debug_set_finder(2)
    try:
        got, st = m.compress_units(fmt, units, ctx=gpu_ctx)
    finally:
        lib.mscomp_amd_debug_set_finder(1)
    for i, (u, g, s) in enumerate(zip(units, got, st)):
        es, exp = oracle.oracle_compress(fmt, u)
        assert s == 0 and g == exp, "unit %d (%d bytes)" % (i, len(u))


def test_lazy_xpress_finder_tiles_runs_and_parked_walks(oracle, gpu_ctx):
    """The lazy Xpress finder (csrc/xpress_lazy.hip, units up to 64 KiB) at the places its state machine has special cases: unit sizes
    around its 16 KiB tiles (walks parked at a tile end and taken up in the next), runs longer than a lane extends itself (112 bytes:
    the wave takes over; the long-match cache), matches longer than 8 KiB (the eight resume points of the lagging Fill) starting at
    awkward offsets, candidates that match exactly 16 / 32 / 48 bytes (the compare's second and third block), and the all-positions
    kernel (debug finder 2) on the same units."""
    import random
    import ms_compress_amd as m
    rnd = random.Random(77)
    text = b"".join(rnd.choice(cases.WORDS) for _ in range(30000))
    noise = bytes(rnd.getrandbits(8) for _ in range(70000))
    units = []
    for n in (16383, 16384, 16385, 16400, 32767, 32768, 32769, 49151, 49152, 49153, 65535, 65536):
        units += [text[:n], cases.family("lz", n, rnd), cases.family("two", n, rnd)]
    for start in (0, 1, 5, 16380, 16384, 30000):                          # a long run that begins before / at / behind a tile boundary
        for run in (100, 113, 128, 129, 300, 8190, 8192, 8193, 8200, 9000, 20000, 40000):
            u = bytearray(noise[:65536])
            end = min(65536, start + 40 + run)
            u[start + 40:end] = bytes([0x41]) * (end - start - 40)
            units.append(bytes(u[:min(65536, end + 3000)]))
    for k in (15, 16, 17, 31, 32, 33, 47, 48, 49, 50, 111, 112, 113, 127):   # a repeat of exactly k bytes, then a different byte
        pat = noise[1000:1000 + k]
        u = bytearray(noise[:3000]) + pat + b"\x01" + bytearray(noise[5000:6000]) + pat + b"\x02" + bytearray(noise[7000:9000]) + pat + b"\x03"
        units.append(bytes(u))
    periodic = (noise[:37] * 2000)[:65536]
    units += [periodic, periodic[:16384 + 5], bytes(65536), bytes(16384), bytes(16385)]
    lib = gpu_ctx.lib
    for finder in (1, 2):
        lib.mscomp_amd_debug_set_finder(finder)
        try:
            got, st = m.compress_units(3, units, ctx=gpu_ctx)
        finally:
            lib.mscomp_amd_debug_set_finder(1)
        for i, (u, g, s) in enumerate(zip(units, got, st)):
            es, exp = oracle.oracle_compress(3, u)
            assert s == 0 and g == exp, "finder %d unit %d (%d bytes)" % (finder, i, len(u))


def test_lazy_finder_long_match_cache_under_concurrent_insertions(oracle, gpu_ctx):
    """Round 6, found by REAL files (tests/test_gpu_realdata.py: 64 KiB pieces of the GPU code tables in the image): the lazy Xpress finder keeps
    the ends of long matches in a 4-entry cache in LDS that all waves of a block read and write. A wave's read of an entry is served in two halves
    of 32 lanes; an insertion by another wave between the halves left the wave with two different views, its hit / miss decision stopped being
    uniform and half a wave ran the cooperative compare -- a match end too far out, the next token start never claimed, literals where the
    reference has a match, in 1-10 % of such units and never twice the same. The shape that does it: one period, a changed byte every 500-3000
    positions (matches of ONE distance, hundreds to thousands of bytes long, extended by all waves at once). Many copies of such units, several
    passes: every copy must give the oracle's bytes."""
    import ms_compress_amd as m
    units = [cases.periodic_with_mutations(seed=5, gap=(500, 3000)), cases.periodic_with_mutations(seed=6, period=3700, gap=(400, 2500)),
             cases.periodic_with_mutations(seed=7, period=1878, gap=(500, 3000)), cases.periodic_with_mutations(n=50000, seed=8, period=7426, gap=(300, 2000))]
    want = [oracle.oracle_compress(3, u)[1] for u in units]
    for _ in range(3):
        got, st = m.compress_units(3, [u for u in units for _ in range(160)], ctx=gpu_ctx)
        assert all(s == 0 for s in st)
        bad = [i for i, g in enumerate(got) if g != want[i // 160]]
        assert not bad, "%d of %d copies differ from the oracle, first: copy %d of unit %d" % (len(bad), len(got), bad[0] % 160, bad[0] // 160)


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_repetitive_families_many_copies(oracle, gpu_ctx, fmt):
    """Highly repetitive inputs -- the shapes on which waves and blocks do the most handing-over (long matches extended cooperatively, speculative
    segments that re-synchronise late, parses that land on the same few positions) -- as many concurrent copies per batch, two passes: every copy
    must give the oracle's bytes (a race between waves shows up in SOME copies of SOME passes; one unit alone can run clean for ever)."""
    import random
    import ms_compress_amd as m
    rnd = random.Random(31)
    noise = bytes(rnd.getrandbits(8) for _ in range(70000))
    units = [cases.periodic_with_mutations(n=65536, period=p, seed=40 + i, gap=g) for i, (p, g) in enumerate(((1, (300, 3000)), (2, (200, 2000)), (3, (500, 2500)), (7, (100, 900)),
                                                                                                           (64, (300, 3000)), (257, (150, 1500)), (4096, (400, 4000)), (8191, (500, 5000))))]
    units += [cases.few_distances(seed=50, run=(600, 3000)), cases.few_distances(dists=(1, 2, 3, 4095, 4096, 4097), seed=51, run=(40, 900)),
              cases.few_distances(dists=(8192, 8191, 8190, 16), seed=52, run=(100, 5000)), (noise[:4000] + bytes(9000)) * 5, (b"ab" * 3000 + noise[:100]) * 10]
    if fmt != 3:
        units += [cases.periodic_with_mutations(n=200000, period=65536, seed=60, gap=(1000, 9000)), cases.periodic_with_mutations(n=150000, period=3756, seed=61, gap=(500, 3000))]
    want = [oracle.oracle_compress(fmt, u)[1] for u in units]
    copies = 48
    for _ in range(2):
        got, st = m.compress_units(fmt, [u for u in units for _ in range(copies)], ctx=gpu_ctx)
        assert all(s == 0 for s in st)
        bad = [i for i, g in enumerate(got) if g != want[i // copies]]
        assert not bad, "%d of %d copies differ from the oracle, first: copy %d of unit %d (%d bytes)" % (len(bad), len(got), bad[0] % copies, bad[0] // copies, len(units[bad[0] // copies]))


def _lznt1_tokens(chunk_image):
    """(position, length, offset) of every token of ONE compressed LZNT1 chunk image (lznt1_decompress.cpp:37-121); offset 0 = literal"""
    hdr = chunk_image[0] | (chunk_image[1] << 8)
    body = chunk_image[2:2 + (hdr & 0xFFF) + 1]
    if not hdr & 0x8000:
        return None                                          # stored raw
    toks, ip, op = [], 0, 0
    while ip < len(body):
        flags = body[ip]; ip += 1
        for i in range(8):
            if ip >= len(body):
                break
            if not (flags >> i) & 1:
                toks.append((op, 1, 0)); ip += 1; op += 1
                continue
            shift = 12 if op <= 16 else 12 - ((op - 1).bit_length() - 4)
            t = body[ip] | (body[ip + 1] << 8); ip += 2
            toks.append((op, (t & ((1 << shift) - 1)) + 3, (t >> shift) + 1))
            op += toks[-1][1]
    return toks


def test_lznt1_find_stage(oracle, gpu_ctx):
    """Stage-level parity of LZNT1Dictionary::Find (LZNT1Dictionary.h:114-143: longest match, OLDEST candidate on ties) as the HIP chunk
    kernel evaluates it -- lazily, at token starts: every token the GPU emitted for a 4 KiB chunk against the oracle's per-position
    match table (orc_lznt1_match_table). A tie-break regression is reported as chunk / position / (length, offset), not as a digest."""
    import ctypes as C
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    rnd = __import__("random").Random(41)
    chunks = []
    for i in range(12):
        data = corpus.file_bytes(i, 600_000).tobytes()
        chunks += [data[o:o + 4096] for o in (0, 4096 * 37, 4096 * 101)]
    chunks += [cases.family(k, n, rnd) for k in ("two", "words", "lz", "run") for n in (4096, 3000, 517)]
    chunks += [bytes((j * 7 + (j // 5)) & 0x3 for j in range(4096)), (b"abcde" * 1000)[:4096], (b"abc" * 2000)[:4096]]   # many equal-length candidates
    got, st = m.compress_units(2, chunks, ctx=gpu_ctx)
    lib = oracle.load_oracle()
    checked = matches = 0
    for ci, (data, comp) in enumerate(zip(chunks, got)):
        toks = _lznt1_tokens(comp)
        if toks is None:
            continue
        n = len(data)
        ln = np.zeros(n, dtype=np.uint16); off = np.zeros(n, dtype=np.uint16)
        lib.orc_lznt1_match_table(data, n, ln.ctypes.data, off.ctypes.data)
        for pos, l, o in toks:
            want = (int(ln[pos]), int(off[pos])) if ln[pos] >= 3 else (1, 0)
            assert (l, o) == want, "chunk %d position %d: GPU token (len %d, off %d), Find gives (len %d, off %d)" % (ci, pos, l, o, want[0], want[1])
            checked += 1; matches += o != 0
        assert sum(t[1] for t in toks) == n
    assert checked > 20000 and matches > 5000


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_one_shot_with_a_terabyte_of_capacity(oracle, gpu_ctx, fmt):
    """*out_len = 1 << 40 is legal in the reference (the capacity of a buffer the caller says it has): ms_compress / ms_decompress must
    answer what they answer with an exact capacity, not MSCOMP_MEM_ERROR from an attempt to mirror the capacity in HBM."""
    import ctypes as C
    import ms_compress_amd as m
    lib = m.load_library()
    for data in (cases.mixed_buffer()[:150000], b"", b"a", bytes(300000), cases.family("lz", 70000, __import__("random").Random(3))):
        want = oracle.oracle_compress(fmt, data)[1]
        out = C.create_string_buffer(len(want) + 64)
        n = C.c_size_t(1 << 40)
        assert lib.ms_compress(fmt, data, len(data), out, C.byref(n)) == 0
        assert n.value == len(want) and out.raw[: n.value] == want
        back = C.create_string_buffer(len(data) + 64)
        n = C.c_size_t(1 << 40)
        est, exp = oracle.oracle_decompress(fmt, want, len(data) + 64)
        st = lib.ms_decompress(fmt, want, len(want), back, C.byref(n))
        # (the reference refuses its own compression of the empty buffer, FF FF FF FF, as Xpress input: xpress_decompress.cpp:414-417)
        assert st == est and (st != 0 or (n.value == len(data) and back.raw[: n.value] == data)), (fmt, len(data), st, est)
        assert st == 0 or (fmt == 3 and len(data) == 0)


def test_order_independent_form_of_the_sort_and_the_links(oracle, gpu_ctx):
    """what a device gets that does NOT serve same-address LDS atomics of one instruction in lane order: the LZNT1 bucket
    sort (both chunk kernels) and the Xpress chain links issue their returning atomic one lane at a time. Forced through the test hook: the same
    bytes as the checker's on the edge families and on a corpus slice with long units (several link chunks, previous-chunk chains)."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    lib = m.load_library()
    units = cases.edge_cases()[::3] + [cases.mixed_buffer(), corpus.by_name("samba", 400_000).tobytes(), corpus.by_name("nci", 300_000).tobytes()]
    lib.mscomp_amd_debug_set_serial_atomics(1)
    try:
        for fmt in (2, 3, 4):
            for lz in ((1, 2) if fmt == 2 else (0,)):
                lib.mscomp_amd_debug_set_lznt1(lz)
                got, st = m.compress_units(fmt, units)
                for i, (u, g, s_) in enumerate(zip(units, got, st)):
                    es, exp = oracle.oracle_compress(fmt, u)
                    assert s_ == 0 and es == 0 and g == exp, (fmt, lz, i, len(u))
    finally:
        lib.mscomp_amd_debug_set_lznt1(0)
        lib.mscomp_amd_debug_set_serial_atomics(0)

"""GPU: the LZNT1 streaming compressor (ms_deflate_init / ms_deflate / ms_deflate_end) against the compiled
reference's streaming compressor, call by call: same status, same bytes consumed and produced by every call, same output, for
random ways of slicing the input and the output windows and random MSCOMP_FLUSH points."""
import ctypes as C
import random

import pytest

import cases

pytestmark = pytest.mark.gpu
NO_FLUSH, FLUSH, FINISH = 0, 2, 4


class Stream(C.Structure):                      # include/mscomp/general.h:95-121 (default build: error + warning texts)
    _fields_ = [("format", C.c_int), ("compressing", C.c_bool), ("in_", C.c_void_p), ("in_avail", C.c_size_t), ("in_total", C.c_size_t),
                ("out", C.c_void_p), ("out_avail", C.c_size_t), ("out_total", C.c_size_t), ("error", C.c_char * 256),
                ("warning", C.c_char * 256), ("state", C.c_void_p)]


def drive(lib, fmt, data, steps, tail_window):
    """steps: (bytes offered, output window, flush). Afterwards MSCOMP_FINISH with `tail_window` until MSCOMP_STREAM_END."""
    for f in (lib.ms_deflate_init, lib.ms_deflate, lib.ms_deflate_end):
        f.restype = C.c_int
    lib.ms_deflate_init.argtypes = [C.c_int, C.POINTER(Stream)]
    lib.ms_deflate.argtypes = [C.POINTER(Stream), C.c_int]
    lib.ms_deflate_end.argtypes = [C.POINTER(Stream)]
    s = Stream()
    assert lib.ms_deflate_init(fmt, C.byref(s)) == 0
    inbuf = C.create_string_buffer(bytes(data), max(1, len(data)))
    outbuf = C.create_string_buffer(len(data) + len(data) // 1000 + 70000)
    ipos = opos = 0
    trace = []

    def call(offer, window, flush):
        nonlocal ipos, opos
        offer = min(offer, len(data) - ipos)
        s.in_ = C.addressof(inbuf) + ipos; s.in_avail = offer
        s.out = C.addressof(outbuf) + opos; s.out_avail = window
        st = lib.ms_deflate(C.byref(s), flush)
        took, gave = offer - s.in_avail, window - s.out_avail
        ipos += took; opos += gave
        trace.append((st, took, gave, s.in_total, s.out_total))
        return st

    for offer, window, flush in steps:
        if call(offer, window, flush) < 0:
            break
    for _ in range(100000):
        if call(len(data) - ipos, tail_window, FINISH) != 0:
            break
    end = lib.ms_deflate_end(C.byref(s))
    return outbuf.raw[:opos], trace, end


def plans(rnd, n):
    yield [], 1 << 30                                              # everything in one MSCOMP_FINISH call
    yield [(n, 1 << 30, NO_FLUSH)], 1 << 30
    yield [(4096, 5000, NO_FLUSH)] * (n // 4096 + 2), 4098
    yield [(1000, 700, NO_FLUSH)] * (n // 1000 + 2), 100            # output windows smaller than a chunk
    for _ in range(6):
        steps = []
        for _ in range(rnd.randint(1, 60)):
            steps.append((rnd.choice((1, 17, 4095, 4096, 4097, 10000, 50000, rnd.randint(1, 70000))),
                          rnd.choice((0, 1, 2, 100, 4097, 4098, 4099, 9000, 1 << 20)), rnd.choice((NO_FLUSH, NO_FLUSH, NO_FLUSH, FLUSH))))
        yield steps, rnd.choice((1, 3, 4098, 1 << 20))


def test_lznt1_deflate_matches_reference_call_by_call(oracle, gpu_ctx):
    import ms_compress_amd as m
    ref = oracle.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    mine = m.load_library()
    rnd = random.Random(9)
    datas = [b"", b"x", cases.family("words", 4096, rnd), cases.family("lz", 12289, rnd), cases.family("random", 9000, rnd),
             cases.mixed_buffer()[95000:95000 + 180000]]
    n_cmp = 0
    for data in datas:
        for steps, tail in plans(rnd, len(data)):
            want = drive(ref, 2, data, steps, tail)
            got = drive(mine, 2, data, steps, tail)
            assert got[1] == want[1], (len(data), steps[:4], tail)
            assert got[0] == want[0] and got[2] == want[2]
            n_cmp += 1
    assert n_cmp >= 50
    # without MSCOMP_FLUSH the stream is the one-shot output
    data = datas[-1]
    out, _, end = drive(mine, 2, data, [(7000, 3000, NO_FLUSH)] * 40, 5000)
    assert end == 0 and out == oracle.oracle_compress(2, data)[1]


def test_copy_codec_and_argument_errors(gpu_ctx):
    import ms_compress_amd as m
    mine = m.load_library()
    out, trace, end = drive(mine, 0, b"hello world" * 10, [(50, 20, NO_FLUSH)] * 3, 1000)
    assert out == b"hello world" * 10 and trace[-1][0] == 1 and end == 0
    s = Stream()
    mine.ms_deflate_init.argtypes = [C.c_int, C.POINTER(Stream)]
    assert mine.ms_deflate_init(3, C.byref(s)) == -4 and mine.ms_deflate_init(4, C.byref(s)) == -2 and mine.ms_deflate_init(9, C.byref(s)) == -2   # like the reference
    assert mine.ms_deflate_init(2, C.byref(s)) == 0
    assert mine.ms_deflate_end(C.byref(s)) == -3                     # ended before MSCOMP_FINISH was answered with MSCOMP_STREAM_END
    assert mine.ms_deflate(C.byref(s), NO_FLUSH) == -2               # no state any more


def drive_inflate(lib, fmt, stream, steps, tail_window, cap):
    """steps: (bytes offered, output window). Afterwards everything left is offered with `tail_window` until the call makes no
    progress, fails, or reports an end."""
    for f in (lib.ms_inflate_init, lib.ms_inflate, lib.ms_inflate_end):
        f.restype = C.c_int
    lib.ms_inflate_init.argtypes = [C.c_int, C.POINTER(Stream)]
    lib.ms_inflate.argtypes = [C.POINTER(Stream)]
    lib.ms_inflate_end.argtypes = [C.POINTER(Stream)]
    s = Stream()
    assert lib.ms_inflate_init(fmt, C.byref(s)) == 0
    inbuf = C.create_string_buffer(bytes(stream), max(1, len(stream)))
    outbuf = C.create_string_buffer(cap + 8192)
    ipos = opos = 0
    trace = []

    def call(offer, window):
        nonlocal ipos, opos
        offer = min(offer, len(stream) - ipos); window = min(window, cap - opos)
        s.in_ = C.addressof(inbuf) + ipos; s.in_avail = offer
        s.out = C.addressof(outbuf) + opos; s.out_avail = window
        st = lib.ms_inflate(C.byref(s))
        took, gave = offer - s.in_avail, window - s.out_avail
        ipos += took; opos += gave
        trace.append((st, took, gave, s.in_total, s.out_total))
        return st, took + gave

    alive = True
    for offer, window in steps:
        st, _ = call(offer, window)
        if st < 0 or st == 1:
            alive = False
            break
    for _ in range(100000):
        if not alive:
            break
        st, progress = call(len(stream) - ipos, tail_window)
        if st != 0 or progress == 0:
            break
    end = lib.ms_inflate_end(C.byref(s))
    return outbuf.raw[:opos], trace, end


def test_lznt1_inflate_matches_reference_call_by_call(oracle, gpu_ctx):
    import ms_compress_amd as m
    ref = oracle.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    mine = m.load_library()
    rnd = random.Random(11)
    datas = [b"", b"x", cases.family("words", 4096, rnd), cases.family("lz", 12289, rnd), cases.family("random", 9000, rnd),
             cases.mixed_buffer()[95000:95000 + 120000]]
    streams = []
    for d in datas:
        c = oracle.oracle_compress(2, d)[1]
        streams += [(c, len(d)), (c + b"\0\0", len(d)), (c + b"\0", len(d)), (c + c, 2 * len(d)), (c[: len(c) // 2], len(d)), (c, max(0, len(d) - 1))]
    big = oracle.oracle_compress(2, datas[-1])[1]
    streams += [(big[:1] + bytes([big[1] ^ 0x40]) + big[2:], len(datas[-1])), (big[:5000] + b"\0\0" + big[5002:], len(datas[-1])),
                (big[:3000] + bytes(rnd.getrandbits(8) for _ in range(50)) + big[3050:], len(datas[-1]))]
    n_cmp = 0
    for stream, cap in streams:
        plans = [([], 1 << 30), ([(4096, 4096)] * 8, 1 << 30), ([(1000, 700)] * 30, 100), ([(3, 5000)] * 40, 4096)]
        for _ in range(3):
            plans.append(([(rnd.choice((1, 2, 3, 17, 4097, 4098, 4099, 10000, 60000)), rnd.choice((0, 1, 100, 4095, 4096, 4097, 9000, 1 << 20)))
                           for _ in range(rnd.randint(1, 40))], rnd.choice((1, 4095, 4096, 1 << 20))))
        for steps, tail in plans:
            want = drive_inflate(ref, 2, stream, steps, tail, cap)
            got = drive_inflate(mine, 2, stream, steps, tail, cap)
            assert got[1] == want[1], (len(stream), cap, steps[:4], tail, got[1][-3:], want[1][-3:])
            assert got[0] == want[0] and got[2] == want[2]
            n_cmp += 1
    assert n_cmp >= 200


def test_deflate_look_ahead_notices_changed_input(oracle, gpu_ctx):
    """The streaming compressor compresses the chunks that FOLLOW the one it has to hand over piecewise in the same GPU call and keeps
    them; the caller owns its buffer between calls and may put other bytes there. Same calls on the reference and on us, with the input
    behind the consumed part rewritten between two calls: same trace, same output."""
    import ms_compress_amd as m
    ref = oracle.load_ref()
    if ref is None:
        pytest.skip("compiled reference not available")
    rnd = random.Random(9)
    a = cases.family("lz", 40000, rnd); b = cases.family("words", 40000, rnd)

    def run(lib):
        for f in (lib.ms_deflate_init, lib.ms_deflate, lib.ms_deflate_end):
            f.restype = C.c_int
        lib.ms_deflate_init.argtypes = [C.c_int, C.POINTER(Stream)]; lib.ms_deflate.argtypes = [C.POINTER(Stream), C.c_int]; lib.ms_deflate_end.argtypes = [C.POINTER(Stream)]
        s = Stream(); assert lib.ms_deflate_init(2, C.byref(s)) == 0
        inbuf = C.create_string_buffer(a, len(a)); outbuf = C.create_string_buffer(200000)
        ipos = opos = 0; trace = []
        for i in range(4000):
            if i == 3:                                               # after three calls: everything not yet consumed becomes other data
                C.memmove(C.addressof(inbuf) + ipos, b[ipos:], len(a) - ipos)
            s.in_ = C.addressof(inbuf) + ipos; s.in_avail = len(a) - ipos; s.out = C.addressof(outbuf) + opos; s.out_avail = 300
            st = lib.ms_deflate(C.byref(s), FINISH)
            ipos += (len(a) - ipos) - s.in_avail; opos += 300 - s.out_avail
            trace.append((st, ipos, opos))
            if st != 0:
                break
        return outbuf.raw[:opos], trace, lib.ms_deflate_end(C.byref(s))
    assert run(m.load_library()) == run(ref)


def test_lznt1_deflate_under_the_suffix_array_flavour(oracle, gpu_ctx):
    """SURVEY.md 8f-4 through the streaming compressor: with mscomp_amd_set_lznt1_sa_dict(1) the calls match the reference BUILT WITH
    -DMSCOMP_WITH_LZNT1_SA_DICT call by call (status, bytes taken / given, output), and without MSCOMP_FLUSH the stream is that build's
    one-shot output -- not the default flavour's."""
    import ms_compress_amd as m
    ref = oracle.load_ref_sa()
    if ref is None:
        pytest.skip("oracle/_ref/libMSCompression_sa.so not built")
    mine = m.load_library()
    rnd = random.Random(21)
    datas = [cases.family("words", 4096, rnd), cases.family("lz", 12289, rnd), bytes(rnd.choice(b"abc") for _ in range(30000)),
             cases.mixed_buffer()[95000:95000 + 120000]]
    mine.mscomp_amd_set_lznt1_sa_dict(1)
    try:
        n_cmp = 0
        for data in datas:
            for steps, tail in plans(rnd, len(data)):
                want = drive(ref, 2, data, steps, tail)
                got = drive(mine, 2, data, steps, tail)
                assert got[1] == want[1], (len(data), steps[:4], tail)
                assert got[0] == want[0] and got[2] == want[2]
                n_cmp += 1
        assert n_cmp >= 40
        data = datas[2]
        out, _, end = drive(mine, 2, data, [(7000, 3000, NO_FLUSH)] * 10, 5000)
        assert end == 0 and out == oracle.oracle_compress_sa(data)[1] and out != oracle.oracle_compress(2, data)[1]
    finally:
        mine.mscomp_amd_set_lznt1_sa_dict(0)

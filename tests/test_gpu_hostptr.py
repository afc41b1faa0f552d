"""-m gpu: the drop-in host-pointer entry points on LARGE buffers (mscomp.h:59: ms_compress with plain host pointers) and fixed-seed
slices of the three end-of-round soaks (tools/dev/fuzz_*.py), so that every product path has a committed test:

* ms_compress(MSCOMP_LZNT1) of mozilla (51 220 480 B) through BOTH large-buffer paths of csrc/api.hip -- the caller's buffers mapped and
  one launch (lznt1_zero_copy, the default) and the slices on three streams (lznt1_compress_pipelined, mscomp_amd_debug_set_one_shot(1)
  == MSCOMP_AMD_ONE_ZEROCOPY=0): bytes against the REAL reference's digest (tests/golden/corpus_full.json), the uncounted 00 00
  End_of_buffer behind the stream (lznt1_compress.cpp:270), exact / one short / half capacities -> MSCOMP_BUF_ERROR with nothing written
  behind the capacity (lznt1_compress.cpp:251,267);
* four host threads, each compressing its own 21-51 MB file at the same time (one context per thread: SURVEY 8b "Threading");
* ms_compress(MSCOMP_XPRESS / MSCOMP_XPRESS_HUFF) of a whole file as ONE buffer against the reference's digests.
"""
import ctypes as C
import hashlib
import json
import os
import random
import threading

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "corpus_full.json")))
GUARD = 0xA5


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def _call(lib, fmt, data, cap, room):
    """ms_compress(fmt, data) with *out_len = cap into a buffer of cap + room guard bytes -> (status, out_len, buffer)"""
    out = np.full(cap + room, GUARD, dtype=np.uint8)
    ol = C.c_size_t(cap)
    st = lib.ms_compress(fmt, data.ctypes.data, len(data), out.ctypes.data, C.byref(ol))
    return st, ol.value, out


@pytest.fixture(scope="module")
def mozilla():
    from ms_compress_amd import corpus
    data = np.ascontiguousarray(corpus.by_name("mozilla"))
    assert sha(data.tobytes()) == GOLD["mozilla"]["input_sha256"], "the corpus generator changed: regenerate the fixtures"
    return data


@pytest.mark.parametrize("sliced", [0, 1])
def test_large_lznt1_host_pointer_call_both_paths(gpu_ctx, mozilla, sliced):
    import ms_compress_amd as m
    lib = m.load_library()
    g = GOLD["mozilla"]["lznt1"]
    n = len(mozilla)
    lib.mscomp_amd_debug_set_one_shot(sliced)
    try:
        # generous capacity: the stream, then the uncounted 00 00, then nothing
        cap = lib.ms_max_compressed_size(2, n) + 2
        st, ol, out = _call(lib, 2, mozilla, cap, 64)
        assert st == 0 and ol == g["len"] and sha(out[:ol]) == g["sha256"]
        assert out[ol] == 0 and out[ol + 1] == 0 and bool((out[cap:] == GUARD).all())
        # exact fit: OK, no room for the terminal, nothing behind the capacity
        st, ol, out = _call(lib, 2, mozilla, g["len"], 64)
        assert st == 0 and ol == g["len"] and sha(out[:ol]) == g["sha256"] and bool((out[g["len"]:] == GUARD).all())
        # one byte of room: OK, still no terminal (it needs two)
        st, ol, out = _call(lib, 2, mozilla, g["len"] + 1, 64)
        assert st == 0 and ol == g["len"] and bool((out[g["len"] + 1:] == GUARD).all())
        # one short / half: MSCOMP_BUF_ERROR, and the caller's memory behind the capacity is untouched
        for cap in (g["len"] - 1, g["len"] // 2, 4096, 1, 0):
            st, ol, out = _call(lib, 2, mozilla, cap, 4096)
            assert st == m.MSCOMP_BUF_ERROR, (sliced, cap, st)
            assert bool((out[cap:] == GUARD).all()), (sliced, cap)
    finally:
        lib.mscomp_amd_debug_set_one_shot(0)


def test_four_threads_compress_large_buffers_at_once(gpu_ctx):
    """promoted from tools/dev/threads_big.py: concurrent large host-pointer calls, every result against the reference's digest"""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    lib = m.load_library()
    names = ["mozilla", "webster", "nci", "samba"]
    datas = {k: np.ascontiguousarray(corpus.by_name(k)) for k in names}
    bad = []

    def work(name, fmt, key):
        data = datas[name]
        cap = lib.ms_max_compressed_size(fmt, len(data)) + 2
        for it in range(2):
            st, ol, out = _call(lib, fmt, data, cap, 0)
            if not (st == 0 and ol == GOLD[name][key]["len"] and sha(out[:ol]) == GOLD[name][key]["sha256"]):
                bad.append((name, fmt, it, st, ol))

    ts = [threading.Thread(target=work, args=(k, 2, "lznt1")) for k in names]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not bad, bad
    # and a mix of formats at once (the Xpress formats run H2D -> kernels -> D2H on the calling thread's own stream)
    ts = [threading.Thread(target=work, args=a) for a in (("samba", 2, "lznt1"), ("samba", 4, "xpress_huff"), ("nci", 4, "xpress_huff"), ("webster", 2, "lznt1"))]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not bad, bad


def test_two_threads_compress_the_same_input_at_once(gpu_ctx, mozilla):
    """legal in the reference (only in / out must not alias): both calls page-lock the SAME input range -- the registration is shared and
    outlives the first call to finish (csrc/api.hip HostPins); through the mapped path and through the slices"""
    import ms_compress_amd as m
    lib = m.load_library()
    g = GOLD["mozilla"]["lznt1"]
    for sliced in (0, 1):
        lib.mscomp_amd_debug_set_one_shot(sliced)
        bad = []

        def work():
            cap = lib.ms_max_compressed_size(2, len(mozilla)) + 2
            for _ in range(3):
                st, ol, out = _call(lib, 2, mozilla, cap, 0)
                if not (st == 0 and ol == g["len"] and sha(out[:ol]) == g["sha256"]):
                    bad.append((st, ol))
        try:
            ts = [threading.Thread(target=work) for _ in range(3)]
            [t.start() for t in ts]; [t.join() for t in ts]
        finally:
            lib.mscomp_amd_debug_set_one_shot(0)
        assert not bad, (sliced, bad)


def test_host_batch_beside_a_large_one_shot_call_on_the_same_input(gpu_ctx, mozilla):
    """one thread compresses mozilla with the large host-pointer LZNT1 call (which page-locks its input for the duration), another reads the
    SAME input through the host-batch entry at the same time (legal: inputs may be shared): the batch entry's staged copies keep the other
    call's registration alive (csrc/api.hip host_pins_hold); every result against the reference's digests"""
    import ms_compress_amd as m
    lib = m.load_library()
    g2, g4 = GOLD["mozilla"]["lznt1"], GOLD["mozilla"]["xpress_huff"]
    bad = []

    def one_shot():
        cap = lib.ms_max_compressed_size(2, len(mozilla)) + 2
        for _ in range(4):
            st, ol, out = _call(lib, 2, mozilla, cap, 0)
            if not (st == 0 and ol == g2["len"] and sha(out[:ol]) == g2["sha256"]):
                bad.append(("one_shot", st, ol))

    def batch():
        for fmt, g in ((4, g4), (2, g2), (4, g4)):
            out = np.zeros(m.max_compressed_size(fmt, len(mozilla)) + 2, dtype=np.uint8)
            rc, lens, st = m.compress_units_host(fmt, [mozilla], [out], devices=(0,))
            if not (rc == 0 and st[0] == 0 and int(lens[0]) == g["len"] and sha(out[: int(lens[0])]) == g["sha256"]):
                bad.append(("batch", fmt, rc, int(st[0]), int(lens[0])))
    ts = [threading.Thread(target=one_shot), threading.Thread(target=batch)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not bad, bad


@pytest.mark.parametrize("fmt,key", [(3, "xpress"), (4, "xpress_huff")])
def test_whole_file_host_pointer_xpress_formats(gpu_ctx, fmt, key):
    """one 21 MB buffer through ms_compress with host pointers: bytes of the reference, exact capacity, one short -> MSCOMP_BUF_ERROR"""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    lib = m.load_library()
    data = np.ascontiguousarray(corpus.by_name("samba"))
    g = GOLD["samba"][key]
    st, ol, out = _call(lib, fmt, data, lib.ms_max_compressed_size(fmt, len(data)), 64)
    assert st == 0 and ol == g["len"] and sha(out[:ol]) == g["sha256"]
    st, ol, out = _call(lib, fmt, data, g["len"], 64)
    assert st == 0 and ol == g["len"] and bool((out[g["len"]:] == GUARD).all())
    st, ol, out = _call(lib, fmt, data, g["len"] - 1, 64)
    assert st == m.MSCOMP_BUF_ERROR and bool((out[g["len"] - 1:] == GUARD).all())


@pytest.mark.parametrize("fmt,key", [(3, "xpress"), (4, "xpress_huff")])
def test_ranged_upload_of_large_xpress_calls(gpu_ctx, mozilla, fmt, key):
    """Round 6: ms_compress(Xpress / Xpress+Huffman) of a buffer of 8 MiB or more uploads it in ranges of 256 chunks and runs the chain links and
    Find of a range under the upload of the next (csrc/api.hip one_shot). The bytes must be the batch path's and the reference's: mozilla (51 MB, 7
    ranges) against the reference's digest; sizes at and around the range boundaries and the switch-over size against the one-copy form of the
    same call (mscomp_amd_debug_set_one_shot(1)) and the HBM-resident batch path."""
    import ms_compress_amd as m
    lib = m.load_library()
    g = GOLD["mozilla"][key]
    st, ol, out = _call(lib, fmt, mozilla, lib.ms_max_compressed_size(fmt, len(mozilla)), 64)
    assert st == 0 and ol == g["len"] and sha(out[:ol]) == g["sha256"]
    for n in ((8 << 20) - 1, 8 << 20, (8 << 20) + 1, (16 << 20) + 63, (16 << 20) + 64, (16 << 20) + 65, (24 << 20) + 65536 - 3, 30_000_001):
        data = np.ascontiguousarray(mozilla[:n])
        cap = lib.ms_max_compressed_size(fmt, n)
        st, ol, out = _call(lib, fmt, data, cap, 64)
        lib.mscomp_amd_debug_set_one_shot(1)
        try:
            st1, ol1, out1 = _call(lib, fmt, data, cap, 64)
        finally:
            lib.mscomp_amd_debug_set_one_shot(0)
        assert st == 0 and st1 == 0 and ol == ol1 and bool((out[:ol] == out1[:ol1]).all()), "ranged and one-copy call differ at %d bytes" % n
        got, st2 = m.compress_units(fmt, [data.tobytes()], ctx=gpu_ctx)
        assert st2 == [0] and got[0] == out[:ol].tobytes(), "the one-shot call and the batch path differ at %d bytes" % n


# ---- fixed-seed slices of the soaks (tools/dev/fuzz_decode.py, fuzz_big.py, fuzz_sa.py) -------------------------------------------------
def _gen(rnd, n, kind=None):
    kind = rnd.randrange(6) if kind is None else kind
    if kind == 0:
        return rnd.randbytes(n)
    if kind == 1:
        return bytes(rnd.choice(b"ab") for _ in range(n))
    if kind == 2:
        return bytes(n)
    out = bytearray(rnd.randbytes(rnd.randint(1, 64)))
    while len(out) < n:
        if rnd.random() < (0.5 if kind == 3 else 0.15):
            out += rnd.randbytes(rnd.randint(1, 6))
        ln = rnd.choice((3, 5, 9, 10, 17, 24, 25, 40, 279, 280, 300, 2000, 70000 if kind == 5 else 33))
        off = rnd.randint(1, min(len(out), 65535 if kind == 4 else 9000))
        chunk = bytes(out[-off:]) if off >= ln else None
        if chunk is not None:
            out += chunk[:ln]
        else:
            for _ in range(ln):
                out.append(out[-off])
    return bytes(out[:n])


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_soak_slice_small_streams(oracle, gpu_ctx, fmt):
    """seed 11 of fuzz_decode.py, one round: structured inputs -> GPU compress == checker; valid, cut, corrupted and short-capacity
    streams -> GPU decompress: status and bytes are the checker's (undefined-behaviour cases of the reference skipped)"""
    import ms_compress_amd as m
    rnd = random.Random(1100 + fmt)
    units = [_gen(rnd, rnd.choice((0, 1, 5, 300, 4096, 4097, 65536, 65537, 70000, rnd.randint(1, 150000)))) for _ in range(24)]
    comp, st = m.compress_units(fmt, units, ctx=gpu_ctx)
    for u, c, s in zip(units, comp, st):
        es, ec = oracle.oracle_compress(fmt, u)
        assert s == 0 and es == 0 and c == ec, ("compress", fmt, len(u))
    streams = []
    for u, c in zip(units, comp):
        streams.append((c, len(u)))
        if len(c) > 8:
            streams.append((c[: rnd.randrange(1, len(c))], len(u)))
            b = bytearray(c)
            for _ in range(rnd.randint(1, 3)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
            streams.append((bytes(b), len(u) + rnd.choice((0, 0, 7, 5000))))
            streams.append((c, max(0, len(u) - rnd.choice((1, 2, 100, len(u) // 2 + 1)))))
    outs, sts = m.decompress_units(fmt, [s for s, _ in streams], [c for _, c in streams], ctx=gpu_ctx)
    for (s_, cap), o, st_ in zip(streams, outs, sts):
        so, oo, undefined = oracle.oracle_decompress_ex(fmt, s_, cap)
        if undefined:
            continue
        assert st_ == so and (so != 0 or o == oo), (fmt, len(s_), cap, st_, so)


@pytest.mark.parametrize("fmt", [3, 4])
def test_soak_slice_large_units(oracle, gpu_ctx, fmt):
    """seed 21 of fuzz_big.py, cut down: units of 0.6-1.5 MB (segment walk / chunk-parallel walk / all-CU byte stage of the decoders;
    multi-chunk streams of the compressors): compress == checker, round trip and a corrupted copy == checker"""
    import ms_compress_amd as m
    rnd = random.Random(2100 + fmt)
    units = [_gen(rnd, rnd.randint(600_000, 1_500_000), kind) for kind in (3, 4, 5)] + [cases.mixed_buffer()]
    comp, st = m.compress_units(fmt, units, ctx=gpu_ctx)
    for u, c, s in zip(units, comp, st):
        assert s == 0 and c == oracle.oracle_compress(fmt, u)[1], ("compress", fmt, len(u))
    streams = [(c, len(u)) for u, c in zip(units, comp)]
    for u, c in zip(units, comp):
        b = bytearray(c)
        b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        streams.append((bytes(b), len(u)))
        streams.append((c[: len(c) // 2], len(u)))
        streams.append((c, len(u) - 1))
    outs, sts = m.decompress_units(fmt, [s for s, _ in streams], [c for _, c in streams], ctx=gpu_ctx)
    for (s_, cap), o, st_ in zip(streams, outs, sts):
        so, oo, undefined = oracle.oracle_decompress_ex(fmt, s_, cap)
        if undefined:
            continue
        assert st_ == so and (so != 0 or o == oo), (fmt, len(s_), cap, st_, so)


def test_soak_slice_sa_dictionary(oracle, gpu_ctx):
    """seed 1 of fuzz_sa.py, cut down: the suffix-array dictionary flavour of LZNT1 on structured inputs against its checker"""
    import ms_compress_amd as m
    rnd = random.Random(3101)
    units = [_gen(rnd, rnd.choice((1, 5, 300, 4096, 4097, 12288, rnd.randint(1, 60000)))) for _ in range(40)]
    units += [bytes(rnd.choice(b"abc") for _ in range(9000))]
    gpu_ctx.lib.mscomp_amd_set_lznt1_sa_dict(1)
    try:
        comp, st = m.compress_units(2, units, ctx=gpu_ctx)
    finally:
        gpu_ctx.lib.mscomp_amd_set_lznt1_sa_dict(0)
    for u, c, s in zip(units, comp, st):
        es, ec = oracle.oracle_compress_sa(u)
        assert s == 0 and es == 0 and c == ec, len(u)

"""-m gpu: mscomp_amd_compress_units_host (include/mscomp_amd.h; SURVEY.md 8e + 8f-3) -- host pointers in, host pointers out, one or several
device ranges in ONE process. Every unit is one ms_compress call of the reference: bytes against the reference's digests
(tests/golden/corpus_full.json) and against the checker, MSCOMP_BUF_ERROR for short capacities with nothing written behind them, the
uncounted LZNT1 00 00. Two ranges on the SAME GPU run everywhere (two worker threads, two contexts, pipelined batches); two ranges on
two GPUs run where the box has them (skipped below 2 devices: the driver's 8-GPU node executes it)."""
import hashlib
import json
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "corpus_full.json")))
KEY = {2: "lznt1", 3: "xpress_units64k", 4: "xpress_huff"}
GUARD = 0x5A


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def _corpus_job(fmt, names):
    """(inputs, per-file unit counts): LZNT1 / Xpress+Huffman one unit per file, Xpress the file cut into 64 KiB units (views of one array)"""
    from ms_compress_amd import corpus
    ins, counts = [], []
    for nm in names:
        data = np.ascontiguousarray(corpus.by_name(nm))
        if fmt == 3:
            units = [data[o:o + 65536] for o in range(0, len(data), 65536)]
        else:
            units = [data]
        ins += units; counts.append(len(units))
    return ins, counts


def _check_files(fmt, names, counts, outs, lens, st):
    u = 0
    for nm, k in zip(names, counts):
        assert bool((st[u:u + k] == 0).all()), nm
        blob = b"".join(bytes(outs[u + i][: int(lens[u + i])]) for i in range(k))
        g = GOLD[nm][KEY[fmt]]
        assert len(blob) == g["len"] and sha(blob) == g["sha256"], (fmt, nm)
        u += k


@pytest.mark.parametrize("fmt", [2, 3, 4])
@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0)])
def test_files_through_host_pointers(gpu_ctx, fmt, devices, monkeypatch):
    """five corpus files (69 MB) in small batches (several per range: the pipeline really alternates its two buffer sets); separate output
    arrays per unit (one download each) -- against the reference's digests"""
    import ms_compress_amd as m
    monkeypatch.setenv("MSCOMP_AMD_HOST_BATCH_MB", "8")          # (read once per process: the first test decides; small either way)
    names = ["xml", "ooffice", "sao", "dickens", "samba"]
    ins, counts = _corpus_job(fmt, names)
    outs = [np.full(m.max_compressed_size(fmt, a.size) + 2, GUARD, dtype=np.uint8) for a in ins]
    rc, lens, st = m.compress_units_host(fmt, ins, outs, devices=devices)
    assert rc == 0
    _check_files(fmt, names, counts, outs, lens, st)
    if fmt == 2:                                                  # the uncounted End_of_buffer bytes behind every stream (lznt1_compress.cpp:270)
        for o, l in zip(outs, lens):
            assert o[int(l)] == 0 and o[int(l) + 1] == 0


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_contiguous_layout_and_short_capacities(oracle, gpu_ctx, fmt):
    """outputs laid out capacity after capacity in ONE array (the layout mscomp_amd_plan_layout gives: one download per batch), edge-case
    units, and capacities that are exact / one short / zero: statuses and bytes of the checker, guards behind every capacity untouched"""
    import ms_compress_amd as m
    rng = np.random.default_rng(7)
    units = [u for u in cases.edge_cases()[::5] if len(u) < 200000] + [cases.mixed_buffer()[:150000], b"", b"a"]
    exp = [oracle.oracle_compress(fmt, u)[1] for u in units]
    caps = []
    for i, e in enumerate(exp):
        caps.append([len(e), max(0, len(e) - 1), m.max_compressed_size(fmt, len(units[i])) + 2, 0][i % 4])
    ins = [np.frombuffer(u, dtype=np.uint8) if len(u) else np.zeros(0, dtype=np.uint8) for u in units]
    blob = np.full(sum(caps) + 64, GUARD, dtype=np.uint8)
    outs, pos = [], 0
    for c in caps:
        outs.append(blob[pos:pos + c]); pos += c
    rc, lens, st = m.compress_units_host(fmt, ins, outs, devices=(0, 0))
    assert rc == 0
    for i, (u, e, c) in enumerate(zip(units, exp, caps)):
        if c >= len(e):
            assert st[i] == 0 and int(lens[i]) == len(e) and bytes(outs[i][: len(e)]) == e, (fmt, i, len(u), c)
        else:
            assert st[i] == m.MSCOMP_BUF_ERROR and int(lens[i]) == 0, (fmt, i, len(u), c)
    assert bool((blob[pos:] == GUARD).all())
    # separate arrays with guards: nothing behind a capacity is touched, whatever the status
    outs2 = [np.full(c + 32, GUARD, dtype=np.uint8) for c in caps]
    rc, lens2, st2 = m.compress_units_host(fmt, ins, [o[:c] for o, c in zip(outs2, caps)], devices=(0,))
    assert rc == 0 and np.array_equal(st2, st) and np.array_equal(lens2, lens)
    for o, c in zip(outs2, caps):
        assert bool((o[c:] == GUARD).all())


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_decoders_through_host_pointers(oracle, gpu_ctx, fmt):
    """mscomp_amd_decompress_units_host: valid, cut and corrupted streams and short capacities, two ranges on one GPU: statuses and bytes of the
    checker (one ms_decompress call per unit), guards behind every capacity untouched"""
    import random
    import ms_compress_amd as m
    rnd = random.Random(40 + fmt)
    units = [u for u in cases.edge_cases()[::7] if 0 < len(u) < 150000] + [cases.mixed_buffer()[:200000]]
    comp = [oracle.oracle_compress(fmt, u)[1] for u in units]
    streams, caps = [], []
    for u, c in zip(units, comp):
        streams.append(c); caps.append(len(u))
        streams.append(c); caps.append(max(0, len(u) - 1))
        if len(c) > 8:
            streams.append(c[: len(c) // 2]); caps.append(len(u))
            b = bytearray(c); b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
            streams.append(bytes(b)); caps.append(len(u) + 5)
    ins = [np.frombuffer(s, dtype=np.uint8) for s in streams]
    outs = [np.full(c + 16, GUARD, dtype=np.uint8) for c in caps]
    rc, lens, st = m.decompress_units_host(fmt, ins, [o[:c] for o, c in zip(outs, caps)], devices=(0, 0))
    assert rc == 0
    for i, (s_, c) in enumerate(zip(streams, caps)):
        so, oo, undefined = oracle.oracle_decompress_ex(fmt, s_, c)
        if undefined:
            continue
        assert st[i] == so and (so != 0 or (int(lens[i]) == len(oo) and bytes(outs[i][: len(oo)]) == oo)), (fmt, i, len(s_), c, st[i], so)
        assert bool((outs[i][c:] == GUARD).all())


def test_generous_capacities_do_not_become_allocations(oracle, gpu_ctx):
    """*out_len far beyond what the format can produce is legal in the reference: the device copy is sized by ms_max_compressed_size (no way to
    hand over a terabyte array through numpy: the C entry is called with capacities larger than the arrays, whose real size still holds any stream)"""
    import ctypes as C
    import ms_compress_amd as m
    lib = m.load_library()
    data = np.frombuffer(cases.mixed_buffer()[:100000], dtype=np.uint8)
    for fmt in (2, 3, 4):
        exp = oracle.oracle_compress(fmt, data.tobytes())[1]
        out = np.full(m.max_compressed_size(fmt, data.size) + 64, GUARD, dtype=np.uint8)
        ip = (C.c_void_p * 1)(data.ctypes.data); il = (C.c_size_t * 1)(data.size)
        op = (C.c_void_p * 1)(out.ctypes.data); oc = (C.c_size_t * 1)(1 << 40)
        ol = (C.c_size_t * 1)(0); st = (C.c_int * 1)(-9)
        dv = (C.c_int * 1)(0)
        assert lib.mscomp_amd_compress_units_host(fmt, 1, dv, 1, ip, il, op, oc, ol, st) == 0
        assert st[0] == 0 and ol[0] == len(exp) and bytes(out[: len(exp)]) == exp


def test_generous_decompress_capacities(oracle, gpu_ctx):
    """the decoders' side of the same rule: capacities of a terabyte (the C entry is called with them; the arrays hold what the streams decode
    to) must not become device allocations -- LZNT1 is bounded by 4096 bytes per 3 of input, the Xpress formats start at 16 x the input and a
    unit that needs more (5 MB of zeros behind a 100-byte stream) is decoded again with room, the others in its batch undisturbed"""
    import ctypes as C
    import ms_compress_amd as m
    lib = m.load_library()
    plain = [bytes(5_000_000), cases.mixed_buffer()[:120000], b"abc" * 70000, bytes(range(256)) * 40]
    for fmt in (2, 3, 4):
        streams = [oracle.oracle_compress(fmt, u)[1] for u in plain]
        n = len(plain)
        ins = [np.frombuffer(s_, dtype=np.uint8) for s_ in streams]
        outs = [np.full(len(u) + 64, GUARD, dtype=np.uint8) for u in plain]
        ip = (C.c_void_p * n)(*[a.ctypes.data for a in ins]); il = (C.c_size_t * n)(*[a.size for a in ins])
        op = (C.c_void_p * n)(*[a.ctypes.data for a in outs]); oc = (C.c_size_t * n)(*([1 << 40] * n))
        ol = (C.c_size_t * n)(); st = (C.c_int * n)(*([-9] * n))
        dv = (C.c_int * 1)(0)
        assert lib.mscomp_amd_decompress_units_host(fmt, 1, dv, n, ip, il, op, oc, ol, st) == 0
        for i, u in enumerate(plain):
            assert st[i] == 0 and ol[i] == len(u) and bytes(outs[i][: len(u)]) == u, (fmt, i)
            assert bool((outs[i][len(u):] == GUARD).all())


def test_argument_errors(gpu_ctx):
    import ms_compress_amd as m
    a = np.zeros(100, dtype=np.uint8); o = np.zeros(200, dtype=np.uint8)
    assert m.compress_units_host(1, [a], [o])[0] == m.MSCOMP_ARG_ERROR            # no such format (mscomp.cpp:115)
    assert m.compress_units_host(2, [a], [o], devices=(99,))[0] == m.MSCOMP_ARG_ERROR
    rc, lens, st = m.compress_units_host(2, [], [])
    assert rc == 0 and len(lens) == 0


def test_two_gpus_in_one_process(gpu_ctx):
    """SURVEY 8e on real hardware: two contexts on two devices of one process (the per-device function attributes, the lane-order check
    and the worker pool are per device). Needs two GPUs: skipped on the one-GPU box, runs on the driver's multi-GPU node."""
    import os
    import torch
    import ms_compress_amd as m
    # MSCOMP_AMD_TEST_DEVICES=0,1 (any two ordinals, e.g. "2,5"): which two devices of a multi-GPU lease to use -- no edit needed to run this there
    env = os.environ.get("MSCOMP_AMD_TEST_DEVICES")
    devs = tuple(int(x) for x in env.split(",")) if env else (0, 1)
    if len(devs) != 2 or devs[0] == devs[1] or torch.cuda.device_count() <= max(devs):
        pytest.skip("needs two GPUs (MSCOMP_AMD_TEST_DEVICES=a,b picks them; %d visible)" % torch.cuda.device_count())
    names = ["xml", "ooffice", "sao", "dickens", "samba", "osdb"]
    for fmt in (2, 3, 4):
        ins, counts = _corpus_job(fmt, names)
        outs = [np.full(m.max_compressed_size(fmt, a.size) + 2, GUARD, dtype=np.uint8) for a in ins]
        rc, lens, st = m.compress_units_host(fmt, ins, outs, devices=devs)
        assert rc == 0
        _check_files(fmt, names, counts, outs, lens, st)
    # and the batch interface directly: a context per device, the same plan on both
    data = cases.mixed_buffer()
    for fmt in (2, 3, 4):
        res = []
        for dev in devs:
            ctx = m.Context(device=dev)
            out, st = m.compress_units(fmt, [data, data[:70000]], ctx=ctx)
            ctx.close()
            res.append(out)
        assert res[0] == res[1]

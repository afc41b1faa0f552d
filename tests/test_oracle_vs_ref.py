"""CPU (dev container): pin the oracle against the REAL reference compiled from /root/reference (oracle/_ref).
Skipped where oracle/_ref was not built. Also pins the Huffman builders against the reference's header-only template."""
import os
import random
import subprocess

import numpy as np
import pytest

import cases

HERE = os.path.dirname(os.path.abspath(__file__))
HUFF_REF = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "huff_ref")


@pytest.fixture(scope="module")
def ref(oracle):
    r = oracle.load_ref()
    if r is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    return r


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_fuzz_families(oracle, ref, fmt):
    for data in cases.edge_cases() + [cases.mixed_buffer()]:
        s1, o1 = oracle.oracle_compress(fmt, data)
        s2, o2 = oracle.ref_compress(fmt, data)
        assert (s1, o1) == (s2, o2), (fmt, len(data))
        if len(data) or fmt != 3:   # the reference's own Xpress decoder rejects the empty stream it encodes
            s3, d3 = oracle.ref_decompress(fmt, o1, len(data))
            assert s3 == 0 and d3 == data


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_corpus_slices(oracle, ref, fmt):
    from ms_compress_amd import corpus
    for i in range(12):
        data = corpus.file_bytes(i, 400_000).tobytes()
        assert oracle.oracle_compress(fmt, data) == oracle.ref_compress(fmt, data), corpus.NAMES[i]


def test_capacity_semantics(oracle, ref):
    """status for short output buffers matches the reference (BUF_ERROR iff the full output does not fit)."""
    data = cases.mixed_buffer()[:20000]
    for fmt in (2, 3, 4):
        full = oracle.ref_compress(fmt, data)[1]
        for cap in (len(full) + 2, len(full), len(full) - 1, len(full) // 2, 300):
            assert oracle.oracle_compress(fmt, data, cap=cap)[0] == oracle.ref_compress(fmt, data, cap=cap)[0], (fmt, cap)


def _huff_ref(mode, counts):
    p = subprocess.run([HUFF_REF, mode], input=" ".join(str(int(c)) for c in counts), capture_output=True, text=True, check=True)
    return [int(x) for x in p.stdout.split()]


def test_huffman_builders(oracle, ref):
    if not os.path.exists(HUFF_REF):
        pytest.skip("huff_ref not built")
    lib = oracle.load_oracle()
    rnd = random.Random(5)
    for case in range(40):
        kind = case % 4
        if kind == 0:
            counts = [rnd.randint(0, 3) * rnd.randint(0, 300) for _ in range(512)]
        elif kind == 1:
            counts = [int(2 ** rnd.uniform(0, 16)) if rnd.random() < 0.3 else 0 for _ in range(512)]      # deep trees -> rescale loop
        elif kind == 2:
            counts = [rnd.randint(200, 300) for _ in range(256)] + [0] * 256
            counts[256] = 1                                                                                # incompressible + EOS
        else:
            counts = [1 if rnd.random() < 0.05 else 0 for _ in range(512)]
        if sum(counts) == 0:
            counts[0] = 1
        c = np.asarray(counts, dtype=np.uint32)
        lens = np.zeros(512, dtype=np.uint8)
        lib.orc_huff_lengths(c.ctypes.data, lens.ctypes.data)
        assert lens.tolist() == _huff_ref("fast", counts), "CreateCodes case %d" % case
        lib.orc_huff_lengths_slow(c.ctypes.data, lens.ctypes.data)
        assert lens.tolist() == _huff_ref("slow", counts), "CreateCodesSlow case %d" % case


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_decompress_semantics(oracle, ref, fmt):
    """The restated decoders return the reference's status and bytes on valid, truncated, concatenated and corrupted
    streams, for exact / larger / smaller capacities (streams on which the reference is undefined are not put to it)."""
    streams = cases.decode_streams(fmt, lambda d: oracle.oracle_compress(fmt, d)[1])
    asked = 0
    for stream, cap in streams:
        so, oo, undefined = oracle.oracle_decompress_ex(fmt, stream, cap)
        if undefined:
            assert so == -3
            continue
        sr, orf = oracle.ref_decompress(fmt, stream, cap)
        assert (so, oo) == (sr, orf), (fmt, len(stream), cap, so, sr)
        asked += 1
    assert asked > len(streams) * 0.9


def test_threaded_cpu_baseline_driver(oracle, ref):
    """bench.py's CPU baselines: orc_time_units drives the reference's ms_compress / ms_decompress over independent units on C threads
    and must hand back what single calls give"""
    data = cases.mixed_buffer()
    units = [data[o:o + 65536] for o in range(0, 5 * 65536, 65536)]
    for fmt in (2, 3, 4):
        caps = [ref.ms_max_compressed_size(fmt, len(u)) for u in units]
        dt, st, ln = oracle.time_units(ref.ms_compress, fmt, units, caps, 3, 2)
        comp = [oracle.ref_compress(fmt, u)[1] for u in units]
        assert dt > 0 and (st == 0).all() and [int(x) for x in ln] == [len(c) for c in comp]
        dt, st, ln = oracle.time_units(ref.ms_decompress, fmt, comp, [len(u) for u in units], 3, 2)
        assert (st == 0).all() and [int(x) for x in ln] == [len(u) for u in units]
        dt, st, ln = oracle.time_units(None, fmt, units, caps, 2, 1)                 # the oracle's own compressor
        assert (st == 0).all() and [int(x) for x in ln] == [len(c) for c in comp]


def test_lznt1_sa_dictionary_flavour(oracle):
    """the oracle's suffix-array dictionary against the reference compiled with -DMSCOMP_WITH_LZNT1_SA_DICT (oracle/_ref/libMSCompression_sa.so)"""
    if oracle.load_ref_sa() is None:
        pytest.skip("oracle/_ref/libMSCompression_sa.so not built")
    from ms_compress_amd import corpus
    units = cases.edge_cases() + [cases.mixed_buffer()] + [corpus.file_bytes(i, 200_000).tobytes() for i in range(12)]
    rng = random.Random(11)
    units += [bytes(rng.choice(b"abc") for _ in range(rng.randint(4, 5000))) for _ in range(60)]
    for data in units:
        assert oracle.oracle_compress_sa(data) == oracle.ref_compress_sa(data), len(data)
    data = cases.mixed_buffer()[:20000]
    full = oracle.ref_compress_sa(data)[1]
    for cap in (len(full), len(full) - 1, 300):
        assert oracle.oracle_compress_sa(data, cap=cap)[0] == oracle.ref_compress_sa(data, cap=cap)[0]

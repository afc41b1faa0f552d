"""-m gpu: BASELINE.json's full-size configurations. The oracle runs multi-threaded on the host (seconds), the comparison
is bit-exact via one digest per unit; round trips through the oracle's decoders are the size-independent property."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_units(oracle, fmt, blob, in_off, in_len, threads=32):
    lib = oracle.load_oracle()
    n = len(in_len)
    caps = np.array([lib.orc_max_compressed_size(fmt, int(x)) + 2 for x in in_len], dtype=np.uint64)
    out_off = np.zeros(n + 1, dtype=np.uint64); out_off[1:] = np.cumsum(caps)
    io = np.zeros(n + 1, dtype=np.uint64); io[:n] = in_off; io[n] = in_off[-1] + in_len[-1] if n else 0
    # orc_compress_units takes contiguous units (in_off[i+1]-in_off[i] = length): the blobs here are contiguous
    assert all(int(io[i]) + int(in_len[i]) == int(io[i + 1]) for i in range(n))
    out = np.zeros(int(out_off[-1]) + 8, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint64); status = np.zeros(n, dtype=np.int32)
    lib.orc_compress_units(fmt, blob.ctypes.data, io.ctypes.data, n, out.ctypes.data, out_off.ctypes.data,
                           out_len.ctypes.data, status.ctypes.data, threads)
    assert (status == 0).all()
    return out, out_off, out_len


def _gpu_units(m, ctx, fmt, blob, in_off, in_len):
    import torch
    caps = [m.max_compressed_size(fmt, int(x)) + 2 for x in in_len]
    out_off, out_total = m.pack_offsets(caps)
    dev = torch.device("cuda", ctx.device)
    d_in = torch.from_numpy(blob).to(dev)
    d_out = torch.zeros(out_total + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(len(in_len), dtype=torch.int64, device=dev)
    d_st = torch.zeros(len(in_len), dtype=torch.int32, device=dev)
    plan = m.Plan(ctx, fmt, in_off, in_len, out_off, caps)
    plan.execute(d_in, d_out, d_len, d_st)
    torch.cuda.synchronize()
    plan.close()
    assert bool((d_st == 0).all().item())
    return d_out.cpu().numpy(), out_off, d_len.cpu().numpy()


def _compare(m, oracle, ctx, fmt, blob, in_off, in_len, roundtrip_units=4):
    go, goff, glen = _gpu_units(m, ctx, fmt, blob, in_off, in_len)
    oo, ooff, olen = _oracle_units(oracle, fmt, blob, in_off, in_len)
    assert np.array_equal(glen.astype(np.uint64), olen), "compressed sizes differ"
    for i in range(len(in_len)):
        a = go[int(goff[i]): int(goff[i]) + int(glen[i])]; b = oo[int(ooff[i]): int(ooff[i]) + int(olen[i])]
        assert np.array_equal(a, b), "unit %d differs" % i
    for i in np.linspace(0, len(in_len) - 1, min(roundtrip_units, len(in_len))).astype(int):   # decode what the GPU wrote
        comp = bytes(go[int(goff[i]): int(goff[i]) + int(glen[i])])
        st, back = oracle.oracle_decompress(fmt, comp, int(in_len[i]))
        assert st == 0 and back == blob[int(in_off[i]): int(in_off[i]) + int(in_len[i])].tobytes()


def test_config2_lznt1_mozilla(oracle, gpu_ctx):
    """config 2: LZNT1, 'mozilla' (51 220 480 B = 12 505 chunks of 4 KiB) as one ms_compress-equivalent unit."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    data = corpus.by_name("mozilla")
    # the oracle is single-threaded per unit: split on 4 KiB boundaries (LZNT1 chunks are independent, concatenation is exact)
    step = 4096 * 512
    in_off = np.arange(0, len(data), step, dtype=np.uint64)
    in_len = np.minimum(step, len(data) - in_off).astype(np.uint64)
    _compare(m, oracle, gpu_ctx, 2, data, in_off, in_len)
    whole, st = m.compress_units(2, [data.tobytes()], ctx=gpu_ctx)          # and as ONE unit: identical to the concatenation
    parts, _ = m.compress_units(2, [data[int(o): int(o) + int(l)].tobytes() for o, l in zip(in_off, in_len)], ctx=gpu_ctx)
    assert st == [0] and whole[0] == b"".join(parts)


def test_config3_xpress_silesia_units(oracle, gpu_ctx):
    """config 3: Xpress, every file cut into independent 64 KiB units (3 239 units over the 12 members)."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    for i in range(12):
        data = corpus.file_bytes(i)
        in_off = np.arange(0, len(data), 65536, dtype=np.uint64)
        in_len = np.minimum(65536, len(data) - in_off).astype(np.uint64)
        _compare(m, oracle, gpu_ctx, 3, data, in_off, in_len, roundtrip_units=2)


def test_config4_xpress_huff_silesia(oracle, gpu_ctx):
    """config 4: Xpress+Huffman, file mode (one unit per file, chunks see the previous 64 KiB) and unit mode."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    files = [corpus.file_bytes(i) for i in range(12)]
    blob = np.concatenate(files)
    in_len = np.array([len(f) for f in files], dtype=np.uint64)
    in_off = np.zeros(12, dtype=np.uint64); in_off[1:] = np.cumsum(in_len)[:-1]
    _compare(m, oracle, gpu_ctx, 4, blob, in_off, in_len, roundtrip_units=3)          # file mode
    data = files[1]                                                                    # unit mode on mozilla
    uo = np.arange(0, len(data), 65536, dtype=np.uint64)
    ul = np.minimum(65536, len(data) - uo).astype(np.uint64)
    _compare(m, oracle, gpu_ctx, 4, data, uo, ul, roundtrip_units=2)


def test_whole_buffer_xpress_large(oracle, gpu_ctx):
    """ms_compress semantics for a single Xpress stream far larger than 64 KiB (one sequential stream, 8 KiB window)."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    data = corpus.by_name("samba", 3_000_000)
    got, st = m.compress_units(3, [data.tobytes()], ctx=gpu_ctx)
    assert st == [0] and hashlib.sha256(got[0]).digest() == hashlib.sha256(oracle.oracle_compress(3, data.tobytes())[1]).digest()


@pytest.mark.parametrize("fmt", [3, 4])
def test_whole_files_come_back(gpu_ctx, fmt):
    """All 12 files as 12 whole buffers -- one Xpress stream each (segment walks, csrc/decompress.hip xps_*), one Xpress+Huffman buffer each
    (chunk-parallel walk with token scratch) -- compressed (digests of the REFERENCE's output: tests/golden/corpus_full.json) and decompressed in one
    batch; the bytes of the large units come from csrc/lzglobal.hip. Also through the token-at-a-time Xpress kernel (mode 1) and without the
    segment walk (mode 2): the same bytes."""
    import json
    import os
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "corpus_full.json")))
    files = [corpus.file_bytes(i).tobytes() for i in range(12)]
    comp, st = m.compress_units(fmt, files, ctx=gpu_ctx)
    key = {3: "xpress", 4: "xpress_huff"}[fmt]
    for name, c, s in zip(corpus.NAMES, comp, st):
        assert s == 0 and len(c) == gold[name][key]["len"] and hashlib.sha256(c).hexdigest() == gold[name][key]["sha256"], name
    for mode in ((0, 2) if fmt == 3 else (0,)):          # (mode 1, one token per step, would take seconds for 51 MB: covered at small sizes)
        gpu_ctx.lib.mscomp_amd_debug_set_xpress_decoder(mode)
        try:
            back, st2 = m.decompress_units(fmt, comp, [len(f) for f in files], ctx=gpu_ctx)
        finally:
            gpu_ctx.lib.mscomp_amd_debug_set_xpress_decoder(0)
        for name, b, f, s in zip(corpus.NAMES, back, files, st2):
            assert s == 0 and b == f, (name, mode)

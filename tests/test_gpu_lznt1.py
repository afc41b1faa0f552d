"""-m gpu: LZNT1 HIP path vs the oracle, through the C-ABI (bit-exact)."""
import pytest

import cases

pytestmark = pytest.mark.gpu
FMT = 2


def _check(m, oracle, units, ctx):
    got, st = m.compress_units(FMT, units, ctx=ctx)
    for i, (u, g, s) in enumerate(zip(units, got, st)):
        es, exp = oracle.oracle_compress(FMT, u)
        assert es == 0 and s == 0, (i, len(u), s)
        assert g == exp, "unit %d (len %d): GPU bytes differ from oracle (got %d B, expected %d B)" % (i, len(u), len(g), len(exp))


def test_lznt1_edge_sizes(oracle, gpu_ctx):
    import ms_compress_amd as m
    _check(m, oracle, cases.edge_cases(), gpu_ctx)


def test_lznt1_corpus_slices(oracle, gpu_ctx):
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    units = [corpus.file_bytes(i, 400_000).tobytes() for i in range(12)] + [cases.mixed_buffer()]
    _check(m, oracle, units, gpu_ctx)


def test_lznt1_one_shot_abi(oracle):
    """ms_compress with host pointers (drop-in path) incl. End_of_buffer and BUF_ERROR behaviour."""
    import ms_compress_amd as m
    data = cases.mixed_buffer()[:50_000]
    exp = oracle.oracle_compress(FMT, data)[1]
    assert m.compress(FMT, data) == exp
    assert m.compress(FMT, data, out_capacity=len(exp)) == exp          # exact fit, no room for 00 00
    with pytest.raises(m.MSCompError) as e:
        m.compress(FMT, data, out_capacity=len(exp) - 1)
    assert e.value.status == m.MSCOMP_BUF_ERROR
    assert m.compress(FMT, b"") == b""

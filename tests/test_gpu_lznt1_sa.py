"""GPU (MI355X): LZNT1 with the suffix-array dictionary flavour (csrc/lznt1_sa.hip) -- byte-exact against the oracle's
restatement of LZNT1Dictionary_SA.h and against tests/golden/lznt1_sa.json, written by the reference compiled with
-DMSCOMP_WITH_LZNT1_SA_DICT (tools/make_golden_sa.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
sha = lambda b: hashlib.sha256(b).hexdigest()


@pytest.fixture()
def sa_mode(gpu_ctx):
    lib = gpu_ctx.lib
    assert lib.mscomp_amd_get_lznt1_sa_dict() == 0
    lib.mscomp_amd_set_lznt1_sa_dict(1)
    yield gpu_ctx
    lib.mscomp_amd_set_lznt1_sa_dict(0)


def test_edge_families_and_adversarial_chunks(oracle, sa_mode):
    import ms_compress_amd as m
    units = list(cases.edge_cases())
    rng = np.random.default_rng(17)
    for period in (1, 2, 3, 5, 64, 65, 1023, 2048):      # long repeats: the doubling rounds run to the end, LCPs of thousands, deep stacks
        base = rng.integers(0, 256, period, dtype=np.uint8)
        units.append(np.tile(base, 9000 // period + 1)[:9000].tobytes())
    units += [bytes(rng.choice(np.frombuffer(b"abc", np.uint8), int(n))) for n in rng.integers(4, 9000, 40)]   # many equal-length candidates: the tie rules
    units.append(cases.mixed_buffer())
    got, st = m.compress_units(2, units, ctx=sa_mode)
    differ = 0
    for i, (u, g, s) in enumerate(zip(units, got, st)):
        es, exp = oracle.oracle_compress_sa(bytes(u))
        assert es == 0 and s == 0 and g == exp, "unit %d (%d bytes): GPU bytes differ from the suffix-array oracle" % (i, len(u))
        differ += exp != oracle.oracle_compress(2, bytes(u))[1]
    assert differ > 20                                   # the flavour is a different byte stream, not the default one by another name
    g = json.load(open(os.path.join(G, "lznt1_sa.json")))["edge_families"]
    h = hashlib.sha256()
    for o in got[:g["units"]]:
        h.update(len(o).to_bytes(8, "little")); h.update(o)
    assert h.hexdigest() == g["sha256"]


def test_corpus_golden_and_round_trip(sa_mode):
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    g = json.load(open(os.path.join(G, "lznt1_sa.json")))
    names = corpus.NAMES + ["mixed_buffer"]
    units = [corpus.file_bytes(i, g["corpus"][n]["input_len"]).tobytes() for i, n in enumerate(corpus.NAMES)] + [cases.mixed_buffer()]
    got, st = m.compress_units(2, units, ctx=sa_mode)
    for n, o, s in zip(names, got, st):
        assert s == 0 and len(o) == g["corpus"][n]["len"] and sha(o) == g["corpus"][n]["sha256"], n
    back, st = m.decompress_units(2, got, [len(u) for u in units], ctx=sa_mode)
    assert all(s == 0 for s in st) and all(b == u for b, u in zip(back, units))
    for name, k in g["kat"].items():
        data = bytes.fromhex(k["input_hex"]) if k["input_hex"] is not None else {"abc*100": b"abc" * 100, "zeros4097": bytes(4097)}[name]
        out, st = m.compress_units(2, [data], ctx=sa_mode)
        assert st[0] == 0 and out[0].hex() == k["hex"], name


def test_switching_back_replays_the_default_flavour(oracle, gpu_ctx):
    """the process default is read when a plan is CREATED: same plan shape, other bytes, and back."""
    import ms_compress_amd as m
    u = [bytes(np.random.default_rng(3).choice(np.frombuffer(b"abc", np.uint8), 30000))]
    a, _ = m.compress_units(2, u, ctx=gpu_ctx)
    gpu_ctx.lib.mscomp_amd_set_lznt1_sa_dict(1)
    try:
        b, _ = m.compress_units(2, u, ctx=gpu_ctx)
    finally:
        gpu_ctx.lib.mscomp_amd_set_lznt1_sa_dict(0)
    c, _ = m.compress_units(2, u, ctx=gpu_ctx)
    assert a[0] == c[0] == oracle.oracle_compress(2, u[0])[1] and b[0] == oracle.oracle_compress_sa(u[0])[1] and a[0] != b[0]


def test_host_pointer_one_shot_calls(oracle, sa_mode):
    """ms_compress with host pointers (mscomp.h; the thread's cached plans notice the switch), small and large (>= 36 MiB: the caller's
    buffers mapped, one launch -- csrc/api.hip lznt1_zero_copy; the sliced path is tests/test_gpu_hostptr.py's): the flavour's bytes
    through the drop-in entry point itself."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    small = bytes(np.random.default_rng(4).choice(np.frombuffer(b"abc", np.uint8), 50000))
    out = m.compress(2, small)                                 # (raises MSCompError on a status other than MSCOMP_OK)
    assert out == oracle.oracle_compress_sa(small)[1] and out != oracle.oracle_compress(2, small)[1]
    big = corpus.file_bytes(1).tobytes()                       # mozilla, 51 220 480 B
    g = json.load(open(os.path.join(G, "corpus_full.json")))["mozilla"]
    out = m.compress(2, big)
    assert len(out) == g["lznt1_sa"]["len"] and sha(out) == g["lznt1_sa"]["sha256"]


def test_the_flavour_belongs_to_the_plan_and_can_be_set_per_context(oracle, gpu_ctx):
    """mscomp_amd_ctx_set_lznt1_sa_dict: two contexts of one process with different flavours; a plan created under one setting keeps its
    bytes when the setting (its context's or the process default) changes afterwards -- nothing process-wide reaches into an existing plan."""
    import torch
    import ms_compress_amd as m
    u = bytes(np.random.default_rng(5).choice(np.frombuffer(b"abcd", np.uint8), 50000))
    want_d, want_sa = oracle.oracle_compress(2, u)[1], oracle.oracle_compress_sa(u)[1]
    assert want_d != want_sa
    other = m.Context()
    other.set_lznt1_sa_dict(True)
    try:
        assert m.compress_units(2, [u], ctx=other)[0][0] == want_sa and m.compress_units(2, [u], ctx=gpu_ctx)[0][0] == want_d
        # one plan, created with the flavour on, executed after every switch was flipped the other way
        cap = m.max_compressed_size(2, len(u)) + 2
        plan = m.Plan(other, 2, [0], [len(u)], [0], [cap])
        other.set_lznt1_sa_dict(False)
        other.lib.mscomp_amd_set_lznt1_sa_dict(0)
        d_in = torch.frombuffer(bytearray(u + bytes(16)), dtype=torch.uint8).cuda()
        d_out = torch.zeros(cap + 16, dtype=torch.uint8, device="cuda"); d_len = torch.zeros(1, dtype=torch.int64, device="cuda"); d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
        plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
        assert bytes(d_out[: int(d_len[0])].cpu().numpy()) == want_sa
        plan.close()
        assert m.compress_units(2, [u], ctx=other)[0][0] == want_d                   # a NEW plan of that context follows its new setting
        other.set_lznt1_sa_dict(None)
        other.lib.mscomp_amd_set_lznt1_sa_dict(1)
        try:
            assert m.compress_units(2, [u], ctx=other)[0][0] == want_sa              # no context setting: the process default of the moment
        finally:
            other.lib.mscomp_amd_set_lznt1_sa_dict(0)
    finally:
        other.close()

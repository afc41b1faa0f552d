"""-m gpu: a decompression plan executed repeatedly (the second execution captures the launch sequence as a hipGraph, later ones replay it): large
Xpress streams (segment kernels, 8 rounds, memset nodes), large Xpress+Huffman buffers (token scratch, gather) and the byte stage of
csrc/lzglobal.hip (33 pointer passes) must give the same bytes every time."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_decompression_plan_replays(gpu_ctx, fmt):
    import torch
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    units = [corpus.file_bytes(1, 3_000_000).tobytes(), corpus.file_bytes(3, 1_200_000).tobytes(), corpus.file_bytes(0, 70_000).tobytes(), b"tiny " * 9]
    comp, st = m.compress_units(fmt, units, ctx=gpu_ctx)
    assert all(s == 0 for s in st)
    dev = torch.device("cuda", gpu_ctx.device)
    clen = np.array([len(c) for c in comp], dtype=np.uint64)
    coff, ctot = m.pack_offsets(clen)
    blob = np.zeros(ctot + 16, dtype=np.uint8)
    for o, c in zip(coff, comp):
        blob[int(o): int(o) + len(c)] = np.frombuffer(c, dtype=np.uint8)
    d_in = torch.from_numpy(blob).to(dev)
    caps = np.array([len(u) for u in units], dtype=np.uint64)
    ooff, otot = m.pack_offsets(caps)
    d_len = torch.zeros(len(units), dtype=torch.int64, device=dev)
    d_st = torch.full((len(units),), -9, dtype=torch.int32, device=dev)
    plan = m.Plan(gpu_ctx, fmt, coff, clen, ooff, caps, decompress=True)
    want = [np.frombuffer(u, dtype=np.uint8) for u in units]
    for it in range(5):
        d_out = torch.full((otot + 16,), 0xEE, dtype=torch.uint8, device=dev)     # (a fresh tensor may reuse the address: the graph replays either way)
        plan.execute(d_in, d_out, d_len, d_st)
        torch.cuda.synchronize()
        assert bool((d_st == 0).all().item()), (fmt, it, d_st.cpu().tolist())
        out = d_out.cpu().numpy()
        for o, w in zip(ooff, want):
            assert np.array_equal(out[int(o): int(o) + len(w)], w), (fmt, it)
    plan.close()

"""-m gpu: config-5-sized batches (replicated corpus): many units, multi-GiB scratch; every replica must produce the
bytes of replica 0 (which the full-size tests compare with the oracle)."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(m, ctx, fmt, blob, in_off, in_len):
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


@pytest.mark.parametrize("fmt,reps", [(2, 16), (3, 8), (4, 8)])
def test_replicated_corpus(gpu_ctx, fmt, reps):
    """`reps` copies of the 12-file corpus as one batch (LZNT1/XH: one unit per file; Xpress: 64 KiB units)."""
    import ms_compress_amd as m
    from ms_compress_amd import corpus
    files = [corpus.file_bytes(i) for i in range(12)]
    one = np.concatenate(files)
    flen = np.array([len(f) for f in files], dtype=np.uint64)
    foff = np.zeros(12, dtype=np.uint64); foff[1:] = np.cumsum(flen)[:-1]
    if fmt == 3:
        offs, lens = [], []
        for o, l in zip(foff, flen):
            s = np.arange(0, int(l), 65536, dtype=np.uint64)
            offs.append(s + o); lens.append(np.minimum(65536, int(l) - s).astype(np.uint64))
        uoff, ulen = np.concatenate(offs), np.concatenate(lens)
    else:
        uoff, ulen = foff, flen
    blob = np.tile(one, reps)
    in_off = np.concatenate([uoff + np.uint64(r * len(one)) for r in range(reps)])
    in_len = np.tile(ulen, reps)
    out, out_off, out_len = _run(m, gpu_ctx, fmt, blob, in_off, in_len)
    nu = len(ulen)
    ref = [hashlib.sha256(out[int(out_off[i]): int(out_off[i]) + int(out_len[i])].tobytes()).digest() for i in range(nu)]
    for r in range(1, reps):
        for i in range(nu):
            j = r * nu + i
            assert out_len[j] == out_len[i]
            if i % 97 == 0 or fmt != 3:        # Xpress has 3 239 units per replica: sample them
                assert hashlib.sha256(out[int(out_off[j]): int(out_off[j]) + int(out_len[j])].tobytes()).digest() == ref[i], (r, i)

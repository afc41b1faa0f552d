"""-m gpu: the multi-GPU path of SURVEY.md 8e on ONE GPU -- a job is split with sharding.shard_job exactly as bench.py's ranks split
it, every shard runs in its OWN context (own scratch, own plan, own device buffers: what a rank has), and the concatenation of the
shards' outputs must equal the unsharded batch unit by unit, which must equal what the real reference wrote
(tests/golden/corpus_1mb.json: SHA-256 per file and codec from oracle/_ref)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "corpus_1mb.json")))
KEY = {2: "lznt1", 3: "xpress_units64k", 4: "xpress_huff"}
N = 1_000_000


def _run(m, ctx, fmt, d_blob, in_off, in_len):
    """-> list of the units' compressed bytes"""
    import torch
    caps = np.array([m.max_compressed_size(fmt, int(x)) + 2 for x in in_len], dtype=np.uint64)
    out_off, out_total = m.pack_offsets(caps)
    dev = d_blob.device
    d_out = torch.zeros(out_total + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(max(1, len(in_len)), dtype=torch.int64, device=dev)
    d_st = torch.full((max(1, len(in_len)),), -9, dtype=torch.int32, device=dev)
    with torch.cuda.stream(ctx.stream):
        plan = m.Plan(ctx, fmt, in_off, in_len, out_off, caps)
        plan.execute(d_blob, d_out, d_len, d_st)
        ctx.stream.synchronize()
        plan.close()
    assert bool((d_st[: len(in_len)] == 0).all().item())
    out, lens = d_out.cpu().numpy(), d_len.cpu().numpy()
    return [out[int(o): int(o) + int(l)].tobytes() for o, l in zip(out_off, lens[: len(in_len)])]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_shards_concatenate_to_the_unsharded_job(gpu_ctx, fmt, world):
    import torch
    import ms_compress_amd as m
    from ms_compress_amd import corpus, sharding
    files = [corpus.file_bytes(i, N) for i in range(12)]
    reps = 2
    total = 12 * N
    if fmt == 3:
        uoff = np.concatenate([np.arange(0, N, 65536, dtype=np.uint64) + np.uint64(i * N) for i in range(12)])
        ulen = np.concatenate([np.minimum(65536, N - np.arange(0, N, 65536)).astype(np.uint64) for _ in range(12)])
    else:
        uoff = np.arange(12, dtype=np.uint64) * np.uint64(N); ulen = np.full(12, N, dtype=np.uint64)
    off = np.concatenate([uoff + np.uint64(r * total) for r in range(reps)])
    ln = np.tile(ulen, reps)
    blob = np.tile(np.concatenate(files), reps)
    dev = torch.device("cuda", gpu_ctx.device)
    whole = _run(m, gpu_ctx, fmt, torch.from_numpy(blob).to(dev), off, ln)
    # the job as `world` ranks see it
    parts, covered = [], 0
    for rank in range(world):
        s, e, g0, g1, my_off, my_len = sharding.shard_job(off, ln, world, rank)
        assert s == covered
        covered = e
        ctx = m.Context(device=gpu_ctx.device, stream=torch.cuda.Stream(device=dev))     # a rank's own context on its own stream
        d_shard = torch.from_numpy(np.concatenate([blob[g0:g1], np.zeros(16, np.uint8)])).to(dev)
        parts += _run(m, ctx, fmt, d_shard, my_off, my_len)
        ctx.close()
    assert covered == len(ln) and len(parts) == len(whole)
    assert parts == whole, "a sharded unit differs from the unsharded batch"
    # ... and both are the reference's bytes (per file: one unit, or its 64 KiB units concatenated)
    per_file = len(ulen) // 12
    for r in range(reps):
        for i, name in enumerate(corpus.NAMES):
            k = (r * 12 + i) * per_file
            cat = b"".join(whole[k:k + per_file])
            g = GOLD[name][KEY[fmt]]
            assert len(cat) == g["len"] and hashlib.sha256(cat).hexdigest() == g["sha256"], (name, r)

"""CPU: the N>1 path (per-GPU shard + whole-job reduction) with world_size-2 gloo processes."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def test_shard_ranges_cover_and_balance():
    from ms_compress_amd.sharding import shard_ranges
    from ms_compress_amd import corpus
    lengths = corpus.SIZES * 16                                   # config 5: 16x replicated Silesia, 192 files
    for world in (1, 2, 4, 8):
        r = shard_ranges(lengths, world)
        assert r[0][0] == 0 and r[-1][1] == len(lengths) and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        per = [sum(lengths[s:e]) for s, e in r]
        assert sum(per) == sum(lengths) == 3391017280
        assert max(per) <= 1.25 * sum(lengths) / world
    # 64 KiB units: near-perfect balance
    units = [65536] * 51000 + [1234] * 824
    for world in (2, 8):
        per = [sum(units[s:e]) for s, e in shard_ranges(units, world)]
        assert max(per) - min(per) <= 2 * 65536
    assert shard_ranges([10, 20], 4)[-1][1] == 2                  # fewer units than ranks: empty ranges allowed


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from ms_compress_amd import sharding
    from oracle import loader                                     # stand-in worker on CPU: the checker, never the product
    r, lr, w = sharding.init_distributed("gloo")
    assert (r, w) == (rank, world)
    import random
    rnd = random.Random(3)
    units = [bytes(rnd.getrandbits(8) & 0x3F for _ in range(rnd.randint(0, 9000))) for _ in range(37)]
    s, e = sharding.shard_ranges([len(u) for u in units], w)[r]
    mine = [loader.oracle_compress(2, u)[1] for u in units[s:e]]
    sharding.barrier()
    t, b = sharding.reduce_job(1.0 + rank, sum(len(u) for u in units[s:e]))
    q.put((rank, s, e, t, b, sum(len(m) for m in mine)))
    dist.destroy_process_group()


def test_two_rank_gloo_job():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, e0, t0, b0, c0), (r1, s1, e1, t1, b1, c1) = res
    assert s0 == 0 and e0 == s1 and e1 == 37                      # disjoint cover
    assert t0 == t1 == 2.0                                         # MAX over ranks
    assert b0 == b1 and b0 > 0                                     # SUM over ranks, same on both

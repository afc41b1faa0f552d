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


def test_shard_job_covers_config5_for_every_codec():
    """bench.py's split of BASELINE configs[4]: every unit in exactly one rank's contiguous range, local offsets inside the bytes
    the rank holds, byte totals add up, balance within one unit (64 KiB units) or one file (file units)"""
    from ms_compress_amd import corpus
    from ms_compress_amd.sharding import shard_job
    flen = np.array(corpus.SIZES, np.uint64)
    foff = np.zeros(12, np.uint64); foff[1:] = np.cumsum(flen)[:-1]
    xo = np.concatenate([np.arange(0, int(l), 65536, dtype=np.uint64) + o for o, l in zip(foff, flen)])
    xl = np.concatenate([np.minimum(65536, int(l) - np.arange(0, int(l), 65536)).astype(np.uint64) for l in flen])
    for uoff, ulen in ((foff, flen), (xo, xl)):
        off = np.concatenate([uoff + np.uint64(r * corpus.TOTAL) for r in range(16)])
        ln = np.tile(ulen, 16)
        for world in (1, 2, 3, 4, 8):
            nxt, per = 0, []
            for rank in range(world):
                s, e, g0, g1, mo, ml = shard_job(off, ln, world, rank)
                assert s == nxt and e >= s
                nxt = e
                per.append(int(ml.sum()))
                if e > s:
                    assert int(mo[0]) == 0 and int(mo[-1] + ml[-1]) == g1 - g0 and g0 == int(off[s])
                    assert np.array_equal(mo + np.uint64(g0), off[s:e])
            assert nxt == len(ln) and sum(per) == 3391017280
            assert max(per) - min(per) <= 2 * int(ulen.max())


def test_bench_refuses_an_n_gpu_run_without_n_gpus():
    """python bench.py --gpus 2 where fewer than 2 GPUs are visible: loud failure, never a one-GPU line under that name"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()
    # launched by a launcher with the wrong world size: also refused
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but the launcher started 1 rank" in r.stderr and not r.stdout.strip()


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
    lens = [len(u) for u in units]
    s, e, g0, g1, mo, ml = sharding.shard_job(np.concatenate([[0], np.cumsum(lens)[:-1]]), lens, w, r)
    assert (e - s, int(ml.sum())) == (len(units[s:e]), sum(lens[s:e]))
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


def _fallback_worker(rank, world, port, q, mode):
    """RCCL asked for, RCCL unavailable: `mode` = "raise" patches the group bring-up to raise on EVERY rank, "one" on rank 0 only (rank 1's
    bring-up "succeeds" with a stand-in group) -- either way every rank must end up on gloo, say why, and the reductions must work."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("MSCOMP_AMD_BENCH_BACKEND", None)
    import torch.distributed as dist
    from ms_compress_amd import sharding

    def broken(local_rank, w):
        if mode == "hang" and rank == 1:
            import time
            time.sleep(3600)                                      # RCCL can hang instead of failing: the deadline must catch it
        if mode == "raise" or rank == 0:
            raise RuntimeError("forced: hipIpcGetMemHandle: invalid argument")
        return dist.group.WORLD                                   # stand-in for a healthy RCCL group on the other rank
    os.environ["MSCOMP_AMD_RCCL_TIMEOUT_S"] = "3"
    if mode != "real":
        sharding._rccl_group = broken
    r, lr, w = sharding.init_distributed("nccl")                  # ("real": no GPU here -> the genuine bring-up fails by itself)
    name = sharding.backend_name()
    sharding.barrier()
    t, b = sharding.reduce_job(3.0 - rank, 10 + rank)
    q.put((rank, name, t, b))
    if sharding.abandoned_bringup():
        q.close(); q.join_thread()                                # (the queue's feeder thread must be through before the process is cut off)
        os._exit(0)
    dist.destroy_process_group()


def test_rccl_failure_falls_back_to_gloo_on_every_rank():
    """the first N > 1 run on RCCL is the driver's; a failing bring-up -- on all ranks or on one -- must leave a working
    gloo job whose line names the backend, not a lost scaling curve."""
    import pytest
    for mode in ("raise", "one", "hang", "real"):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q, mode)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=180) for _ in procs)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        for rank, name, t, b in res:
            assert name.startswith("gloo (nccl failed"), (mode, name)
            assert t == 3.0 and b == 21
        if mode == "one":
            assert "here" in res[0][1] and "another rank" in res[1][1]
        if mode == "hang":
            assert "no answer after" in res[1][1]


def test_backend_env_switch(monkeypatch):
    """MSCOMP_AMD_BENCH_BACKEND=gloo: no RCCL bring-up is attempted at all (one process: nothing is initialised, the name stays None)"""
    from ms_compress_amd import sharding
    monkeypatch.setenv("MSCOMP_AMD_BENCH_BACKEND", "gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert sharding.init_distributed() == (0, 0, 1) and sharding.backend_name() is None
    assert sharding.reduce_job(1.5, 7) == (1.5, 7)

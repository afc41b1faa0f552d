"""Per-GPU sharding of a batch of independent units: one process per GPU, contiguous unit ranges balanced
by input bytes, NO data-path collective (chunks/files are independent; xGMI/RCCL carry only the timing reduction).

The reference has no counterpart (single-threaded library); this is the "plain per-GPU shard of the input batch" that
BASELINE.json's north_star asks for.
"""
import os

import numpy as np


def shard_ranges(lengths, world):
    """Split units (by their byte lengths) into `world` contiguous ranges with near-equal byte totals.
    Returns [(start, end)] * world; every unit is in exactly one range; ranges may be empty when units < world."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    cum = np.concatenate([[0], np.cumsum(lengths)])
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        # choose the boundary closest to the target, keep cuts monotone
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, n)] - target):
            k -= 1
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_job(unit_off, unit_len, world, rank):
    """This rank's share of a job whose units lie back to back (ascending offsets) in one global byte range: units [s, e), the
    global bytes [g0, g1) it has to hold in HBM, and the units' offsets relative to g0. An empty share gives s == e, g0 == g1."""
    unit_off = np.asarray(unit_off, dtype=np.uint64)
    unit_len = np.asarray(unit_len, dtype=np.uint64)
    s, e = shard_ranges(unit_len, world)[rank]
    if e == s:
        return s, e, 0, 0, unit_off[s:e], unit_len[s:e]
    g0 = int(unit_off[s])
    g1 = int(unit_off[e - 1] + unit_len[e - 1])
    return s, e, g0, g1, unit_off[s:e] - np.uint64(g0), unit_len[s:e]


def dist_env():
    """(rank, local_rank, world) from the torch.distributed.run environment (1 process when absent)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1 ('nccl' = RCCL on ROCm; 'gloo' for the CPU tests)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_job(elapsed_s, units_bytes, device=None):
    """Whole-job figures from per-rank ones: time = MAX over ranks, bytes = SUM over ranks (the only collectives used)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), int(units_bytes)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    b = torch.tensor([int(units_bytes)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return float(t.item()), int(b.item())

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


_GROUP = None          # the RCCL group the timing collectives run on when it came up on every rank (None: the gloo world group, host tensors)
_BACKEND = None        # "nccl" | "gloo" | None (one process)
_NOTE = None           # why gloo is in use although RCCL was asked for
_ABANDONED = False     # an RCCL bring-up thread did not come back by its deadline and was left behind


def abandoned_bringup():
    """True when some rank of the job left a hung RCCL bring-up thread behind: every rank then ends with barrier + os._exit instead of tearing the
    process group down (a rank that destroys it while another still sits in a collective resets that rank's connection)."""
    return _ABANDONED


def backend_name():
    """'nccl' (= RCCL on ROCm), 'gloo', 'gloo (nccl failed: ...)' or None for a one-process run -- goes into the bench line's config."""
    if _BACKEND == "gloo" and _NOTE:
        return "gloo (%s)" % _NOTE
    return _BACKEND


def _rccl_group(local_rank, world):
    """A process group on RCCL over all ranks, proven by one all-reduce on this rank's GPU. Raises when anything in that fails."""
    import datetime
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180))
    x = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local_rank))
    dist.all_reduce(x, group=g)
    torch.cuda.synchronize()
    if int(x.item()) != world:
        raise RuntimeError("RCCL all-reduce over %d ranks gave %d" % (world, int(x.item())))
    return g


def init_distributed(backend=None, device=None):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1. The data path has no collective; what runs over the
    process group is the barrier around the timed region and the MAX / SUM reduction of its figures. The WORLD group is always gloo (host
    tensors, 127.0.0.1 / loopback: it cannot fail for GPU reasons) and is the channel on which the ranks AGREE; on GPUs an RCCL group
    ('nccl' = RCCL on ROCm) is created beside it and proven with one all-reduce -- if that fails on ANY rank (or `backend` /
    MSCOMP_AMD_BENCH_BACKEND says "gloo") every rank stays on gloo and `backend_name()` says why. The first real N > 1 run on N GPUs is the
    driver's: a failing RCCL bring-up must cost the line's "backend" key, not the whole scaling curve.
    device: this rank's GPU ordinal (default: LOCAL_RANK)."""
    global _GROUP, _BACKEND, _NOTE
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        forced = backend or os.environ.get("MSCOMP_AMD_BENCH_BACKEND") or None
        want_rccl = forced == "nccl" or (forced is None and torch.cuda.is_available())
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")          # one node: the container's hostname may not resolve
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        _BACKEND, _GROUP, _NOTE = "gloo", None, None
        if want_rccl:
            # the bring-up runs in a thread of its own with a deadline: RCCL can also HANG instead of failing (seen here: two ranks on one GPU sit
            # in communicator creation forever) -- a rank that is not through after MSCOMP_AMD_RCCL_TIMEOUT_S seconds votes "no" and the job goes on
            # over gloo; the abandoned thread is a daemon, bench.py leaves with os._exit when sharding.abandoned_bringup() says one is left behind
            import threading
            global _ABANDONED
            ok, why, g, box = 1, "", None, []

            def bring_up():
                try:
                    box.append(_rccl_group(local_rank if device is None else int(device), world))
                except Exception as ex:                            # noqa: BLE001 -- anything: a missing backend, IPC handles, a dead link
                    box.append(ex)
            th = threading.Thread(target=bring_up, name="rccl-bring-up", daemon=True)
            th.start()
            th.join(float(os.environ.get("MSCOMP_AMD_RCCL_TIMEOUT_S", "150")))
            if th.is_alive():
                ok, why, _ABANDONED = 0, "no answer after %s s" % os.environ.get("MSCOMP_AMD_RCCL_TIMEOUT_S", "150"), True
            elif isinstance(box[0], Exception):
                ex = box[0]
                ok, why = 0, "%s: %s" % (type(ex).__name__, str(ex).splitlines()[0][:120] if str(ex) else "")
            else:
                g = box[0]
            flag = torch.tensor([ok, 0 if _ABANDONED else 1], dtype=torch.int64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)            # gloo: every rank learns whether EVERY rank has RCCL -- and whether any rank left a hung thread behind
            _ABANDONED = int(flag[1].item()) == 0                  # (then ALL ranks leave the same way: barrier, os._exit)
            if int(flag[0].item()) == 1:
                _GROUP, _BACKEND = g, "nccl"
            else:
                _NOTE = "nccl failed" + (" here, " + why if not ok else " on another rank")
                if rank == 0 or not ok:
                    import sys
                    print("ms_compress_amd.sharding: rank %d: RCCL did not come up (%s): timing collectives over gloo" % (rank, why or "another rank"), file=sys.stderr)
    return rank, local_rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if _GROUP is not None:
            import torch
            dist.barrier(group=_GROUP, device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def reduce_job(elapsed_s, units_bytes, device=None):
    """Whole-job figures from per-rank ones: time = MAX over ranks, bytes = SUM over ranks (the only collectives used). On RCCL the two
    scalars live on this rank's GPU, on gloo on the host (`device` is kept for callers of the earlier signature and ignored on gloo)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), int(units_bytes)
    dev = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if _GROUP is not None else None
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=dev)
    b = torch.tensor([int(units_bytes)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_GROUP)
    dist.all_reduce(b, op=dist.ReduceOp.SUM, group=_GROUP)
    return float(t.item()), int(b.item())

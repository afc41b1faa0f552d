#!/usr/bin/env python3
"""bench.py -- input MB/s of the MI355X-native MS-XCA compress path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is ONE pass of the hot path over one batch of synthetic input that is already resident in HBM:
  headline workload (BASELINE.json configs[1]): LZNT1, Silesia 'mozilla' (51 220 480 B) as a batch of 12 505 4-KiB chunks.
With N ranks every rank compresses its own copy of the batch (weak scaling, chunks/files are independent: no data-path
collective); value = bytes all ranks compressed / max-over-ranks time. Rank 0 prints ONE JSON line with
`roofline` (dominant kernel: algorithmic bytes per launch / HIP-event kernel time vs the 8 TB/s HBM peak) and
`cpu_baseline` (the reference's own CPU encoder, oracle/_ref, on this host's cores; a bounded sample).
The other codecs (BASELINE configs 3 and 4: Xpress 64 KiB units, Xpress+Huffman file mode, full 212 MB corpus) are timed
after the headline region and reported under "extra" (N=1 only, or with --all).
"""
import argparse
import json
import re
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
DOMINANT = {2: "lznt1_chunk_kernel", 3: "xp_find_kernel", 4: "xp_find_kernel"}


_FILES = {}


def _file(corpus, i):
    if i not in _FILES:
        _FILES[i] = corpus.file_bytes(i)
    return _FILES[i]


def build_workload(m, corpus, fmt, which):
    """-> (blob uint8, in_off, in_len, description). Units are independent ms_compress() inputs."""
    if which == "mozilla":                         # configs[1]: one unit, 12 505 chunks of 4 KiB inside
        data = _file(corpus, 1)
        return data, np.zeros(1, np.uint64), np.array([len(data)], np.uint64), "mozilla 51220480 B as one unit"
    files = [_file(corpus, i) for i in range(12)]
    if "blob" not in _FILES:
        _FILES["blob"] = np.concatenate(files)
    blob = _FILES["blob"]
    flen = np.array([len(f) for f in files], np.uint64)
    foff = np.zeros(12, np.uint64); foff[1:] = np.cumsum(flen)[:-1]
    if which == "silesia_files":                   # configs[3]: XH file mode (also LZNT1 per file)
        return blob, foff, flen, "12 Silesia-shaped files, 211938580 B, one unit per file"
    offs, lens = [], []                            # configs[2]: every file cut into independent 64 KiB units
    for o, l in zip(foff, flen):
        s = np.arange(0, int(l), 65536, dtype=np.uint64)
        offs.append(s + o); lens.append(np.minimum(65536, int(l) - s).astype(np.uint64))
    return blob, np.concatenate(offs), np.concatenate(lens), "12 files cut into 3239 independent 64 KiB units, 211938580 B"


class Job:
    def __init__(self, m, ctx, fmt, blob, in_off, in_len):
        import torch
        self.m, self.ctx, self.fmt = m, ctx, fmt
        self.in_bytes = int(in_len.sum())
        caps = [m.max_compressed_size(fmt, int(x)) + 2 for x in in_len]
        out_off, out_total = m.pack_offsets(caps)
        dev = torch.device("cuda", ctx.device)
        self.d_in = torch.from_numpy(blob).to(dev)
        self.d_out = torch.empty(out_total + 16, dtype=torch.uint8, device=dev)
        self.d_len = torch.zeros(len(in_len), dtype=torch.int64, device=dev)
        self.d_st = torch.zeros(len(in_len), dtype=torch.int32, device=dev)
        self.plan = m.Plan(ctx, fmt, in_off, in_len, out_off, caps)

    def step(self):
        self.plan.execute(self.d_in, self.d_out, self.d_len, self.d_st)

    def out_bytes(self):
        assert bool((self.d_st == 0).all().item()), "a unit reported an error status"
        return int(self.d_len.sum().item())

    def close(self):
        self.plan.close()


def decompress_leg(m, ctx, fmt, blob, in_off, in_len, desc, steps, sharding):
    """decode on the GPU what the GPU compressor wrote for this workload, check that the input comes back,
    report decompressed MB/s (HBM-resident, like `value`)."""
    import torch
    job = Job(m, ctx, fmt, blob, in_off, in_len)
    job.step(); torch.cuda.synchronize()
    comp_len = job.d_len.cpu().numpy().astype(np.uint64)
    assert bool((job.d_st == 0).all().item())
    d_back = torch.zeros(len(blob) + 16, dtype=torch.uint8, device=job.d_in.device)
    caps = [m.max_compressed_size(fmt, int(x)) + 2 for x in in_len]
    comp_off, _ = m.pack_offsets(caps)
    plan = m.Plan(ctx, fmt, comp_off, comp_len, in_off, in_len, decompress=True)

    class D:
        pass
    d = D(); d.ctx = ctx
    d.step = lambda: plan.execute(job.d_out, d_back, job.d_len2, job.d_st2)
    job.d_len2 = torch.zeros_like(job.d_len); job.d_st2 = torch.full_like(job.d_st, -9)
    dt, prof = timed(d, steps, 1, sharding)
    ok = bool((job.d_st2 == 0).all().item()) and bool(torch.equal(job.d_len2.cpu(), torch.from_numpy(in_len.astype(np.int64)))) \
        and bool(torch.equal(d_back[: len(blob)], job.d_in[: len(blob)]))
    res = {"MB_per_s": round(job.in_bytes * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 3), "workload": desc,
           "round_trip_ok": ok, "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}
    plan.close(); job.close()
    return res


def timed(job, steps, warmup, sharding):
    import torch
    for _ in range(warmup):
        job.step()
    torch.cuda.synchronize()
    job.ctx.profile_read()                         # drop warm-up records
    job.ctx.profile_enable(True)                   # HIP events around every kernel, on the stream they are launched on
    sharding.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    torch.cuda.synchronize(); sharding.barrier()
    dt = time.perf_counter() - t0
    prof = job.ctx.profile_read()
    job.ctx.profile_enable(False)
    return dt, prof


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r01_pmc_traffic.json: separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench workload, FETCH_SIZE doubled per the gfx950 note of the microarch
    guide). Collected offline -- a live bench run cannot read PMCs -- so it is attached only when the file is present."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        for k, v in doc["kernels"].items():
            # (the library times kernel FAMILIES: lznt1_chunk_kernel covers lznt1_chunk4_kernel, the four-wave variant)
            if re.sub(r"\d+_kernel$", "_kernel", k.split("<")[0]).endswith(kernel):
                return v["hbm_bytes_per_launch_corrected"]
    except Exception:
        pass
    return None


def roofline(fmt, prof, in_bytes, out_bytes, steps):
    name = DOMINANT[fmt]
    tot_ms = sum(v[0] for v in prof.values())
    dom = max(prof.items(), key=lambda kv: kv[1][0])[0] if prof else name
    ms, cnt = prof.get(dom, (0.0, 0))
    per_launch_ms = ms / cnt if cnt else float("nan")
    launches_per_step = cnt / steps if cnt else 1          # 2 when the batch runs as two halves on two streams
    # algorithmic bytes: 1 B HBM read + CR B HBM write per input byte, for the units one launch processes
    alg = (in_bytes + out_bytes) / launches_per_step
    ach = alg / (per_launch_ms * 1e-3) / 1e9 if cnt else float("nan")
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(dom),
            "kernel_ms_per_launch": round(per_launch_ms, 4), "launches_per_step": launches_per_step,
            "kernel_share_of_gpu_time": round(ms / tot_ms, 3) if tot_ms else None,
            "algorithmic_bytes_per_launch": int(alg),
            "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}


def _cpu_timed(fn, fmt, units, caps, cores, budget_s):
    """passes and seconds of fn over the units on `cores` C threads (oracle.loader.time_units: no interpreter between the calls)"""
    from oracle import loader
    dt1, st, _ = loader.time_units(fn, fmt, units, caps, cores, 1)
    assert bool((st == 0).all()), "the CPU baseline reported an error status"
    more = max(0, min(19, int(budget_s / max(dt1, 1e-3)) - 1))
    dt = dt1 + (loader.time_units(fn, fmt, units, caps, cores, more)[0] if more else 0.0)
    return 1 + more, dt


def cpu_baseline(fmt, blob, budget_s=12.0):
    """The reference's own CPU encoder (oracle/_ref, compiled from /root/reference) -- or our C port when that file did not
    travel -- on this host's cores, over a bounded sample of the SAME workload (prefix of the batch, split on 64 KiB / 4 KiB
    aligned boundaries so that every thread does independent ms_compress calls). Reported baseline, not the target."""
    from oracle import loader
    ref = loader.load_ref()
    kind = "reference" if ref is not None else "port"
    cores = max(1, min(os.cpu_count() or 1, 64))
    per = 4 << 20                                  # 4 MiB per thread: ~0.1-0.15 s of single-core work per pass
    sample = min(len(blob), per * cores) // 65536 * 65536
    data = blob[:sample].tobytes()
    piece = max(65536, sample // cores // 65536 * 65536)
    slices = [data[o:o + piece] for o in range(0, sample, piece)]
    caps = [loader.load_oracle().orc_max_compressed_size(fmt, len(x)) + 2 for x in slices]
    passes, dt = _cpu_timed(ref.ms_compress if ref is not None else None, fmt, slices, caps, len(slices), budget_s)
    return {"value": round(sample * passes / dt / 1e6, 1), "unit": "MB/s", "cores": len(slices), "kind": kind,
            "sample": "%d passes over the first %d B of the batch, %d threads x %d B independent ms_compress calls" % (passes, sample, len(slices), piece)}


def cpu_decompress_baseline(fmt, blob, budget_s=3.0):
    """The reference's CPU decoder beside the GPU decompression leg: the same kind of bounded sample as cpu_baseline (64 KiB units
    for the Xpress formats, 4 MiB pieces for LZNT1; compressed by the reference, untimed), all host cores, output MB/s."""
    from oracle import loader
    ref = loader.load_ref()
    kind = "reference" if ref is not None else "port"
    cores = max(1, min(os.cpu_count() or 1, 64))
    unit = (4 << 20) if fmt == 2 else 65536
    per_thread = 4 << 20
    sample = min(len(blob), per_thread * cores) // unit * unit
    data = blob[:sample].tobytes()
    comp = loader.ref_compress if ref is not None else loader.oracle_compress
    units = [data[o:o + unit] for o in range(0, sample, unit)]
    streams = [comp(fmt, u)[1] for u in units]
    threads = min(cores, len(units))
    fn = ref.ms_decompress if ref is not None else loader.load_oracle().orc_decompress
    passes, dt = _cpu_timed(fn, fmt, streams, [len(u) for u in units], threads, budget_s)
    return {"value": round(sample * passes / dt / 1e6, 1), "unit": "MB/s (output)", "cores": threads, "kind": kind,
            "sample": "%d passes over the first %d B of the batch as %d independent ms_decompress calls of %d B" % (passes, sample, len(units), unit)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--codec", default="lznt1", choices=["lznt1", "xpress", "xpress_huff"])
    ap.add_argument("--all", action="store_true", help="also time the other codecs (default at N=1)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--replicas", type=int, default=1, help="tile the workload R times per GPU (BASELINE config 5 uses 16x Silesia over 8 GPUs)")
    args = ap.parse_args()

    import torch
    import ms_compress_amd as m
    from ms_compress_amd import corpus, sharding
    rank, local_rank, world = sharding.init_distributed()
    assert world == max(1, args.gpus) or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    ctx = m.Context(device=local_rank)
    fmt = m.FORMATS[args.codec]
    which = {"lznt1": "mozilla", "xpress": "silesia_units64k", "xpress_huff": "silesia_files"}[args.codec]
    blob, in_off, in_len, desc = build_workload(m, corpus, fmt, which)
    if args.replicas > 1:
        R = args.replicas
        in_off = np.concatenate([in_off + np.uint64(r * len(blob)) for r in range(R)])
        in_len = np.tile(in_len, R)
        blob = np.tile(blob, R)
        desc += " x%d replicas" % R
    job = Job(m, ctx, fmt, blob, in_off, in_len)
    dt, prof = timed(job, args.steps, args.warmup, sharding)
    out_bytes = job.out_bytes()
    dev = torch.device("cuda", local_rank)
    job_dt, job_bytes = sharding.reduce_job(dt, job.in_bytes * args.steps, device=dev)
    res = {
        "metric": "input MB/s (%s compress, bit-exact with the reference CPU encoder)" % args.codec,
        "value": round(job_bytes / job_dt / 1e6, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(job_dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic" if corpus.source() == "synthetic" else corpus.source(),
        "config": {"workload": "%s: %s; one %s block per chunk" % (args.codec, desc, {2: "4 KiB", 3: "64 KiB", 4: "64 KiB"}[fmt]),
                   "bytes_per_step_per_gpu": job.in_bytes, "units_per_gpu": int(len(in_len)), "compression_ratio": round(out_bytes / job.in_bytes, 4),
                   "parallelism": "replica-per-gpu x%d (independent units, no collective)" % world, "MiB_per_s": round(job_bytes / job_dt / 2 ** 20, 1)},
        "roofline": roofline(fmt, prof, job.in_bytes, out_bytes, args.steps),
    }
    job.close()
    if rank == 0 and world == 1 and not args.no_cpu:
        res["cpu_baseline"] = cpu_baseline(fmt, blob)
    if world == 1 or args.all:
        extra = {}
        for codec, wl in (("lznt1", "mozilla"), ("xpress", "silesia_units64k"), ("xpress_huff", "silesia_files")):
            if codec == args.codec:
                continue
            f2 = m.FORMATS[codec]
            b2, o2, l2, d2 = build_workload(m, corpus, f2, wl)
            j2 = Job(m, ctx, f2, b2, o2, l2)
            t2, p2 = timed(j2, max(3, args.steps // 4), 1, sharding)
            ob = j2.out_bytes()
            steps2 = max(3, args.steps // 4)
            extra[codec] = {"MB_per_s": round(j2.in_bytes * steps2 / t2 / 1e6, 1), "ms_per_step": round(t2 / steps2 * 1e3, 3),
                            "workload": d2, "compression_ratio": round(ob / j2.in_bytes, 4), "roofline": roofline(f2, p2, j2.in_bytes, ob, steps2)}
            j2.close()
        dec = {}
        for codec, wl in (("lznt1", "mozilla"), ("xpress", "silesia_units64k"), ("xpress_huff", "silesia_units64k")):
            f2 = m.FORMATS[codec]
            b2, o2, l2, d2 = build_workload(m, corpus, f2, wl)
            dec[codec] = decompress_leg(m, ctx, f2, b2, o2, l2, d2, 3, sharding)
            if rank == 0 and not args.no_cpu:
                dec[codec]["cpu_baseline"] = cpu_decompress_baseline(f2, b2)
        extra["decompress"] = dec
        res["extra"] = extra
    if rank == 0:
        print(json.dumps(res))
    ctx.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

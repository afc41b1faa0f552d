#!/usr/bin/env python3
"""bench.py -- input MB/s of the MI355X-native MS-XCA compress path (BASELINE.json metric), 1 to N GPUs.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script starts its N ranks itself (torch.distributed.run, one
process per GPU, rendezvous on 127.0.0.1); launched BY torch.distributed.run it is one of the ranks. A world size other
than N is an error -- there is no silent one-GPU run of an N-GPU request.

Workload (every N, so that the per-N values form ONE curve): BASELINE.json configs[4] -- the 12-file Silesia(-shaped) set
replicated 16x (192 files, 3 391 017 280 B), cut into independent units, split over the ranks with
`sharding.shard_ranges` (contiguous ranges balanced by bytes, no data-path collective: SURVEY.md 8e) -- STRONG scaling:
the job is fixed, a rank holds 1/N of it in HBM. A "step" is ONE pass of the hot path over the rank's shard, inputs
resident in HBM. value = bytes of the whole job x K / max-over-ranks time of the K steps (barrier + synchronize on
both sides). After the timed region every rank checks what its first replica of the 12 files compressed to against the
digests of the REAL reference (tests/golden/corpus_full.json) -- `parity_checked` -- and refuses to print a number otherwise. The headline codec is LZNT1 (one unit per file, 4 KiB chunks inside: 827 936 chunks); Xpress (51 824
independent 64 KiB units) and Xpress+Huffman (one unit per file, 64 KiB chunks with the previous chunk as window) run on
the same batch right after and are reported under extra.config5, reduced over the ranks the same way.

Rank 0 prints ONE SHORT JSON line (< 4 KB, scalars only; `short_line` below): the contract's keys, `config` (the workload and, per codec, MB/s,
ms per step, roofline fraction, parity verdict, CPU figures, the one-rank-of-8 ratio as flat keys), `roofline` (dominant kernel: algorithmic bytes
per launch / HIP-event kernel time vs the 8 TB/s HBM peak; `traffic` from the committed PMC passes of the same command) and, at N = 1, `cpu_baseline`
(the reference's own CPU encoder, oracle/_ref, on this host's cores: the leg's unit list in one pass, the load-balanced figure, one thread). The WHOLE
document -- BASELINE configs[1..3] one by one with their host-pointer end-to-end figures, what one rank of an 8-GPU run holds, the suffix-array
flavour, the decompression legs, every kernel's time and the counter-derived `secondary` block of every roofline -- is written to
bench_extra.json beside this script (and to gpurun_out/ when that exists); `--full` prints it instead of the short line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
REPLICAS = 16                # BASELINE.json configs[4]
PROFILE_TAG = "r06"              # the committed profiles the PMC / SQ figures are looked up in; a missing file falls back to the round before and says so
# the kernel SYMBOL behind the library's timer name of each codec's dominant kernel (template instance included, so that the
# PMC figures of the Xpress and the Xpress+Huffman match finder are never mixed up)
SYMBOLS = {
    (2, "lznt1_chunk_kernel"): "msc::lznt1_chunk4_kernel<false>",
    (3, "xp_find_kernel"): "msc::xp_find_kernel<8192u, 8192u, 512u, 4096u>",
    (4, "xp_find_kernel"): "msc::xp_find_kernel<65536u, 0u, 1024u, 8192u>",
    (3, "xp_lazy2_kernel"): "msc::xp_lazy2_kernel<16384u, 32u, 4u>",
}


CODEC_OF = {2: "lznt1", 3: "xpress", 4: "xpress_huff"}


# ---------------------------------------------------------------- launch ----------------------------------------------------------------
def spawn_ranks(n):
    """python bench.py --gpus N without torch.distributed.run around it: become the launcher."""
    import torch
    have = torch.cuda.device_count()
    if have < n and "--oversubscribe" not in sys.argv:
        sys.exit("bench.py: --gpus %d asked for, %d visible: refusing to run a smaller job under that name" % (n, have))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


# ---------------------------------------------------------------- workloads ----------------------------------------------------------------
class Corpus:
    """The 12 files on the host (generated once) and on this rank's device."""

    def __init__(self, corpus, dev):
        import torch
        self.files = [corpus.file_bytes(i) for i in range(12)]
        self.flen = np.array([len(f) for f in self.files], np.uint64)
        self.foff = np.zeros(12, np.uint64); self.foff[1:] = np.cumsum(self.flen)[:-1]
        self.total = int(self.flen.sum())
        self.d_files = [torch.from_numpy(f).to(dev) for f in self.files]
        self.dev = dev
        self._blob = None

    def blob(self):
        if self._blob is None:
            self._blob = np.concatenate(self.files)
        return self._blob

    def device_range(self, g0, g1):
        """bytes [g0, g1) of the 16x replicated concatenation, assembled on the device from the 12 resident files"""
        import torch
        parts, pos = [], g0
        while pos < g1:
            r, o = divmod(pos, self.total)
            f = int(np.searchsorted(self.foff, o, side="right")) - 1
            a = o - int(self.foff[f])
            n = min(int(self.flen[f]) - a, g1 - pos)
            parts.append(self.d_files[f][a:a + n])
            pos += n
        parts.append(torch.zeros(16, dtype=torch.uint8, device=self.dev))
        return torch.cat(parts)


def config5_units(cor, fmt):
    """(offset, length) of every unit of the whole job in the replicated concatenation, and a description"""
    if fmt == 3:                                   # Xpress: every file cut into independent 64 KiB units (the reference has no chunking of its own)
        offs, lens = [], []
        for o, l in zip(cor.foff, cor.flen):
            s = np.arange(0, int(l), 65536, dtype=np.uint64)
            offs.append(s + o); lens.append(np.minimum(65536, int(l) - s).astype(np.uint64))
        uoff, ulen = np.concatenate(offs), np.concatenate(lens)
        what = "independent 64 KiB units"
    else:                                          # LZNT1 / Xpress+Huffman: one ms_compress call per file, the codec chunks it
        uoff, ulen = cor.foff, cor.flen
        what = "files, one unit each (%s chunks inside)" % ("4 KiB" if fmt == 2 else "64 KiB")
    off = np.concatenate([uoff + np.uint64(r * cor.total) for r in range(REPLICAS)])
    ln = np.tile(ulen, REPLICAS)
    return off, ln, "%dx replicated Silesia-shaped set: %d %s, %d B" % (REPLICAS, len(ln), what, int(ln.sum()))


def single_gpu_workload(cor, which):
    """BASELINE configs[1..3] as round 1 measured them -> (host blob, in_off, in_len, description)"""
    if which == "mozilla":                         # configs[1]: one unit, 12 505 chunks of 4 KiB inside
        data = cor.files[1]
        return data, np.zeros(1, np.uint64), np.array([len(data)], np.uint64), "mozilla 51220480 B as one unit"
    if which == "silesia_files":                   # configs[3]: XH file mode
        return cor.blob(), cor.foff, cor.flen, "12 Silesia-shaped files, 211938580 B, one unit per file"
    offs, lens = [], []                            # configs[2]: every file cut into independent 64 KiB units
    for o, l in zip(cor.foff, cor.flen):
        s = np.arange(0, int(l), 65536, dtype=np.uint64)
        offs.append(s + o); lens.append(np.minimum(65536, int(l) - s).astype(np.uint64))
    return cor.blob(), np.concatenate(offs), np.concatenate(lens), "12 files cut into 3239 independent 64 KiB units, 211938580 B"


class Job:
    """one batch resident in HBM + its plan"""

    def __init__(self, m, ctx, fmt, d_in, in_off, in_len):
        import torch
        self.m, self.ctx, self.fmt = m, ctx, fmt
        self.in_bytes = int(in_len.sum())
        caps = np.array([m.max_compressed_size(fmt, int(x)) + 2 for x in in_len], np.uint64)
        out_off, out_total = m.pack_offsets(caps)
        dev = torch.device("cuda", ctx.device)
        self.d_in = d_in if hasattr(d_in, "data_ptr") else torch.from_numpy(d_in).to(dev)
        self.d_out = torch.empty(out_total + 16, dtype=torch.uint8, device=dev)
        self.d_len = torch.zeros(max(1, len(in_len)), dtype=torch.int64, device=dev)
        self.d_st = torch.zeros(max(1, len(in_len)), dtype=torch.int32, device=dev)
        self.caps, self.out_off, self.n = caps, out_off, len(in_len)
        self.plan = m.Plan(ctx, fmt, in_off, in_len, out_off, caps)

    def step(self):
        self.plan.execute(self.d_in, self.d_out, self.d_len, self.d_st)

    def out_bytes(self):
        assert bool((self.d_st[: self.n] == 0).all().item()), "a unit reported an error status"
        return int(self.d_len[: self.n].sum().item())

    def close(self):
        self.plan.close()
        self.d_in = self.d_out = None


GOLD_KEY = {2: "lznt1", 3: "xpress_units64k", 4: "xpress_huff"}


def _host_reference_digests(fmt, cor):
    """(length, sha256) of what the reference's CPU encoder (oracle/_ref; our C restatement when that file did not travel) writes for each of
    the 12 files as this job cuts them -- one ms_compress call per file, for Xpress one per 64 KiB unit with the streams concatenated. Used by
    the parity gate when the data is NOT the synthetic corpus (no committed digests exist for it). Host threads, outside the timed region."""
    import hashlib
    from oracle import loader
    ref = loader.load_ref()
    fn = ref.ms_compress if ref is not None else None
    blob = cor.blob()
    if fmt == 3:
        offs, lens = [], []
        for o, l in zip(cor.foff, cor.flen):
            st = np.arange(0, int(l), 65536, dtype=np.uint64)
            offs.append(st + o); lens.append(np.minimum(65536, int(l) - st).astype(np.uint64))
        uoff, ulen = np.concatenate(offs), np.concatenate(lens)
    else:
        uoff, ulen = cor.foff, cor.flen
    caps = np.array([loader.load_oracle().orc_max_compressed_size(fmt, int(x)) + 2 for x in ulen], np.uint64)
    order = np.argsort(-ulen.astype(np.int64), kind="stable")                      # longest first: the makespan is the largest file's
    _, st, ln, out, ooff = loader.time_units_ex(fn, fmt, blob, uoff[order], ulen[order], caps[order], max(1, min(os.cpu_count() or 1, 256)), 1, keep_output=True)
    assert bool((st == 0).all()), "the host reference reported an error status"
    where = np.empty(len(order), np.int64); where[order] = np.arange(len(order))
    res, u = [], 0
    for l in cor.flen:
        k = 1 if fmt != 3 else (int(l) + 65535) // 65536
        h, total = hashlib.sha256(), 0
        for i in range(u, u + k):
            j = int(where[i]); a = int(ooff[j]); n = int(ln[j])
            h.update(out[a:a + n].tobytes()); total += n
        res.append((total, h.hexdigest())); u += k
    return res


_HOST_DIGESTS = {}


def parity_gate(m, job, fmt, cor, first_unit):
    """The in-run parity gate: SHA-256 of what this rank's FIRST replica of the 12 files compressed to, file by file, against
    the reference's. Synthetic corpus: the digests the real reference gave (tests/golden/corpus_full.json, written by tools/make_golden_full.py
    from oracle/_ref). Any other data (SILESIA_DIR): the reference's encoder is run on the host, here, on the same files (oracle/_ref travels
    with the repository). Outside the timed region. True / False, or None when the shard does not start at a replica."""
    import hashlib
    import torch
    from ms_compress_amd import corpus
    per_file = [1 if fmt != 3 else (int(l) + 65535) // 65536 for l in cor.flen]
    nu = sum(per_file)
    if first_unit % nu != 0 or job.n < nu:
        return None
    if corpus.source() == "synthetic":
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "corpus_full.json")))
        want = [(gold[name][GOLD_KEY[fmt]]["len"], gold[name][GOLD_KEY[fmt]]["sha256"]) for name in corpus.NAMES]
    else:
        if fmt not in _HOST_DIGESTS:
            _HOST_DIGESTS[fmt] = _host_reference_digests(fmt, cor)
        want = _HOST_DIGESTS[fmt]
    d_packed, d_poff = m.compact_batch(job.ctx, job.out_off[:nu], job.caps[:nu], job.d_out, job.d_len)
    torch.cuda.synchronize()
    poff = d_poff.cpu().numpy()
    first = d_packed[: int(poff[nu])].cpu().numpy()
    u, ok = 0, True
    for i, name in enumerate(corpus.NAMES):
        a, b = int(poff[u]), int(poff[u + per_file[i]])
        ok = ok and (b - a == want[i][0]) and hashlib.sha256(first[a:b].tobytes()).hexdigest() == want[i][1]
        u += per_file[i]
    return bool(ok)


def timed(job, steps, warmup, sharding):
    """EXACTLY `steps` steps between barrier + synchronize on both sides; HIP events around every kernel on its launch stream"""
    import torch
    for _ in range(warmup):
        job.step()
    torch.cuda.synchronize()
    job.ctx.profile_read()                         # drop warm-up records
    job.ctx.profile_enable(True)
    sharding.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    torch.cuda.synchronize(); sharding.barrier()
    dt = time.perf_counter() - t0
    prof = job.ctx.profile_read()
    job.ctx.profile_enable(False)
    return dt, prof


# ---------------------------------------------------------------- roofline ----------------------------------------------------------------
def _profile_doc(name):
    """(document, tag it came from): the current round's committed profile, else the round before's (and the line says which)"""
    for tag in (PROFILE_TAG, "r05"):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name)))), tag
        except Exception:
            continue
    return None, None


def pmc_traffic(fmt, timer_name, workload_key):
    """HBM bytes per launch of the kernel behind `timer_name` for this codec, from the committed PMC passes
    (profiles/<tag>_pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench command, FETCH_SIZE
    doubled per the gfx950 note of the microarch guide). Collected offline -- a live run cannot read PMCs -- and keyed by the exact
    kernel symbol (template arguments included) AND the workload, so it is attached only to the launch it was measured on."""
    doc, tag = _profile_doc("pmc_traffic")
    sym = SYMBOLS.get((fmt, timer_name))
    if not doc or not sym:
        return None, None
    rec = doc.get("by_codec", {}).get(workload_key, {}).get(CODEC_OF[fmt], {}).get(sym)
    return (rec["hbm_bytes_per_launch_corrected"], tag) if rec else (None, None)


def secondary_bound(fmt, timer_name, workload_key):
    """What actually limits the kernel (the HBM fraction of these latency / issue-bound integer kernels says little): busiest
    pipe and wait shares from the committed SQ-counter passes (profiles/<tag>_sq_counters.json)."""
    doc, tag = _profile_doc("sq_counters")
    sym = SYMBOLS.get((fmt, timer_name))
    if not doc or not sym:
        return None
    probes, ptag = _profile_doc("probes")
    for wl in (workload_key, "single_gpu"):          # this workload's own counters when they were collected, else the single-GPU leg's (shares carry over, totals do not)
        rec = doc.get("workloads", {}).get(wl, {}).get(sym)
        if rec and rec.get("codec") == CODEC_OF[fmt]:
            out = dict(rec["derived"])
            out["from"] = "profiles/%s_sq_counters.json" % tag
            out["workload"] = wl
            if probes and sym in probes:                 # what the kernel's time is made of, measured by changing its hot loop: the table is in that file, not here
                out["probes"] = "profiles/%s_probes.json" % ptag
            return out
    return None


def roofline(fmt, prof, in_bytes, out_bytes, steps, workload_key):
    tot_ms = sum(v[0] for v in prof.values())
    if not prof:
        return None
    dom = max(prof.items(), key=lambda kv: kv[1][0])[0]
    ms, cnt = prof[dom]
    per_launch_ms = ms / cnt
    launches_per_step = cnt / steps
    # algorithmic bytes: 1 B HBM read + CR B HBM write per input byte, for the units one launch processes
    alg = (in_bytes + out_bytes) / launches_per_step
    ach = alg / (per_launch_ms * 1e-3) / 1e9
    traffic, traffic_tag = pmc_traffic(fmt, dom, workload_key)
    return {"bound": "hbm", "kernel": SYMBOLS.get((fmt, dom), dom), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
            "traffic_from": ("profiles/%s_pmc_traffic.json" % traffic_tag) if traffic else None,
            "hbm_read_frac": round(in_bytes / launches_per_step / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "kernel_ms_per_launch": round(per_launch_ms, 4), "launches_per_step": launches_per_step,
            "kernel_share_of_gpu_time": round(ms / tot_ms, 3) if tot_ms else None,
            "algorithmic_bytes_per_launch": int(alg),
            "secondary": secondary_bound(fmt, dom, workload_key),
            "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}


# ---------------------------------------------------------------- CPU baselines ----------------------------------------------------------------
def _cpu_timed(fn, fmt, units, caps, cores, budget_s):
    """passes and seconds of fn over the units on `cores` C threads (oracle.loader.time_units: no interpreter between the calls)"""
    from oracle import loader
    dt1, st, _ = loader.time_units(fn, fmt, units, caps, cores, 1)
    assert bool((st == 0).all()), "the CPU baseline reported an error status"
    more = max(0, min(19, int(budget_s / max(dt1, 1e-3)) - 1))
    dt = dt1 + (loader.time_units(fn, fmt, units, caps, cores, more)[0] if more else 0.0)
    return 1 + more, dt


def _malloc_tuning():
    """the reference's LZNT1 encoder reallocs its output as it grows and does not scale across threads under glibc's
    defaults (every shrink / grow of a large block is an mmap_sem round trip: 23 MB/s on 8 threads against 30 MB/s on one, 138 MB/s with
    the two tunables raised). The same tunables MALLOC_TOP_PAD_ / MALLOC_TRIM_THRESHOLD_ set, through mallopt, for this process's CPU legs."""
    import ctypes
    try:
        libc = ctypes.CDLL(None)
        ok = libc.mallopt(-2, 64 << 20) == 1 and libc.mallopt(-1, 512 << 20) == 1        # M_TOP_PAD, M_TRIM_THRESHOLD
        return "mallopt(M_TOP_PAD = 64 MiB, M_TRIM_THRESHOLD = 512 MiB) set for the CPU legs (SURVEY 8d)" if ok else "mallopt refused: glibc defaults"
    except Exception:
        return "glibc defaults (mallopt not available)"


def cpu_baseline(fmt, cor, budget_s=10.0, sa_dict=False):
    """The reference's own CPU encoder (oracle/_ref, compiled from /root/reference) -- or our C port when that file did not travel -- on
    this host's cores over THE LEG'S OWN UNIT LIST (SURVEY.md 8d "exactly the same unit list"), bounded: Xpress = the independent 64 KiB
    units of the first replica (3 239 ms_compress calls per pass, handed to the threads one at a time); LZNT1 / Xpress+Huffman = whole files,
    one ms_compress call each, as many replicas of the 12 files as there are threads to take them (at most the job's 16), longest first --
    with whole files as units the pass lasts as long as the largest file on one core, whatever the core count. And on ONE thread over the
    first unit(s). sa_dict: LZNT1 by the reference compiled with MSCOMP_WITH_LZNT1_SA_DICT (oracle/_ref/libMSCompression_sa.so) over the
    12 files. Reported baseline, not the target."""
    from oracle import loader
    ref = (loader.load_ref_sa() if sa_dict else loader.load_ref())
    if sa_dict and ref is None:
        return None
    kind = "reference" if ref is not None else "port"
    host_cores = os.cpu_count() or 1
    threads = max(1, min(host_cores, 256))            # (oracle/mscomp_oracle.c orc_time_units*: at most 256 threads)
    malloc = _malloc_tuning()
    blob = cor.blob()
    if fmt == 3:
        offs, lens = [], []
        for o, l in zip(cor.foff, cor.flen):
            st = np.arange(0, int(l), 65536, dtype=np.uint64)
            offs.append(st + o); lens.append(np.minimum(65536, int(l) - st).astype(np.uint64))
        uoff, ulen = np.concatenate(offs), np.concatenate(lens)
        what = "the %d independent 64 KiB units of the first replica (configs[2]'s unit list), one ms_compress call each" % len(ulen)
    else:
        reps = 1 if sa_dict else max(1, min(REPLICAS, threads // 12))
        uoff, ulen = np.tile(cor.foff, reps), np.tile(cor.flen, reps)
        order = np.argsort(-ulen.astype(np.int64), kind="stable")
        uoff, ulen = uoff[order], ulen[order]
        what = "%d replica(s) of the 12 files = %d whole-file units of the job, one ms_compress call each, longest first" % (reps, len(ulen))
    caps = np.array([loader.load_oracle().orc_max_compressed_size(fmt, int(x)) + 2 for x in ulen], np.uint64)
    fn = ref.ms_compress if ref is not None else None
    nthr = min(threads, len(ulen))
    busy_l = []
    dt1, st, _ = loader.time_units_ex(fn, fmt, blob, uoff, ulen, caps, nthr, 1, busy=busy_l)
    assert bool((st == 0).all()), "the CPU baseline reported an error status"
    more = max(0, min(19, int(budget_s / 4 / max(dt1, 1e-3)) - 1))
    dt = dt1
    if more:
        dt += loader.time_units_ex(fn, fmt, blob, uoff, ulen, caps, nthr, more, busy=busy_l)[0]
    passes, sample = 1 + more, int(ulen.sum())
    # the load-balanced figure: bytes x threads / thread-seconds spent inside ms_compress -- what these cores give when every thread always has
    # a unit to take (a queue of many files); `value` is the SAME UNIT LIST as the GPU leg in one pass, whose whole-file legs last as long as
    # the largest file on one core. Both are printed; a speed-up should be read against the balanced one.
    balanced = sample * passes * nthr / max(sum(busy_l), 1e-9) / 1e6
    k1 = 1 if fmt != 3 else min(64, len(ulen))         # single thread: the first file (its first 64 units for Xpress)
    s1 = int(ulen[:k1].sum())
    d1, st1, _ = loader.time_units_ex(fn, fmt, blob, uoff[:k1], ulen[:k1], caps[:k1], 1, 1)
    p1 = 1 + max(0, min(9, int(min(2.0, budget_s / 4) / max(d1, 1e-3)) - 1))
    if p1 > 1:
        d1 += loader.time_units_ex(fn, fmt, blob, uoff[:k1], ulen[:k1], caps[:k1], 1, p1 - 1)[0]
    return {"value": round(sample * passes / dt / 1e6, 1), "unit": "MB/s", "cores": nthr, "host_cores": host_cores, "kind": kind,
            "balanced_value": round(balanced, 1),
            "single_thread": {"value": round(s1 * p1 / d1 / 1e6, 1), "unit": "MB/s", "sample": "%d pass(es) over the first %d unit(s), %d B, one thread" % (p1, k1, s1)},
            "malloc": malloc,
            "sample": "%d pass(es) over %s: %d B per pass on %d threads" % (passes, what, sample, nthr)}


def cpu_decompress_baseline(fmt, blob, budget_s=3.0, whole=None):
    """The reference's CPU decoder beside the GPU decompression leg: the same kind of bounded sample (64 KiB units for the Xpress
    formats, 4 MiB pieces for LZNT1; compressed by the reference, untimed), all host cores, output MB/s. whole = [(offset, length)]:
    these buffers instead, one ms_decompress call each (the 12 files: at most 12 threads have work)."""
    from oracle import loader
    ref = loader.load_ref()
    kind = "reference" if ref is not None else "port"
    cores = max(1, min(os.cpu_count() or 1, 256))
    unit = (4 << 20) if fmt == 2 else 65536
    per_thread = 4 << 20
    sample = min(len(blob), per_thread * cores) // unit * unit
    data = blob[:sample].tobytes() if whole is None else None
    comp = loader.ref_compress if ref is not None else loader.oracle_compress
    if whole is not None:
        units = [blob[int(o):int(o) + int(l)].tobytes() for o, l in whole]
        sample = sum(len(u) for u in units)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(cores, len(units))) as ex:      # (ctypes drops the GIL inside the call)
            streams = list(ex.map(lambda u: comp(fmt, u)[1], units))
        threads = min(cores, len(units))
        fn = ref.ms_decompress if ref is not None else loader.load_oracle().orc_decompress
        passes, dt = _cpu_timed(fn, fmt, streams, [len(u) for u in units], threads, budget_s)
        return {"value": round(sample * passes / dt / 1e6, 1), "unit": "MB/s (output)", "cores": threads, "kind": kind,
                "sample": "%d passes over %d whole buffers (%d B), one ms_decompress call each" % (passes, len(units), sample)}
    units = [data[o:o + unit] for o in range(0, sample, unit)]
    streams = [comp(fmt, u)[1] for u in units]
    threads = min(cores, len(units))
    fn = ref.ms_decompress if ref is not None else loader.load_oracle().orc_decompress
    passes, dt = _cpu_timed(fn, fmt, streams, [len(u) for u in units], threads, budget_s)
    return {"value": round(sample * passes / dt / 1e6, 1), "unit": "MB/s (output)", "cores": threads, "kind": kind,
            "sample": "%d passes over the first %d B of the batch as %d independent ms_decompress calls of %d B" % (passes, sample, len(units), unit)}


# ---------------------------------------------------------------- legs ----------------------------------------------------------------
def decompress_leg(m, ctx, fmt, blob, in_off, in_len, desc, steps, sharding):
    """decode on the GPU what the GPU compressor wrote for this workload, check that the input comes back,
    report decompressed MB/s (HBM-resident, like `value`)."""
    import torch
    job = Job(m, ctx, fmt, blob, in_off, in_len)
    job.step(); torch.cuda.synchronize()
    comp_len = job.d_len.cpu().numpy().astype(np.uint64)
    assert bool((job.d_st == 0).all().item())
    d_back = torch.zeros(len(blob) + 16, dtype=torch.uint8, device=job.d_in.device)
    plan = m.Plan(ctx, fmt, job.out_off, comp_len, in_off, in_len, decompress=True)
    d_len2 = torch.zeros_like(job.d_len); d_st2 = torch.full_like(job.d_st, -9)

    class D:
        pass
    d = D(); d.ctx = ctx
    d.step = lambda: plan.execute(job.d_out, d_back, d_len2, d_st2)
    dt, prof = timed(d, steps, 1, sharding)
    ok = bool((d_st2 == 0).all().item()) and bool(torch.equal(d_len2.cpu(), torch.from_numpy(in_len.astype(np.int64)))) \
        and bool(torch.equal(d_back[: len(blob)], job.d_in[: len(blob)]))
    res = {"MB_per_s": round(job.in_bytes * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 3), "workload": desc,
           "round_trip_ok": ok, "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}
    plan.close(); job.close()
    return res


def end_to_end_leg(m, fmt, blob, in_off, in_len, desc, reps=3):
    """Host pointers in, host pointers out, wall clock (SURVEY.md 8d "end-to-end"): the units as views of one pageable numpy array,
    the outputs capacity after capacity in another, through mscomp_amd_compress_units_host on this GPU (uploads, kernels and downloads
    of 512 MiB batches overlapped; one upload per run of adjacent units). One untimed call first (contexts, scratch, staging), then the
    median of `reps`. Never `value` (that is HBM-resident)."""
    caps = np.array([m.max_compressed_size(fmt, int(l)) + 2 for l in in_len], np.uint64)
    out = np.zeros(int(caps.sum()) + 64, dtype=np.uint8)
    ooff = np.zeros(len(caps), np.uint64); ooff[1:] = np.cumsum(caps)[:-1]
    ins, outs = m.HostViews(blob, in_off, in_len), m.HostViews(out, ooff, caps)       # (pointer tables built by numpy: what a C caller hands over)
    ts = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        rc, lens, st = m.compress_units_host(fmt, ins, outs, devices=(0,))
        ts.append(time.perf_counter() - t0)
        assert rc == 0 and bool((st == 0).all()), "the end-to-end leg reported an error status"
    t = sorted(ts[1:])[len(ts[1:]) // 2]
    n = int(np.sum(in_len))
    return {"MB_per_s": round(n / t / 1e6, 1), "ms": round(t * 1e3, 3), "out_bytes": int(lens.sum()),
            "what": "mscomp_amd_compress_units_host, %s: pageable host memory in and out, wall clock, median of %d calls" % (desc, reps)}


def real_files_leg(m, ctx, sharding, max_mb, data_dir, steps=3, cpu=True):
    """REAL files instead of the generated corpus (tools/real_corpus.py: a deterministic list of files from this box -- ELF
    objects, archives, GPU code tables, Python source, text -- or every file of --data-dir / SILESIA_DIR): per codec the HBM-resident rate, the
    compression ratio, the kernels' times, the reference's CPU encoder over the same unit list on this host's threads, and the verdict of a
    byte-for-byte comparison of EVERY unit with what that encoder wrote. Goes to bench_extra.json (`extra.real_files`), never into `value`."""
    import torch
    from tools import real_corpus
    from oracle import loader
    t0 = time.perf_counter()
    rc = real_corpus.RealCorpus(max_mb << 20, data_dir)
    res = {"source": rc.source, "files": len(rc.paths), "bytes": rc.total, "kinds": rc.kinds(), "largest_file": int(rc.len.max()) if len(rc.len) else 0,
           "read_s": round(time.perf_counter() - t0, 1)}
    dev = torch.device("cuda", ctx.device)
    d_blob = torch.from_numpy(rc.blob).to(dev)
    ref = loader.load_ref()
    fn = ref.ms_compress if ref is not None else None
    threads = max(1, min(os.cpu_count() or 1, 256))
    for codec in ("lznt1", "xpress", "xpress_huff"):
        fmt = m.FORMATS[codec]
        uoff, ulen, _ = rc.units(65536 if fmt == 3 else None)
        job = Job(m, ctx, fmt, d_blob, uoff, ulen)
        dt, prof = timed(job, steps, 1, sharding)
        out_bytes = job.out_bytes()
        leg = {"MB_per_s": round(job.in_bytes * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 3), "units": int(len(ulen)),
               "what": "independent 64 KiB units" if fmt == 3 else "one unit per file", "compression_ratio": round(out_bytes / job.in_bytes, 4),
               "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}
        if cpu:
            caps = np.array([loader.load_oracle().orc_max_compressed_size(fmt, int(x)) + 2 for x in ulen], np.uint64)
            order = np.argsort(-ulen.astype(np.int64), kind="stable")
            busy = []
            nthr = min(threads, len(ulen))
            cdt, st, ln, out, ooff = loader.time_units_ex(fn, fmt, rc.blob, uoff[order], ulen[order], caps[order], nthr, 1, keep_output=True, busy=busy)
            assert bool((st == 0).all()), "the host reference reported an error status"
            d_packed, d_poff = m.compact_batch(ctx, job.out_off, job.caps, job.d_out, job.d_len)
            torch.cuda.synchronize()
            poff = d_poff.cpu().numpy().astype(np.int64)
            packed = d_packed[: int(poff[-1])].cpu().numpy()
            same = bool(np.array_equal(np.diff(poff).astype(np.uint64)[order], ln))
            for k, i in enumerate(order):
                if not same:
                    break
                same = bool(np.array_equal(packed[int(poff[i]):int(poff[i + 1])], out[int(ooff[k]):int(ooff[k]) + int(ln[k])]))
            leg.update({"identical_to_reference": same, "checker": "reference" if ref is not None else "port",
                        "cpu_MB_per_s": round(job.in_bytes / cdt / 1e6, 1), "cpu_balanced_MB_per_s": round(job.in_bytes * nthr / max(sum(busy), 1e-9) / 1e6, 1), "cpu_threads": nthr})
            if not same:
                sys.exit("bench.py: the %s output for the real files differs from the reference's: no number is reported" % codec)
        job.close()
        # the same codec per KIND of file (>= 128 MB of it): how much the rate depends on what the bytes are
        by_kind = {}
        kof = np.array(rc.kind_of_files())
        fo, fl, _ = rc.units(None)
        for kind in sorted(set(kof.tolist())):
            sel = np.nonzero(kof == kind)[0]
            if int(fl[sel].sum()) < (128 << 20):                    # (a smaller batch measures its size, not its bytes: 38 MB run at a third of these rates)
                continue
            if fmt == 3:
                ko, kl = [], []
                for i in sel:
                    st = np.arange(0, int(fl[i]), 65536, dtype=np.uint64)
                    ko.append(st + fo[i]); kl.append(np.minimum(65536, int(fl[i]) - st).astype(np.uint64))
                ko, kl = np.concatenate(ko), np.concatenate(kl)
            else:
                ko, kl = fo[sel], fl[sel]
            jk = Job(m, ctx, fmt, d_blob, ko, kl)
            tk, _ = timed(jk, 2, 1, sharding)
            by_kind[kind] = {"MB_per_s": round(jk.in_bytes * 2 / tk / 1e6, 1), "bytes": jk.in_bytes, "compression_ratio": round(jk.out_bytes() / jk.in_bytes, 4)}
            jk.close()
        leg["by_kind"] = by_kind
        res[codec] = leg
    return res


def sharded_leg(m, ctx, cor, fmt, rank, world, steps, warmup, sharding, dev):
    """One codec over the config-5 job: this rank's contiguous unit range, timed; whole-job figures by MAX / SUM over the ranks."""
    off, ln, desc = config5_units(cor, fmt)
    s, e, g0, g1, my_off, my_len = sharding.shard_job(off, ln, world, rank)
    d_in = cor.device_range(g0, g1)
    job = Job(m, ctx, fmt, d_in, my_off, my_len)
    dt, prof = timed(job, steps, warmup, sharding)
    out_bytes = job.out_bytes()
    parity = parity_gate(m, job, fmt, cor, int(s))
    if parity is False:
        sys.exit("bench.py: the %s output of this run differs from the reference's (tests/golden/corpus_full.json): no number is reported" % CODEC_OF[fmt])
    job_dt, job_bytes = sharding.reduce_job(dt, job.in_bytes * steps, device=dev)
    _, job_out = sharding.reduce_job(0.0, out_bytes, device=dev)
    assert job_bytes == int(ln.sum()) * steps, "the shards do not cover the job"
    res = {"MB_per_s": round(job_bytes / job_dt / 1e6, 1), "MiB_per_s": round(job_bytes / job_dt / 2 ** 20, 1), "ms_per_step": round(job_dt / steps * 1e3, 4),
           "steps": steps, "workload": desc, "units": int(len(ln)), "units_rank0": int(e - s), "bytes_rank0": job.in_bytes,
           "compression_ratio": round(job_out / int(ln.sum()), 4), "parity_checked": parity,
           "roofline": roofline(fmt, prof, job.in_bytes, out_bytes, steps, "config5_n%d" % world)}
    job.close()
    return res


def short_line(res):
    """What rank 0 prints: the contract's keys, scalars only -- `config` flat, `roofline` and `cpu_baseline` cut to their headline keys, no `extra`, no
    sentence longer than a workload name. (tests/test_bench_host.py checks this on the committed document of the last profiled run.)"""
    line = {k: v for k, v in res.items() if k != "extra"}
    if res.get("roofline"):
        line["roofline"] = {k: res["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_per_launch",
                                                               "launches_per_step", "algorithmic_bytes_per_launch")}
    if res.get("cpu_baseline"):
        cb = res["cpu_baseline"]
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "balanced_value": cb["balanced_value"],
                                "single_thread_value": cb["single_thread"]["value"], "sample": cb["sample"][:160]}
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--codec", default="lznt1", choices=["lznt1", "xpress", "xpress_huff"], help="headline codec")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-extra", action="store_true", help="headline leg only")
    ap.add_argument("--config5-only", action="store_true", help="the three codecs over BASELINE configs[4], none of the other legs")
    ap.add_argument("--full", action="store_true", help="print the whole document (every leg, every kernel) instead of the short line; it is written to bench_extra.json either way")
    ap.add_argument("--real-files", type=int, nargs="?", const=1100, default=0, metavar="MB", help="also run the three codecs over REAL files of this box (tools/real_corpus.py; "
                    "~1.1 GB by default, read from disk: minutes on a fresh box) and compare every unit with the reference's CPU encoder; results in bench_extra.json")
    ap.add_argument("--data-dir", default=None, help="the real-files leg over every file of this directory instead")
    ap.add_argument("--oversubscribe", action="store_true", help="TEST ONLY: let the N ranks share the visible GPUs (gloo for the timing reduction); "
                    "exercises the sharded multi-rank path on a 1-GPU box, the line is marked and is not an N-GPU measurement")
    args = ap.parse_args()
    want = max(1, args.gpus)
    if want > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(want)

    import torch
    import ms_compress_amd as m
    from ms_compress_amd import corpus, sharding
    # (MSCOMP_AMD_BENCH_BACKEND overrides; two ranks on ONE GPU cannot form an RCCL group, so the test mode asks for gloo)
    rank, local_rank, world = sharding.dist_env()
    if args.oversubscribe:
        local_rank %= max(1, torch.cuda.device_count())
    rank, _, world = sharding.init_distributed(os.environ.get("MSCOMP_AMD_BENCH_BACKEND") or ("gloo" if args.oversubscribe else None), device=local_rank)
    if world != want:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s); launch with torch.distributed.run --nproc-per-node %d "
                 "(or let bench.py spawn them: no WORLD_SIZE in the environment)" % (want, world, want))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rdev = None                                          # the timing reduction's tensors: sharding.reduce_job puts them where its backend needs them
    ctx = m.Context(device=local_rank)
    cor = Corpus(corpus, dev)
    fmt = m.FORMATS[args.codec]
    steps2 = max(3, args.steps // 4)

    head = sharded_leg(m, ctx, cor, fmt, rank, world, args.steps, args.warmup, sharding, rdev)
    res = {
        "metric": "input MB/s (%s compress, bit-exact with the reference CPU encoder)" % args.codec,
        "value": head["MB_per_s"], "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic" if corpus.source() == "synthetic" else corpus.source(),
        "config": {"workload": "BASELINE configs[4], %s: %s" % (args.codec, head["workload"]),
                   "bytes_per_step": cor.total * REPLICAS, "units": head["units"],
                   "bytes_rank0": head["bytes_rank0"], "compression_ratio": head["compression_ratio"],
                   "parallelism": "shard-per-gpu x%d, no data-path collective" % world,
                   "backend": sharding.backend_name(),
                   "MiB_per_s": head["MiB_per_s"]},
        "roofline": head["roofline"],
        "parity_checked": {args.codec: head["parity_checked"]},
    }
    if args.oversubscribe:
        res["oversubscribed"] = "TEST RUN: %d ranks shared %d GPU(s); not an %d-GPU measurement" % (world, torch.cuda.device_count(), world)
    if rank == 0 and world == 1 and not args.no_cpu:
        res["cpu_baseline"] = cpu_baseline(fmt, cor)
    extra = {}
    if not args.no_extra:
        c5 = {}
        for codec in ("lznt1", "xpress", "xpress_huff"):
            if codec != args.codec:
                c5[codec] = sharded_leg(m, ctx, cor, m.FORMATS[codec], rank, world, steps2, 1, sharding, rdev)
                res["parity_checked"][codec] = c5[codec]["parity_checked"]
                if rank == 0 and world == 1 and not args.no_cpu:
                    c5[codec]["cpu_baseline"] = cpu_baseline(m.FORMATS[codec], cor)
        extra["config5"] = c5
    if world == 1 and not args.no_extra and not args.config5_only:
        single = {}
        for key, wl in (("lznt1", "mozilla"), ("xpress", "silesia_units64k"), ("xpress_huff", "silesia_files"), ("xpress_huff_unit_mode", "silesia_units64k")):
            codec = "xpress_huff" if key.startswith("xpress_huff") else key      # Xpress+Huffman in file mode AND in unit mode (every 64 KiB unit independent, each with EOS)
            f2 = m.FORMATS[codec]
            b2, o2, l2, d2 = single_gpu_workload(cor, wl)
            j2 = Job(m, ctx, f2, b2, o2, l2)
            n2 = args.steps if codec == "lznt1" else steps2
            t2, p2 = timed(j2, n2, 3 if codec == "lznt1" else 1, sharding)
            ob = j2.out_bytes()
            single[key] = {"MB_per_s": round(j2.in_bytes * n2 / t2 / 1e6, 1), "ms_per_step": round(t2 / n2 * 1e3, 4), "steps": n2,
                             "workload": d2, "compression_ratio": round(ob / j2.in_bytes, 4), "roofline": roofline(f2, p2, j2.in_bytes, ob, n2, "single_gpu")}
            j2.close()
            single[key]["end_to_end"] = end_to_end_leg(m, f2, b2, o2, l2, d2)
            assert single[key]["end_to_end"]["out_bytes"] == ob, "the host-pointer path and the HBM-resident path disagree on the output size"
        extra["single_gpu"] = single
        # What ONE rank of an 8-GPU run of this job holds (2 of the 16 replicas, 423 877 160 B): its rate against an eighth of the
        # one-GPU rate is the strong-scaling efficiency the partition itself allows at N = 8 (no exchange step exists; what is lost is the
        # tail of kernels whose duration is their slowest chunk). Not an 8-GPU measurement -- none is possible on this box.
        r8 = {}
        for codec in ("lznt1", "xpress", "xpress_huff"):
            f2 = m.FORMATS[codec]
            off, ln, _ = config5_units(cor, f2)
            nu = len(ln) // REPLICAS * 2
            j2 = Job(m, ctx, f2, cor.device_range(0, 2 * cor.total), off[:nu], ln[:nu])
            t2, p2 = timed(j2, steps2, 1, sharding)
            full = head if codec == args.codec else extra["config5"][codec]
            r8[codec] = {"MB_per_s": round(j2.in_bytes * steps2 / t2 / 1e6, 1), "ms_per_step": round(t2 / steps2 * 1e3, 4), "bytes": j2.in_bytes,
                         "rate_vs_whole_job_on_one_gpu": round((j2.in_bytes * steps2 / t2 / 1e6) / full["MB_per_s"], 3),
                         "kernels_ms_per_step": {k: round(v[0] / steps2, 4) for k, v in sorted(p2.items(), key=lambda kv: -kv[1][0])}}
            j2.close()
        extra["one_rank_of_8"] = r8
        # the suffix-array dictionary flavour of LZNT1 (csrc/lznt1_sa.hip) on the 12 files; HIP events per kernel as everywhere
        b2, o2, l2, d2 = single_gpu_workload(cor, "silesia_files")
        ctx.set_lznt1_sa_dict(True)                      # (per context: the plans this context creates from here on; nothing process-wide)
        try:
            j2 = Job(m, ctx, m.FORMATS["lznt1"], b2, o2, l2)
            t2, p2 = timed(j2, steps2, 1, sharding)
            extra["lznt1_sa_dict"] = {"MB_per_s": round(j2.in_bytes * steps2 / t2 / 1e6, 1), "ms_per_step": round(t2 / steps2 * 1e3, 4), "steps": steps2, "workload": d2,
                                      "compression_ratio": round(j2.out_bytes() / j2.in_bytes, 4),
                                      "kernels_ms_per_step": {k: round(v[0] / steps2, 4) for k, v in p2.items()}}
            j2.close()
            if not args.no_cpu:
                extra["lznt1_sa_dict"]["cpu_baseline"] = cpu_baseline(m.FORMATS["lznt1"], cor, sa_dict=True)
        finally:
            ctx.set_lznt1_sa_dict(None)
        dec = {}
        for codec, wl in (("lznt1", "mozilla"), ("xpress", "silesia_units64k"), ("xpress_huff", "silesia_units64k")):
            f2 = m.FORMATS[codec]
            b2, o2, l2, d2 = single_gpu_workload(cor, wl)
            dec[codec] = decompress_leg(m, ctx, f2, b2, o2, l2, d2, 3, sharding)
            if not args.no_cpu:
                dec[codec]["cpu_baseline"] = cpu_decompress_baseline(f2, b2)
        extra["decompress"] = dec
        # the same for whole files as single buffers (large units: chunk-parallel Huffman walk / one serial Xpress token chain; bytes by all CUs, lzglobal.hip)
        decf = {}
        for codec in ("xpress_huff", "xpress"):
            f2 = m.FORMATS[codec]
            b2, o2, l2, d2 = single_gpu_workload(cor, "silesia_files")
            decf[codec] = decompress_leg(m, ctx, f2, b2, o2, l2, d2, 2, sharding)
            if not args.no_cpu:
                decf[codec]["cpu_baseline"] = cpu_decompress_baseline(f2, b2, whole=list(zip(o2, l2)))
        extra["decompress_files"] = decf
    if world == 1 and (args.real_files or args.data_dir):
        extra["real_files"] = real_files_leg(m, ctx, sharding, args.real_files or 1100, args.data_dir, cpu=not args.no_cpu)
    # The line the driver parses: SHORT (< 4 KB), scalars only, no sentences. Everything else -- the other legs, every kernel's time,
    # the secondary-bound counters -- goes to bench_extra.json beside this script (and gpurun_out/ when that exists, so that it travels back).
    legs = {args.codec: head}
    legs.update(extra.get("config5", {}))
    for codec, leg in legs.items():
        res["config"]["%s_MB_per_s" % codec] = leg["MB_per_s"]
        res["config"]["%s_ms_per_step" % codec] = leg["ms_per_step"]
        res["config"]["%s_roofline_frac" % codec] = (leg["roofline"] or {}).get("frac")
        res["config"]["%s_parity_checked" % codec] = leg["parity_checked"]
        cb = res.get("cpu_baseline") if codec == args.codec else leg.get("cpu_baseline")
        if cb:
            res["config"]["%s_cpu_MB_per_s" % codec] = cb["value"]
            res["config"]["%s_cpu_balanced_MB_per_s" % codec] = cb["balanced_value"]
    if extra.get("real_files"):
        for codec in ("lznt1", "xpress", "xpress_huff"):
            res["config"]["%s_real_files_MB_per_s" % codec] = extra["real_files"][codec]["MB_per_s"]
    if extra.get("one_rank_of_8"):
        for codec, r in extra["one_rank_of_8"].items():
            res["config"]["%s_one_rank_of_8_rate_vs_whole_job" % codec] = r["rate_vs_whole_job_on_one_gpu"]
    full = dict(res)
    if extra:
        full["extra"] = extra
    line = short_line(res)
    if rank == 0:
        for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
            if os.path.isdir(d):
                try:
                    json.dump(full, open(os.path.join(d, "bench_extra.json"), "w"), indent=1)
                except OSError:
                    pass
        if args.full:
            line = full
        text = json.dumps(line, separators=(",", ":"))
        if not args.full and len(text) >= 4000:            # never print a line the driver cannot keep: shed the optional parts instead (bench_extra.json has them)
            for k in [k for k in line["config"] if k.endswith(("_cpu_MB_per_s", "_cpu_balanced_MB_per_s", "_one_rank_of_8_rate_vs_whole_job", "_ms_per_step", "_real_files_MB_per_s"))]:
                del line["config"][k]
            if "cpu_baseline" in line:
                line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:60]
            line["config"]["workload"] = line["config"]["workload"][:120]
            text = json.dumps(line, separators=(",", ":"))
        print(text, flush=True)
    ctx.close()
    if world > 1:
        sys.stdout.flush(); sys.stderr.flush()
        sharding.barrier()
        if sharding.abandoned_bringup():                 # a hung RCCL bring-up thread is still inside some rank's runtime: no teardown through it, on any rank
            os._exit(0)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

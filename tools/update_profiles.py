"""Condense a gpurun_out/prof_<tag> directory (tools/collect_profiles.sh) into the tracked profiles/<tag>_* files.
usage: python tools/update_profiles.py gpurun_out/prof_r02 r02"""
import collections, csv, glob, json, os, shutil, sys
src, tag = sys.argv[1], sys.argv[2]
P = lambda name: os.path.join("profiles", "%s_%s" % (tag, name))
short = lambda k: k.replace("void ", "").split("(")[0]


def stats_table(path, title):
    rows = list(csv.DictReader(open(path)))
    out = ["# %s" % title, "# %-70s %8s %14s %12s %7s" % ("kernel", "calls", "avg_us", "total_ms", "share")]
    for r in rows:
        if "msc::" in r["Name"]:
            out.append("%-72s %8s %14.2f %12.3f %6.2f%%" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
    return "\n".join(out) + "\n"


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if "msc::" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(x) / len(x), len(x)) for c, x in v.items()} for k, v in agg.items()}


if os.path.exists(os.path.join(src, "bench.json")) and os.path.getsize(os.path.join(src, "bench.json")):
    shutil.copy(os.path.join(src, "bench.json"), P("bench.json"))                 # the ONE short line the driver parses
if os.path.exists(os.path.join(src, "bench_extra.json")) and os.path.getsize(os.path.join(src, "bench_extra.json")):
    shutil.copy(os.path.join(src, "bench_extra.json"), P("bench_extra.json"))     # the whole document of the same run (every leg, every kernel)
for f in ("oneshot_host_pointers.txt", "issue_peak.txt", "issue_peak2.txt"):
    if os.path.exists(os.path.join(src, f)) and os.path.getsize(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), P(f))
for f in ("bench_headline_under_rocprof.json", "bench_full_under_rocprof.json"):
    if os.path.exists(os.path.join(src, f)) and os.path.getsize(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), P(f))
if os.path.exists(os.path.join(src, "kt_head_kernel_stats.csv")):
    txt = stats_table(os.path.join(src, "kt_head_kernel_stats.csv"),
                      "rocprofv3 --kernel-trace --stats -- python bench.py --no-extra --no-cpu --steps 5 --warmup 1   (the headline leg alone: LZNT1 over BASELINE configs[4])")
    if os.path.exists(os.path.join(src, "kt_full_kernel_stats.csv")):
        txt += "\n" + stats_table(os.path.join(src, "kt_full_kernel_stats.csv"),
                                  "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --steps 4 --warmup 1   (all legs, compressors and decompressors: a kernel's launches mix workloads)")
    for codec in ("lznt1", "xpress", "xpress_huff"):
        f = os.path.join(src, "kt_config5_%s_kernel_stats.csv" % codec)
        if os.path.exists(f):
            txt += "\n" + stats_table(f, "rocprofv3 --kernel-trace --stats -- python tools/gpu_leg.py config5:%s 3   (ONE codec over BASELINE configs[4], 1 warm-up + 3 timed passes: every launch is this workload)" % codec)
    open(P("kernel_stats.txt"), "w").write(txt)

WL = {"config5": "config5_n1", "single": "single_gpu", "decompress": "decompress_units64k"}
traffic = {}
for f in sorted(glob.glob(os.path.join(src, "fetch_*_counter_collection.csv"))):
    leg = os.path.basename(f)[len("fetch_"):-len("_counter_collection.csv")]
    kind, codec = leg.split("_", 1)
    fe, wr = per_kernel(f), per_kernel(f.replace("fetch_", "write_"))
    for k in fe:
        rec = {"codec": codec, "FETCH_SIZE_KB_per_launch": round(fe[k]["FETCH_SIZE"][0], 2), "launches_FETCH_SIZE": fe[k]["FETCH_SIZE"][1],
               "WRITE_SIZE_KB_per_launch": round(wr.get(k, {}).get("WRITE_SIZE", (0, 0))[0], 2), "launches_WRITE_SIZE": wr.get(k, {}).get("WRITE_SIZE", (0, 0))[1]}
        # gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section)
        rec["hbm_bytes_per_launch_corrected"] = int(rec["FETCH_SIZE_KB_per_launch"] * 1024 * 2 + rec["WRITE_SIZE_KB_per_launch"] * 1024)
        traffic.setdefault(WL[kind], {}).setdefault(codec, {})[k] = rec
flat = {w: {k: r for c in v.values() for k, r in c.items() if k not in ("msc::scan_tiles_kernel", "msc::scan_add_kernel", "msc::scan_tile_sums_kernel", "msc::finalize_units_kernel")
            or True} for w, v in traffic.items()}
json.dump({"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python tools/gpu_leg.py <leg> 3 (one bench.py leg per run, so every launch of a kernel "
                  "belongs to one workload); counters are KB; FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE uncalibrated. "
                  "Kernels that several codecs share (xp_links_kernel, scan / finalize) appear under the codec that ran them; bench.py looks a kernel up by workload + codec.",
           "workload_keys": {"config5_n1": "BASELINE configs[4] on one GPU: 16x replicated corpus, 3391017280 B", "single_gpu": "BASELINE configs[1..3]: mozilla 51220480 B (lznt1), 3239 x 64 KiB units (xpress), 12 files (xpress_huff), 211938580 B"},
           "by_codec": traffic, "workloads": {w: {k: r for c in v.values() for k, r in c.items()} for w, v in traffic.items()}}, open(P("pmc_traffic.json"), "w"), indent=1)

N_XCD, N_CU = 8, 256
PEAK = {"valu_per_cu_cycle": 0.95, "valu_single_source_stream_per_cu_cycle": 1.8, "salu_per_cu_cycle": 1.0}      # tools/dev/issue_peak.hip + issue_peak2.hip, measured on this GPU (profiles/<tag>_issue_peak*.txt)
clock = {}                                     # codec -> {kernel: (GHz, avg ns)} from the GRBM_GUI_ACTIVE passes of the configs[4] legs
for f in sorted(glob.glob(os.path.join(src, "clk_config5_*_counter_collection.csv"))):
    codec = os.path.basename(f)[len("clk_config5_"):-len("_counter_collection.csv")]
    cyc = per_kernel(f)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f.replace("_counter_collection.csv", "_kernel_trace.csv"))):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    clock[codec] = {k: (v["GRBM_GUI_ACTIVE"][0] / N_XCD / (sum(dur[k]) / len(dur[k])), sum(dur[k]) / len(dur[k])) for k, v in cyc.items() if k in dur and "GRBM_GUI_ACTIVE" in v}

sq = {}
for f in sorted(glob.glob(os.path.join(src, "sqa_*_counter_collection.csv"))):
    leg = os.path.basename(f)[len("sqa_"):-len("_counter_collection.csv")]
    kind, codec = leg.split("_", 1)
    a, b = per_kernel(f), per_kernel(f.replace("sqa_", "sqb_"))
    for k in a:
        raw = {c: round(v[0]) for c, v in a[k].items()}
        raw.update({c: round(v[0]) for c, v in b.get(k, {}).items()})
        wc = max(1, raw.get("SQ_WAVE_CYCLES", 1)); act = max(1, raw.get("SQ_ACTIVE_INST_ANY", 1)); idx = max(1, raw.get("SQ_LDS_IDX_ACTIVE", 1))
        d = {"wave_time_waiting_frac": round(raw.get("SQ_WAIT_ANY", 0) / wc, 3), "wave_time_issue_stalled_frac": round(raw.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
             "wave_time_issuing_frac": round(act / wc, 3),
             "issue_mix": {"valu": round(raw.get("SQ_ACTIVE_INST_VALU", 0) / act, 3), "salu": round(raw.get("SQ_ACTIVE_INST_SCA", 0) / act, 3), "lds": round(raw.get("SQ_ACTIVE_INST_LDS", 0) / act, 3)},
             "lds_pipe_busy_frac": round(raw.get("SQ_LDS_IDX_ACTIVE", 0) / max(1, raw.get("SQ_BUSY_CU_CYCLES", 1)), 3),
             "lds_bank_conflict_share": round(raw.get("SQ_LDS_BANK_CONFLICT", 0) / idx, 3), "lds_unaligned_stall_share": round(raw.get("SQ_LDS_UNALIGNED_STALL", 0) / idx, 4),
             "lds_cycles_per_lds_instruction": round(idx / max(1, raw.get("SQ_INSTS_LDS", 1)), 2)}
        # Issue rates against what a CU can issue (round 5; replaces round 4's `valu_busy_frac`, a ratio of two counters with different units that came
        # out above 1): wave-instructions per CU and cycle over the CU's BUSY cycles. Measured peaks of this GPU (tools/dev/issue_peak.hip, issue_peak2.hip):
        # 0.95 vector instructions per CU cycle for any real mix -- the classic 4 cycles per wave64 instruction on each of the 4 SIMDs; only a stream made
        # of nothing but single-VGPR-source VOP1 / VOP2 instructions reaches 1.8, and one other instruction in four brings the whole stream back to 0.95
        # (so the bracket the first half of round 5 printed had the right upper end and a meaningless lower one) -- and 1.0 scalar (ONE scalar unit per CU).
        busy = max(1, raw.get("SQ_BUSY_CU_CYCLES", 1))
        v_rate, s_rate = raw.get("SQ_INSTS_VALU", 0) / busy, raw.get("SQ_INSTS_SALU", 0) / busy
        d["valu_insts_per_cu_cycle"] = round(v_rate, 3)
        d["salu_insts_per_cu_cycle"] = round(s_rate, 3)
        d["valu_issue_frac"] = round(min(1.0, v_rate / PEAK["valu_per_cu_cycle"]), 3)
        d["salu_issue_frac"] = round(min(1.0, s_rate / PEAK["salu_per_cu_cycle"]), 3)
        if kind == "config5" and k in clock.get(codec, {}):
            ghz, ns = clock[codec][k]
            d["clock_GHz_measured"] = round(ghz, 3)
            d["cu_busy_share_of_kernel"] = round(busy / N_CU / (ghz * ns), 3)
        d["bound"] = ("lds-pipe" if d["lds_pipe_busy_frac"] >= 0.8 else
                      "vector issue" if d["valu_issue_frac"] >= 0.9 else
                      "scalar unit (one per CU)" if d["salu_issue_frac"] >= 0.6 else
                      "lds-pipe" if d["lds_pipe_busy_frac"] >= 0.6 else
                      "mixed: vector + scalar + LDS issue, none saturated; latency at the resident waves" if d["wave_time_waiting_frac"] < 0.6 else
                      "latency (waves parked in s_waitcnt / barriers)")
        sq.setdefault(WL[kind], {})[k] = {"codec": codec, "derived": d, "per_launch": raw}
json.dump({"issue_peaks_measured": PEAK, "how": "rocprofv3 --kernel-trace --pmc <8 SQ counters> -- python tools/gpu_leg.py single:<codec> 3, two passes (A: SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                  "SQ_ACTIVE_INST_ANY/_VALU/_SCA/_LDS; B: SQ_INSTS_VALU/_SALU/_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS). "
                  "Averages per launch. wave-time fractions are shares of SQ_WAVE_CYCLES (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES); lds_pipe_busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES.",
           "workloads": sq}, open(P("sq_counters.json"), "w"), indent=1)
r = json.load(open(P("bench.json"))) if os.path.exists(P("bench.json")) else {"value": None, "roofline": {"frac": None, "traffic": None}}
print(r["value"], r["roofline"]["frac"], r["roofline"]["traffic"])
if os.path.exists(P("bench_extra.json")):
    r = json.load(open(P("bench_extra.json")))                     # (the decode tables below are made from the whole document)

# the decompression legs of the same bench line, kernel by kernel (HIP events), with the SQ shares of the decoder kernels
if "extra" in r and "decompress" in r["extra"]:
    out = ["# decompression legs of profiles/%s_bench.json (bench.py: HIP events around every kernel, ms per pass; output MB/s; cpu = the reference's decoder on the host cores named)" % tag]
    for group in ("decompress", "decompress_files"):
        for codec, v in r["extra"].get(group, {}).items():
            cb = v.get("cpu_baseline", {})
            out.append("%-17s %-14s %-70s %9.3f ms/pass %10.1f MB/s   cpu %s MB/s on %s threads (%s)   round trip %s" %
                       (group, codec, v["workload"][:70], v["ms_per_step"], v["MB_per_s"], cb.get("value"), cb.get("cores"), cb.get("kind"), "ok" if v["round_trip_ok"] else "BAD"))
            for k, ms in v["kernels_ms_per_step"].items():
                out.append("       %-28s %10.4f ms" % (k, ms))
    out.append("")
    out.append("# SQ counters of the decoder kernels on the 3 239-unit legs (profiles/%s_sq_counters.json, workload decompress_units64k): shares of wave time, issue mix, LDS pipe" % tag)
    for k, v in sq.get("decompress_units64k", {}).items():
        if any(x in k for x in ("xpt_", "lz_copy", "xhc_", "xhd_", "lzg_", "xps_")):
            d = v["derived"]
            out.append("%-34s %-12s waiting %.2f  issue-stalled %.2f  issuing %.2f | valu %.2f salu %.2f lds %.2f | LDS pipe busy %.2f | %s" %
                       (k.replace("msc::", ""), v["codec"], d["wave_time_waiting_frac"], d["wave_time_issue_stalled_frac"], d["wave_time_issuing_frac"],
                        d["issue_mix"]["valu"], d["issue_mix"]["salu"], d["issue_mix"]["lds"], d["lds_pipe_busy_frac"], d["bound"]))
    open(P("decompress_kernels.txt"), "w").write("\n".join(out) + "\n")

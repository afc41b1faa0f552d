"""Copy the summaries of a gpurun_out/prof_* directory (rocprofv3 kernel trace db, FETCH/WRITE PMC csv, bench JSON) into profiles/.
usage: python tools/update_profiles.py gpurun_out/prof_r01d r01"""
import csv, collections, json, os, shutil, subprocess, sys
src, tag = sys.argv[1], sys.argv[2]
open("profiles/%s_kernel_stats.txt" % tag, "w").write(subprocess.run([sys.executable, "tools/rocpd_summary.py", os.path.join(src, "kt_results.db")], capture_output=True, text=True, check=True).stdout)
shutil.copy(os.path.join(src, "bench.json"), "profiles/%s_bench.json" % tag)
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), "profiles/%s_bench_under_rocprof.json" % tag)
out = {}
for f, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, "%s_counter_collection.csv" % f))):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if "msc::" in k:
                agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[c + "_KB_per_launch"] = round(sum(v) / len(v), 2)
        out[k]["launches_" + c] = len(v)
for k, v in out.items():
    # gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section)
    v["hbm_bytes_per_launch_corrected"] = int(v.get("FETCH_SIZE_KB_per_launch", 0) * 1024 * 2 + v.get("WRITE_SIZE_KB_per_launch", 0) * 1024)
doc = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --steps 3 --warmup 1 --no-cpu ; "
              "counters are KB; FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE uncalibrated",
       "workloads": "lznt1: mozilla 51220480 B; xpress: 3239 x 64 KiB units (211938580 B); xpress_huff: 12 files (211938580 B); decompression (lzd_* / xpd / xhd kernels): what the compressor wrote for mozilla (lznt1) and for the 3239 units (xpress, xpress_huff)", "kernels": out}
json.dump(doc, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
r = json.load(open("profiles/%s_bench.json" % tag))
print(r["value"], r["roofline"]["frac"], r["roofline"]["traffic"], r["cpu_baseline"]["value"])
for k, v in r["extra"].items():
    if k == "decompress":
        for c, d in v.items():
            print("decompress", c, d["MB_per_s"], d["round_trip_ok"], d["kernels_ms_per_step"])
    else:
        print(k, v["MB_per_s"], v["roofline"]["frac"], v["roofline"]["kernels_ms_per_step"])

#!/bin/bash
# dev (GPU box): cache / TLB / latency counters of one leg's kernels (separate --pmc passes, kernel trace only)
#   tools/dev/r5_mem_probe.sh <leg, e.g. config5:xpress_huff> <out dir under gpurun_out>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
D=$R/gpurun_out/$2; mkdir -p $D
cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" \
           "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D -o mem$i -- python $R/tools/gpu_leg.py $1 2 > $D/mem$i.out 2>&1
done
python - "$D" <<'PY'
import csv, glob, sys, collections
D = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob(D + "/**/mem*_counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (f, r["Dispatch_Id"]) not in seen: seen.add((f, r["Dispatch_Id"])); cnt[(k, f)] += 1
for k, v in agg.items():
    if sum(v.values()) < 1e6: continue
    n = max(c for (kk, f), c in cnt.items() if kk == k)
    print(k, "launches/pass", n)
    for c, x in sorted(v.items()): print("   %-40s %.4g per launch" % (c, x / n))
PY

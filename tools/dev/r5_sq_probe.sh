#!/bin/bash
# dev (GPU box): issue / LDS counters of one leg's kernels, one --pmc pass per set (kernel trace only)
#   tools/dev/r5_sq_probe.sh <leg, e.g. single:lznt1> <out dir under gpurun_out> [MSCOMP_AMD_LIB]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
D=$R/gpurun_out/$2; mkdir -p $D
[ -n "$3" ] && export MSCOMP_AMD_LIB=$3
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D -o sq$i -- python $R/tools/gpu_leg.py $1 3 > $D/sq$i.out 2>&1
done
python - "$D" <<'PY'
import csv, glob, sys, collections
D = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob(D + "/**/sq*_counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (f, r["Dispatch_Id"]) not in seen: seen.add((f, r["Dispatch_Id"])); cnt[(k, f)] += 1
for k, v in agg.items():
    if v.get("SQ_INSTS_VALU", 0) < 1e7: continue
    n = max(c for (kk, f), c in cnt.items() if kk == k)
    cu = v["GRBM_GUI_ACTIVE"] / 8 * 256          # CU cycles of the launch(es)
    print("%s launches %d | per CU cycle: valu %.3f salu %.3f lds-inst %.3f | lds busy %.3f (conflict share %.2f) | wave time: waiting %.2f issuing %.2f | ms %.3f" % (
        k, n, v["SQ_INSTS_VALU"] / cu, v["SQ_INSTS_SALU"] / cu, v["SQ_INSTS_LDS"] / cu, v["SQ_LDS_IDX_ACTIVE"] / cu if cu else 0,
        v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), v.get("SQ_ACTIVE_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1),
        v["GRBM_GUI_ACTIVE"] / 8 / n / 2.38e6))
    print("    per launch: VALU %.4g SALU %.4g LDS %.4g" % (v["SQ_INSTS_VALU"] / n, v["SQ_INSTS_SALU"] / n, v["SQ_INSTS_LDS"] / n))
PY

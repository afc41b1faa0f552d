#!/bin/bash
# dev (GPU box): LDS stall counters of one leg   tools/dev/r5_lds_probe.sh <leg> <out dir> [lib]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
D=$R/gpurun_out/$2; mkdir -p $D
[ -n "$3" ] && export MSCOMP_AMD_LIB=$3
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $D -o lds -- python $R/tools/gpu_leg.py $1 3 > $D/lds.out 2>&1
python - "$D" <<'PY'
import csv, glob, sys, collections
D = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(D + "/**/lds_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)): agg[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if v.get("SQ_INSTS_LDS", 0) < 1e6: continue
    print(k, {c: "%.4g" % x for c, x in sorted(v.items())})
PY

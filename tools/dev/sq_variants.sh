#!/bin/bash
# dev (GPU box): SQ counters of the Xpress+Huffman finder variants (MSCOMP_AMD_XS=0,1,2 and the chain-walk kernel)
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/sqv
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
cd $R
SQA="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
SQB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
for v in ${1:-0 1 2}; do
  export MSCOMP_AMD_XS=$v
  timeout 300 rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d $D -o a_$v -- python tools/gpu_leg.py single:xpress_huff 2 > $D/a_$v.out 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d $D -o b_$v -- python tools/gpu_leg.py single:xpress_huff 2 > $D/b_$v.out 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$D/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "find" not in k and "sort" not in k and "links" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, cs in acc.items():
        print(f.split("/")[-1], k[:60], {c: round(v / n[(k, c)] / 1e6, 2) for c, v in cs.items()})
PY

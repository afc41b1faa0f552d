#!/bin/bash
# dev (GPU box, round 5): huff variants through the parity gate, then the SUBTRACTIVE probes of the two finders with their SQ counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/r5c
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $D/pytest.txt; cat $D/pytest.txt
lib() { if [ "$1" = base ]; then unset MSCOMP_AMD_LIB; else export MSCOMP_AMD_LIB=$R/build/libmscomp_amd_$1.so; fi; }
for v in base hd0 hd2; do lib $v; echo "== huff $v"; python tools/gpu_ab.py xpress_huff 1 config5 3 2>&1 | grep codec; done 2>&1 | tee $D/huff.txt
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
for v in base xf4 xf5; do lib $v; echo "== find $v"; python tools/gpu_ab.py xpress_huff 1 single 5 2>&1 | grep codec
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $D -o sq_xh_$v -- python tools/gpu_leg.py single:xpress_huff 2 > $D/sq_xh_$v.out 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $D -o clk_xh_$v -- python tools/gpu_leg.py single:xpress_huff 2 > $D/clk_xh_$v.out 2>&1
done 2>&1 | tee $D/find.txt
for v in base lz5 lz6; do lib $v; echo "== lznt1 $v"; python tools/gpu_ab.py lznt1 1 single 10 2>&1 | grep codec
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $D -o sq_lz_$v -- python tools/gpu_leg.py single:lznt1 3 > $D/sq_lz_$v.out 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $D -o clk_lz_$v -- python tools/gpu_leg.py single:lznt1 3 > $D/clk_lz_$v.out 2>&1
done 2>&1 | tee $D/lz.txt
python - <<PY
import csv, glob, collections, os
for f in sorted(glob.glob("$D/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "find" in k or "lznt1_chunk4" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(os.path.basename(f)[:28], k[:50], {c: round(sum(v) / len(v)) for c, v in cs.items()})
for f in sorted(glob.glob("$D/**/clk_*kernel_trace.csv", recursive=True)):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "find" in k or "lznt1_chunk4" in k:
            d[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        print(os.path.basename(f)[:28], k[:50], "avg_ns", round(sum(v) / len(v)))
PY
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|fclk\|mclk" | head -5
find $D -name "*.db" -delete; du -sh $D

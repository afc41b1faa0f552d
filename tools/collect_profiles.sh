#!/bin/bash
# Run on the GPU box (through gpurun): everything profiles/<tag>_* is made of. Output: gpurun_out/prof_<tag>/, condensed by
# tools/update_profiles.py into profiles/.
#   usage: bash tools/collect_profiles.sh r02 [quick]
#  1. (last, so that it carries the counters of this very run) the default bench line;
#  2. rocprofv3 --kernel-trace --stats of the HEADLINE command (bench.py --no-extra: every launch of the dominant kernel is the
#     headline workload, so its average duration is comparable with the HIP-event figure of the line) and of the full default command;
#  3. HBM traffic: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (TCC counters do not fit one pass), per leg (tools/gpu_leg.py);
#  4. SQ counters of the configs[4] legs and the single-GPU legs in two passes of 8 (issue / wait shares, instruction mix, LDS pipe, conflicts).
set -u
TAG=${1:-run}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/prof_$TAG
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
cd $R
rm -f $D/bench.json
LEGS="config5:lznt1 config5:xpress config5:xpress_huff single:lznt1 single:xpress single:xpress_huff"
[ -n "$QUICK" ] && LEGS="config5:lznt1"
for leg in $LEGS; do
  t=${leg/:/_}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D -o fetch_$t -- python tools/gpu_leg.py $leg 3 > $D/fetch_$t.out 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D -o write_$t -- python tools/gpu_leg.py $leg 3 > $D/write_$t.out 2>&1
done
SQA="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
SQB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS"
SLEGS="config5:lznt1 config5:xpress config5:xpress_huff single:lznt1 single:xpress single:xpress_huff decompress:xpress decompress:xpress_huff"
[ -n "$QUICK" ] && SLEGS="single:lznt1"
for leg in $SLEGS; do
  t=${leg/:/_}
  rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d $D -o sqa_$t -- python tools/gpu_leg.py $leg 3 > $D/sqa_$t.out 2>&1
  rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d $D -o sqb_$t -- python tools/gpu_leg.py $leg 3 > $D/sqb_$t.out 2>&1
done
# 5. (round 5) per codec: rocprofv3 --kernel-trace --stats of ONE configs[4] leg (profiles/<tag>_kernel_stats.txt gets one table per codec: every
#    roofline.frac of the bench line is then reproducible from profiles/ alone), and GRBM_GUI_ACTIVE of the same leg (the clock the kernels ran at:
#    cycles summed over the 8 XCDs / duration) -- what the issue rates of the SQ counters are divided by
for codec in lznt1 xpress xpress_huff; do
  [ -n "$QUICK" ] && [ $codec != lznt1 ] && continue
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o kt_config5_$codec -- python tools/gpu_leg.py config5:$codec 3 > $D/kt_config5_$codec.out 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $D -o clk_config5_$codec -- python tools/gpu_leg.py config5:$codec 3 > $D/clk_config5_$codec.out 2>&1
done
# 6. the drop-in entry with host pointers (PCIe-inclusive; never `value`): one 51 MB buffer per call, all three codecs, both directions
python tools/gpu_oneshot.py > $D/oneshot_host_pointers.txt 2>&1
[ -x tools/dev/issue_peak ] && tools/dev/issue_peak > $D/issue_peak.txt 2>&1
[ -x tools/dev/issue_peak2 ] && tools/dev/issue_peak2 > $D/issue_peak2.txt 2>&1
# the counter summaries go into profiles/ of THIS copy of the repository first: the bench line below attaches them (roofline.traffic / .secondary)
python tools/update_profiles.py $D $TAG > $D/update_on_box.log 2>&1
python bench.py 2> $D/bench.err | tail -1 > $D/bench.json
cp bench_extra.json $D/bench_extra.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o kt_head -- python bench.py --no-extra --no-cpu --steps 5 --warmup 1 2> $D/kt_head.err | tail -1 > $D/bench_headline_under_rocprof.json
[ -n "$QUICK" ] || rocprofv3 --kernel-trace --stats --output-format csv -d $D -o kt_full -- python bench.py --no-cpu --steps 4 --warmup 1 2> $D/kt_full.err | tail -1 > $D/bench_full_under_rocprof.json
find $D -name "*.db" -delete; find $D -name "*.csv" | wc -l
du -sh $D

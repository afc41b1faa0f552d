cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/pmc_dec; mkdir -p $D
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D -o a -- python tools/gpu_decomp.py 3 4 nofiles > $D/a.out 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d $D -o b -- python tools/gpu_decomp.py 3 4 nofiles > $D/b.out 2>&1
python tools/pmc_sum.py $D 2>&1 | grep -A12 "xhd_parse\|xpd_kernel\|lz_copy" | head -60
tail -3 $D/a.out

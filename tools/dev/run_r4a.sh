set -x
mkdir -p gpurun_out/r4a
python tools/gpu_lz4.py > gpurun_out/r4a/lz4.txt 2>&1
for m in 2 3; do MSCOMP_AB_LZ=$m python tools/gpu_ab.py lznt1 1 config5 3 >> gpurun_out/r4a/ab_lz.txt 2>&1; done
for e in 1 2 4; do MSCOMP_AB_EMIT=$e MSCOMP_AB_REPS=2,4,16 python tools/gpu_r8.py xpress >> gpurun_out/r4a/r8_emit.txt 2>&1; done
MSCOMP_AB_REPS=2,16 python tools/gpu_r8.py xpress_huff lznt1 >> gpurun_out/r4a/r8_other.txt 2>&1
cat gpurun_out/r4a/*.txt

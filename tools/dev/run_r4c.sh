set -x
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_hostbatch.py tests/test_gpu_hostptr.py -x -q > gpurun_out/r4c/tests.txt 2>&1
tail -5 gpurun_out/r4c/tests.txt
for mb in 8 16 32 64 128; do MSCOMP_AMD_HOST_BATCH_MB=$mb timeout 300 python tools/gpu_e2e.py >> gpurun_out/r4c/e2e.txt 2>&1; done
grep batch_mb gpurun_out/r4c/e2e.txt

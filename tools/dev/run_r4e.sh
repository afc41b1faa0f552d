mkdir -p gpurun_out/r4e
timeout 900 python -m pytest tests/test_gpu_hostbatch.py tests/test_gpu_hostptr.py tests/test_gpu_parity.py -x -q > gpurun_out/r4e/tests.txt 2>&1
tail -3 gpurun_out/r4e/tests.txt
for sl in 2 4 8; do for mb in 16 32; do echo "slots $sl mb $mb"; MSCOMP_AMD_HOST_SLOTS=$sl MSCOMP_AMD_HOST_BATCH_MB=$mb timeout 300 python tools/gpu_e2e.py 2>&1 | grep batch_mb; done; done
MSCOMP_AMD_HOST_TRACE=1 MSCOMP_AMD_HOST_SLOTS=4 MSCOMP_AMD_HOST_BATCH_MB=32 timeout 300 python tools/gpu_e2e.py > gpurun_out/r4e/trace4.txt 2>&1
grep -B9 "total" gpurun_out/r4e/trace4.txt | tail -44

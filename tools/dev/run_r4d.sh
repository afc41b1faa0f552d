mkdir -p gpurun_out/r4d
MSCOMP_AMD_HOST_TRACE=1 MSCOMP_AMD_HOST_BATCH_MB=32 timeout 300 python tools/gpu_e2e.py > gpurun_out/r4d/e2e_trace.txt 2>&1
tail -150 gpurun_out/r4d/e2e_trace.txt

for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_hostptr.py tests/test_gpu_hostbatch.py -x -q 2>&1 | tail -3; done
for c in 1 0; do MSCOMP_AMD_HOST_CLEAR=$c timeout 300 python tools/gpu_e2e.py 2>&1 | grep batch_mb; done

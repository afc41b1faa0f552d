for o in 0 1; do echo "order $o"; MSCOMP_AMD_HOST_ORDER=$o timeout 300 python tools/gpu_e2e.py 2>&1 | grep batch_mb; done
MSCOMP_AMD_HOST_TRACE=1 timeout 300 python tools/gpu_e2e.py 2>&1 | grep -B9 "total" | tail -10
timeout 900 python -m pytest tests/test_gpu_hostbatch.py -x -q 2>&1 | tail -2

for mb in 24 48 64 96 128 256; do MSCOMP_AMD_HOST_BATCH_MB=$mb timeout 300 python tools/gpu_e2e.py 2>&1 | grep batch_mb; done

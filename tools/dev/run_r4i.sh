timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
for i in 1 2; do timeout 300 python tools/gpu_e2e.py 2>&1 | grep batch_mb; done

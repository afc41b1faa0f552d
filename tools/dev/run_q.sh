timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "match_finder or corpus_golden or edge" 2>&1 | tail -5
for f in 1 2; do python tools/gpu_ab.py xpress_huff $f single 3 2>&1 | grep codec; done
for f in 1 2; do python tools/gpu_ab.py xpress_huff $f config5 2 2>&1 | grep codec; done

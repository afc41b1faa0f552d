for c in xpress_huff xpress; do python tools/dev/gpu_overlap.py $c 16 1 2 4 8 2>&1 | grep codec; python tools/dev/gpu_overlap.py $c 2 1 2 4 2>&1 | grep codec; done

"""Decompression at the scale of BASELINE configs[4] (16 x the corpus, 3.39 GB of output) on one GPU: the 3 239-unit legs of bench.py are bound by their
slowest unit (all units resident at once); this one shows the throughput when there is more work than residency.   python tools/dev/gpu_dec_config5.py <codec>"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus, sharding
import bench
codec = sys.argv[1]
fmt = m.FORMATS[codec]
ctx = m.Context(); dev = torch.device("cuda", 0)
cor = bench.Corpus(corpus, dev)
off, ln, desc = bench.config5_units(cor, fmt)
d_in = cor.device_range(0, int(off[-1] + ln[-1]))
r = bench.decompress_leg(m, ctx, fmt, d_in, off, ln, desc, 2, sharding)
print(json.dumps({"leg": "decompress config5:" + codec, **r}))

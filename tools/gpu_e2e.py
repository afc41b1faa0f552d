"""Dev helper (GPU box): the end-to-end (host pointers in / out) leg of bench.py alone, for one batch size (MSCOMP_AMD_HOST_BATCH_MB)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
import bench
cor = bench.Corpus(corpus, torch.device("cuda", 0))
out = {"batch_mb": os.environ.get("MSCOMP_AMD_HOST_BATCH_MB")}
for codec, wl in (("lznt1", "mozilla"), ("lznt1", "silesia_files"), ("xpress", "silesia_units64k"), ("xpress_huff", "silesia_files")):
    b, o, l, d = bench.single_gpu_workload(cor, wl)
    r = bench.end_to_end_leg(m, m.FORMATS[codec], b, o, l, d)
    out[codec + ":" + wl] = (r["MB_per_s"], r["ms"])
print(json.dumps(out))

"""Run ONE leg of bench.py (for rocprofv3 --pmc passes: every kernel launch then belongs to one known workload).
    python tools/gpu_leg.py config5:<codec> | single:<codec> | decompress:<codec>   [steps]
config5 = BASELINE configs[4] on this GPU (all 16 replicas), single = configs[1..3] as round 1 measured them."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus, sharding
import bench
kind, codec = sys.argv[1].split(":")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = m.Context()
dev = torch.device("cuda", 0)
cor = bench.Corpus(corpus, dev)
fmt = m.FORMATS[codec]
if kind == "config5":
    r = bench.sharded_leg(m, ctx, cor, fmt, 0, 1, steps, 1, sharding, None)
elif kind == "single":
    wl = {"lznt1": "mozilla", "xpress": "silesia_units64k", "xpress_huff": "silesia_files"}[codec]
    b, o, l, d = bench.single_gpu_workload(cor, wl)
    j = bench.Job(m, ctx, fmt, b, o, l)
    t, p = bench.timed(j, steps, 1, sharding)
    r = {"MB_per_s": round(j.in_bytes * steps / t / 1e6, 1), "workload": d, "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in p.items()}}
else:
    wl = {"lznt1": "mozilla", "xpress": "silesia_units64k", "xpress_huff": "silesia_units64k"}[codec]
    b, o, l, d = bench.single_gpu_workload(cor, wl)
    r = bench.decompress_leg(m, ctx, fmt, b, o, l, d, steps, sharding)
print(json.dumps({"leg": sys.argv[1], "steps": steps, **{k: v for k, v in r.items() if k != "roofline"}}))

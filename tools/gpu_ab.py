"""Dev helper (GPU box): one codec on BASELINE configs[4] (or the 12 files alone) with the finder variant given, per-kernel ms.
    python tools/gpu_ab.py <codec> <finder mode: 1 default | 2 previous kernel> [config5|single] [steps]
Also checks the packed output of replica 0 against tests/golden/corpus_full.json when run on the synthetic corpus (single)."""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")   # (the kernel switches: csrc/api.hip test_hooks_on)
import json, os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus, sharding
import bench
codec = sys.argv[1]; mode = int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "config5"; steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ctx = m.Context()
ctx.lib.mscomp_amd_debug_set_finder(mode)
if os.environ.get('MSCOMP_AB_LZ'): ctx.lib.mscomp_amd_debug_set_lznt1(int(os.environ['MSCOMP_AB_LZ']))
dev = torch.device("cuda", 0)
cor = bench.Corpus(corpus, dev)
fmt = m.FORMATS[codec]
if kind == "config5":
    r = bench.sharded_leg(m, ctx, cor, fmt, 0, 1, steps, 1, sharding, None)
    out = {"MB_per_s": r["MB_per_s"], "ms_per_step": r["ms_per_step"], "CR": r["compression_ratio"], "kernels": r["roofline"]["kernels_ms_per_step"]}
else:
    wl = {"lznt1": "mozilla", "xpress": "silesia_units64k", "xpress_huff": "silesia_files"}[codec]
    b, o, l, d = bench.single_gpu_workload(cor, wl)
    j = bench.Job(m, ctx, fmt, b, o, l)
    t, p = bench.timed(j, steps, 1, sharding)
    out = {"MB_per_s": round(j.in_bytes * steps / t / 1e6, 1), "ms_per_step": round(t / steps * 1e3, 3), "CR": round(j.out_bytes() / j.in_bytes, 4),
           "kernels": {k: round(v[0] / steps, 4) for k, v in sorted(p.items(), key=lambda kv: -kv[1][0])}}
print(json.dumps({"codec": codec, "finder": mode, "kind": kind, "tile_env": os.environ.get("MSCOMP_AMD_XF2_TILE"), **out}))

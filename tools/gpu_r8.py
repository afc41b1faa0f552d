"""Dev helper (GPU box): what one rank of an 8-GPU run holds (2 replicas, 424 MB), per-kernel ms next to an eighth of the whole job's."""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")   # (the kernel switches: csrc/api.hip test_hooks_on)
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus, sharding
import bench
ctx = m.Context()
if os.environ.get('MSCOMP_AB_FINDER'): ctx.lib.mscomp_amd_debug_set_finder(int(os.environ['MSCOMP_AB_FINDER']))
if os.environ.get('MSCOMP_AB_EMIT'): ctx.lib.mscomp_amd_debug_set_xpress_emit(int(os.environ['MSCOMP_AB_EMIT']))
cor = bench.Corpus(corpus, torch.device("cuda", 0))
for codec in sys.argv[1:] or ["lznt1", "xpress", "xpress_huff"]:
    f = m.FORMATS[codec]
    off, ln, _ = bench.config5_units(cor, f)
    for reps in [int(x) for x in os.environ.get('MSCOMP_AB_REPS', '2,16').split(',')]:
        nu = len(ln) // 16 * reps
        j = bench.Job(m, ctx, f, cor.device_range(0, reps * cor.total), off[:nu], ln[:nu])
        t, p = bench.timed(j, 3, 1, sharding)
        print(codec, reps, "replicas: %.3f ms/step = %.3f ms per replica" % (t / 3 * 1e3, t / 3 * 1e3 / reps), {k: round(v[0] / 3 / reps, 4) for k, v in sorted(p.items(), key=lambda kv: -kv[1][0])}, flush=True)
        j.close()

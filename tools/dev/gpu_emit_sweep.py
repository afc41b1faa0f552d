"""dev (GPU box): the Xpress pass (links + lazy finder + emit) over the first N 64 KiB units of the replicated corpus for N around the emit kernels'
crossover, one wave per unit (mode 1) against four waves per unit (mode 2) -- VERDICT r05 item 3a: is `n_units <= 1024 ? 2 : 1` the right constant?"""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus, sharding
import bench
ctx = m.Context()
cor = bench.Corpus(corpus, torch.device("cuda", 0))
off, ln, _ = bench.config5_units(cor, 3)
per = len(ln) // 16
d_in = cor.device_range(0, 4 * cor.total)
rng = np.random.default_rng(1)
for n in (128, 256, 512, 768, 1024, 1536, 2048, 3239, 6478, 12956):
    # units spread over the files like a shard is (every k-th unit), not the first file's only
    idx = np.sort(rng.choice(4 * per, size=n, replace=False))
    row = []
    for mode in (1, 2):
        ctx.lib.mscomp_amd_debug_set_xpress_emit(mode)
        j = bench.Job(m, ctx, 3, d_in, off[idx], ln[idx])
        t, p = bench.timed(j, 6, 2, sharding)
        k = {k: v[0] / 6 for k, v in p.items()}
        row.append((t / 6 * 1e3, max((v for kk, v in k.items() if "emit" in kk), default=0.0)))
        j.close()
    print("%6d units: one wave %.3f ms (emit %.3f) | four waves %.3f ms (emit %.3f)" % (n, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
ctx.lib.mscomp_amd_debug_set_xpress_emit(0)

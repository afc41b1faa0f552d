"""Dev experiment (GPU box): does running a job as K parts on K streams (own context each), so that the latency-bound serial kernels of
one part run beside the throughput-bound finder of another, beat one stream?   python tools/dev/gpu_overlap.py <codec> [replicas] [K ...]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus, sharding
import bench
codec = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
Ks = [int(x) for x in sys.argv[3:]] or [1, 2, 4]
dev = torch.device("cuda", 0)
cor = bench.Corpus(corpus, dev)
fmt = m.FORMATS[codec]
off, ln, _ = bench.config5_units(cor, fmt)
nu = len(ln) // 16 * reps
off, ln = off[:nu], ln[:nu]
total = int(ln.sum())
for K in Ks:
    jobs = []
    for r in range(K):
        s, e, g0, g1, my_off, my_len = sharding.shard_job(off, ln, K, r)
        st = torch.cuda.Stream()
        ctx = m.Context(device=0, stream=st)
        jobs.append((bench.Job(m, ctx, fmt, cor.device_range(g0, g1), my_off, my_len), ctx, st))
    def step():
        for j, c, s in jobs:
            j.step()
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 4
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(json.dumps({"codec": codec, "replicas": reps, "parts": K, "ms_per_pass": round(dt * 1e3, 3), "GB_per_s": round(total / dt / 1e9, 2)}), flush=True)
    for j, c, s in jobs:
        j.close(); c.close()
    del jobs
    torch.cuda.empty_cache()

"""Dev helper (GPU box): quick throughput of one codec on a corpus member, with per-kernel times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus

fmt = int(sys.argv[1]) if len(sys.argv) > 1 else 2
name = sys.argv[2] if len(sys.argv) > 2 else "mozilla"
unit = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # 0: whole file is one unit, else split in units of this size
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
data = corpus.by_name(name)
n = len(data)
if unit:
    lens = [min(unit, n - o) for o in range(0, n, unit)]
    in_off = np.arange(0, n, unit, dtype=np.uint64)
else:
    lens = [n]; in_off = np.zeros(1, dtype=np.uint64)
caps = [m.max_compressed_size(fmt, l) + 2 for l in lens]
out_off, out_total = m.pack_offsets(caps)
ctx = m.Context()
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
d_out = torch.empty(out_total + 16, dtype=torch.uint8, device=dev)
d_len = torch.zeros(len(lens), dtype=torch.int64, device=dev)
d_st = torch.zeros(len(lens), dtype=torch.int32, device=dev)
plan = m.Plan(ctx, fmt, in_off, lens, out_off, caps)
plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
ctx.profile_enable(True)
t0 = time.time()
for _ in range(reps):
    plan.execute(d_in, d_out, d_len, d_st)
torch.cuda.synchronize()
dt = (time.time() - t0) / reps
prof = ctx.profile_read()
out_bytes = int(d_len.sum().item())
print("%s fmt %d: %d B in %d units, %.3f ms/pass, %.2f GB/s in, CR %.3f, status ok=%s" % (name, fmt, n, len(lens), dt * 1e3, n / dt / 1e9, out_bytes / n, bool((d_st == 0).all().item())))
for k, (ms, cnt) in prof.items():
    print("   %-28s %8.3f ms/launch  x%d" % (k, ms / cnt, cnt))

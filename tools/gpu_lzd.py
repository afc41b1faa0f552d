"""LZNT1 decompression timing on the bench corpus (mozilla as one unit, and Silesia as 12 units): per-kernel ms via the
library's event profiler."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import ms_compress_amd as m
from ms_compress_amd import corpus

def run(units, label):
    ctx = m.Context(); dev = torch.device("cuda", 0)
    lens = [len(u) for u in units]
    in_off, in_total = m.pack_offsets(lens)
    caps = [m.max_compressed_size(2, n) + 2 for n in lens]
    out_off, out_total = m.pack_offsets(caps)
    blob = np.zeros(in_total + 16, dtype=np.uint8)
    for o, u in zip(in_off, units): blob[int(o):int(o) + len(u)] = u
    d_in = torch.from_numpy(blob).to(dev); d_c = torch.zeros(out_total + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(len(units), dtype=torch.int64, device=dev); d_st = torch.zeros(len(units), dtype=torch.int32, device=dev)
    p = m.Plan(ctx, 2, in_off, lens, out_off, caps); p.execute(d_in, d_c, d_len, d_st); torch.cuda.synchronize(); p.close()
    clens = [int(x) for x in d_len.cpu()]
    back_off, back_total = m.pack_offsets(lens)
    d_back = torch.zeros(back_total + 16, dtype=torch.uint8, device=dev)
    q = m.Plan(ctx, 2, out_off, clens, back_off, lens, decompress=True)
    d_len2 = torch.zeros(len(units), dtype=torch.int64, device=dev); d_st2 = torch.zeros(len(units), dtype=torch.int32, device=dev)
    for _ in range(3): q.execute(d_c, d_back, d_len2, d_st2)
    torch.cuda.synchronize()
    ok = all(int(x) == 0 for x in d_st2.cpu()) and [int(x) for x in d_len2.cpu()] == lens
    for o, n in zip(back_off, lens): ok = ok and torch.equal(d_back[int(o):int(o) + n], d_in[int(in_off[list(back_off).index(o)]):int(in_off[list(back_off).index(o)]) + n])
    t0 = time.perf_counter(); K = 20
    for _ in range(K): q.execute(d_c, d_back, d_len2, d_st2)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    ctx.profile_enable(True)
    for _ in range(5): q.execute(d_c, d_back, d_len2, d_st2)
    prof = ctx.profile_read(); ctx.profile_enable(False)
    ctx.lib.mscomp_amd_debug_lzd_walked(ctx._h); q.execute(d_c, d_back, d_len2, d_st2)
    print("   walked segments:", ctx.lib.mscomp_amd_debug_lzd_walked(ctx._h), "of", sum((c + 49151) // 49152 for c in clens))
    print(label, "ok" if ok else "MISMATCH", "%.3f ms/pass" % (dt * 1e3), "%.1f GB/s out" % (sum(lens) / dt / 1e9), "comp %d" % sum(clens))
    for k, (ms, c) in prof.items(): print("   %-24s %.4f ms" % (k, ms / c))
    q.close(); ctx.close()

if len(sys.argv) > 1 and sys.argv[1] == "files":
    for i in range(12): run([corpus.file_bytes(i)], corpus.NAMES[i])
    sys.exit(0)
moz = corpus.file_bytes(corpus.NAMES.index("mozilla"), 51_220_480)
run([moz], "mozilla x1 unit")
run([corpus.file_bytes(i) for i in range(12)], "silesia 12 units")
units = []
for i in range(12):
    f = corpus.file_bytes(i)
    units += [f[k:k + 65536] for k in range(0, len(f), 65536)]
run(units, "silesia %d units of 64 KiB" % len(units))

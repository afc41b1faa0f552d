"""Decompression timing on the bench corpus: for each format, (a) the bench workload of the compressor (LZNT1: mozilla as one
unit; Xpress: 3239 units of 64 KiB; Xpress+Huffman: 12 files), (b) 3239 units of 64 KiB. Compressed on the GPU first, decoded
back and compared; per-kernel ms from the library's event profiler."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import ms_compress_amd as m
from ms_compress_amd import corpus

def run(fmt, units, label, reps=10):
    ctx = m.Context(); dev = torch.device("cuda", 0)
    lens = [len(u) for u in units]
    in_off, in_total = m.pack_offsets(lens)
    caps = [m.max_compressed_size(fmt, n) + 2 for n in lens]
    out_off, out_total = m.pack_offsets(caps)
    blob = np.zeros(in_total + 16, dtype=np.uint8)
    for o, u in zip(in_off, units): blob[int(o):int(o) + len(u)] = u
    d_in = torch.from_numpy(blob).to(dev); d_c = torch.zeros(out_total + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(len(units), dtype=torch.int64, device=dev); d_st = torch.zeros(len(units), dtype=torch.int32, device=dev)
    p = m.Plan(ctx, fmt, in_off, lens, out_off, caps); p.execute(d_in, d_c, d_len, d_st); torch.cuda.synchronize(); p.close()
    clens = [int(x) for x in d_len.cpu()]
    d_back = torch.zeros(in_total + 16, dtype=torch.uint8, device=dev)
    q = m.Plan(ctx, fmt, out_off, clens, in_off, lens, decompress=True)
    d_len2 = torch.zeros(len(units), dtype=torch.int64, device=dev); d_st2 = torch.zeros(len(units), dtype=torch.int32, device=dev)
    q.execute(d_c, d_back, d_len2, d_st2); torch.cuda.synchronize()
    ok = all(int(x) == 0 for x in d_st2.cpu()) and [int(x) for x in d_len2.cpu()] == lens
    h_back = d_back.cpu().numpy()
    for o, n in zip(in_off, lens): ok = ok and np.array_equal(h_back[int(o):int(o) + n], blob[int(o):int(o) + n])
    t0 = time.perf_counter()
    for _ in range(reps): q.execute(d_c, d_back, d_len2, d_st2)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    ctx.profile_enable(True)
    for _ in range(3): q.execute(d_c, d_back, d_len2, d_st2)
    prof = ctx.profile_read(); ctx.profile_enable(False)
    print("fmt %d %-34s %s %9.3f ms/pass %7.2f GB/s out (%d -> %d B)" % (fmt, label, "ok" if ok else "MISMATCH", dt * 1e3, sum(lens) / dt / 1e9, sum(clens), sum(lens)))
    for k, (ms, c) in prof.items(): print("       %-24s %.4f ms" % (k, ms / c))
    q.close(); ctx.close()

files = [corpus.file_bytes(i) for i in range(12)]
u64k = []
for f in files: u64k += [f[k:k + 65536] for k in range(0, len(f), 65536)]
moz = corpus.file_bytes(corpus.NAMES.index("mozilla"), 51_220_480)
which = sys.argv[1:] or ["2", "3", "4"]
nofiles = "nofiles" in which
if "2" in which:
    run(2, [moz], "mozilla as one unit"); run(2, u64k, "%d units of 64 KiB" % len(u64k))
if "3" in which:
    run(3, u64k, "%d units of 64 KiB" % len(u64k)); (None if nofiles else run(3, files, "12 files, one stream each", reps=1))
if "4" in which:
    run(4, u64k, "%d units of 64 KiB" % len(u64k)); (None if nofiles else run(4, files, "12 files", reps=1))

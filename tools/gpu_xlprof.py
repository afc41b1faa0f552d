import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
lib = m.load_library()
data = np.concatenate([corpus.by_name(x) for x in corpus.NAMES]); n = len(data)
lens = [min(65536, n - o) for o in range(0, n, 65536)]; in_off = np.arange(0, n, 65536, dtype=np.uint64)
caps = [m.max_compressed_size(3, l) + 2 for l in lens]; out_off, tot = m.pack_offsets(caps)
ctx = m.Context(); dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev); d_out = torch.empty(tot + 16, dtype=torch.uint8, device=dev)
d_len = torch.zeros(len(lens), dtype=torch.int64, device=dev); d_st = torch.zeros(len(lens), dtype=torch.int32, device=dev)
plan = m.Plan(ctx, 3, in_off, lens, out_off, caps)
plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)(); lib.mscomp_amd_debug_xl_prof(buf)
plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize(); lib.mscomp_amd_debug_xl_prof(buf)
nc = buf[3]
print("links per chunk: consumer busy %.0f, producer busy %.0f, whole block %.0f cycles (%d chunks)" % (buf[0] / nc, buf[1] / nc, buf[2] / nc, nc))

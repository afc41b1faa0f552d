import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
lib = m.load_library()
data = np.concatenate([corpus.by_name(x) for x in corpus.NAMES]); n = len(data)
ctx = m.Context(); dev = torch.device("cuda", 0)
cap = m.max_compressed_size(4, n) + 2
d_in = torch.from_numpy(data).to(dev); d_out = torch.empty(cap + 16, dtype=torch.uint8, device=dev)
d_len = torch.zeros(1, dtype=torch.int64, device=dev); d_st = torch.zeros(1, dtype=torch.int32, device=dev)
plan = m.Plan(ctx, 4, [0], [n], [0], [cap])
plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)(); lib.mscomp_amd_debug_xp_prof(buf)
plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize(); lib.mscomp_amd_debug_xp_prof(buf)
nc = (n + 65535) // 65536
print("parse cycles per chunk (avg / max): speculative %.0f / %d, seams %.0f / %d, histogram %.0f / %d" % (buf[0]/nc, buf[4], buf[1]/nc, buf[5], buf[2]/nc, buf[6]))

"""dev: counters of the lazy Xpress+Huffman finder (a -DXHZ_PROFILE build through MSCOMP_AMD_LIB, MSCOMP_AMD_XH_LAZY=1): python tools/dev/gpu_xhz.py mozilla dickens ..."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
lib = m.load_library()
ctx = m.Context(); dev = torch.device("cuda", 0)
for name in sys.argv[1:] or ["mozilla"]:
    data = corpus.by_name(name); n = len(data); cap = m.max_compressed_size(4, n) + 2
    d_in = torch.from_numpy(data).to(dev); d_out = torch.empty(cap + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(1, dtype=torch.int64, device=dev); d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    plan = m.Plan(ctx, 4, [0], [n], [0], [cap])
    buf = (C.c_ulonglong * 8)(); lib.mscomp_amd_debug_xhz_prof(buf)
    plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize(); lib.mscomp_amd_debug_xhz_prof(buf)
    nch = (n + 65535) // 65536
    print("%-8s per chunk: wave steps %.0f | claims %.0f (%.3f per byte) | near candidates %.0f, far %.0f (far lane-steps %.0f) | extension passes %.0f | cycles: staging %.0f, walks %.0f" %
          (name, buf[0] / nch, buf[1] / nch, buf[1] / n, buf[2] / nch, buf[3] / nch, buf[6] / nch, buf[7] / nch, buf[4] / nch, buf[5] / nch))
    plan.close()

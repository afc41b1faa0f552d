import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
lib = m.load_library()
names = ["A load", "B1 rank", "B2/3 scan+scatter", "C1 self-scan", "C2 walk+finish", "C3 emit", "D out"]
for name in sys.argv[1:]:
    data = corpus.by_name(name); n = len(data)
    ctx = m.Context(); dev = torch.device("cuda", 0)
    cap = m.max_compressed_size(2, n) + 2
    d_in = torch.from_numpy(data).to(dev); d_out = torch.empty(cap + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(1, dtype=torch.int64, device=dev); d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    plan = m.Plan(ctx, 2, [0], [n], [0], [cap])
    plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.mscomp_amd_debug_lz_prof(buf, 1)
    plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
    lib.mscomp_amd_debug_lz_prof(buf, 1)
    nch = (n + 4095) // 4096
    tot = sum(buf[:7])
    print("   per chunk: asm-walk cycles %.0f in %.0f calls, taken matches %.0f" % (buf[12] / nch, buf[13] / nch, buf[14] / nch))
    print("   per chunk: finish steps %.0f, unresolved %.0f, finishes %.0f (>16 cand: %.0f, >64: %.0f), finishes with match %.0f, cycles in finish loops %.0f" % tuple(buf[i] / nch for i in (9, 10, 11, 12, 13, 14, 15)))
    print(name, "chunks", nch, "avg cycles/chunk %.0f" % (tot / nch), " | ".join("%s %.0f (%.0f%%)" % (nm, buf[i] / nch, 100.0 * buf[i] / tot) for i, nm in enumerate(names)))

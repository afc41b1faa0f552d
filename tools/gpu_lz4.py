"""Dev: LZNT1 chunk kernels (1 = one wave per chunk, 2 = four waves per chunk): parity on the edge units + speed on mozilla."""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")   # (the kernel switches: csrc/api.hip test_hooks_on)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
import cases
from oracle import loader
orc = loader
lib = m.load_library()
units = cases.edge_cases()
for mode in (1, 2):
    lib.mscomp_amd_debug_set_lznt1(mode)
    got, st = m.compress_units(2, units)
    bad = 0
    for u, g, s in zip(units, got, st):
        es, exp = orc.oracle_compress(2, u)
        bad += (s != 0) or (g != exp)
    data = np.concatenate([corpus.by_name(x) for x in (sys.argv[1:] or ["mozilla"])]); n = len(data)
    ctx = m.Context(); dev = torch.device("cuda", 0)
    cap = m.max_compressed_size(2, n) + 2
    d_in = torch.from_numpy(data).to(dev); d_out = torch.empty(cap + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(1, dtype=torch.int64, device=dev); d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    plan = m.Plan(ctx, 2, [0], [n], [0], [cap])
    plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): plan.execute(d_in, d_out, d_len, d_st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    import hashlib
    print(("lznt1 mode %d: edge units %d bad %d; " + "+".join(sys.argv[1:] or ["mozilla"]) + " %.3f ms per pass (%.1f GB/s), out %d sha %s") % (mode, len(units), bad, dt * 1e3, n / dt / 1e9, int(d_len[0]), hashlib.sha256(bytes(d_out[: int(d_len[0])].cpu().numpy())).hexdigest()[:16]))
lib.mscomp_amd_debug_set_lznt1(0)
if hasattr(lib, "mscomp_amd_debug_lz4_prof") or True:
    import ctypes as C
    try:
        lib.mscomp_amd_debug_set_lznt1(2)
        buf = (C.c_ulonglong * 8)(); lib.mscomp_amd_debug_lz4_prof(buf)
        plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize(); lib.mscomp_amd_debug_lz4_prof(buf)
        nch = (n + 4095) // 4096
        print("wave 0 cycles per chunk: load %.0f | sort %.0f | parse(seg 0) %.0f | seam %.0f | wait %.0f | cascade+scan %.0f | emit %.0f" % tuple(buf[i] / nch for i in range(7)))
    except AttributeError:
        pass
    lib.mscomp_amd_debug_set_lznt1(0)

"""Dev: one big Xpress stream (whole file as ONE unit): emit kernel time, one wave vs four waves per unit."""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")   # (the kernel switches: csrc/api.hip test_hooks_on)
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
lib = m.load_library()
for name in sys.argv[1:] or ["mozilla"]:
    data = corpus.by_name(name); n = len(data)
    ctx = m.Context(); dev = torch.device("cuda", 0)
    cap = m.max_compressed_size(3, n) + 2
    d_in = torch.from_numpy(data).to(dev); d_out = torch.empty(cap + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(1, dtype=torch.int64, device=dev); d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    plan = m.Plan(ctx, 3, [0], [n], [0], [cap])
    outs = []
    for mode in (1, 2, 3, 4):
        lib.mscomp_amd_debug_set_xpress_emit(mode)
        plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): plan.execute(d_in, d_out, d_len, d_st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        outs.append(bytes(d_out[: int(d_len[0])].cpu().numpy()))
        print(name, n, "B as ONE stream, emit mode", mode, ": %.2f ms per pass (%.2f GB/s), out %d" % (dt * 1e3, n / dt / 1e9, int(d_len[0])))
    print("   identical:", len(set(outs)) == 1)
    lib.mscomp_amd_debug_set_xpress_emit(0)

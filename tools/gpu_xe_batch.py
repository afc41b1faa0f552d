"""Dev: the bench's Xpress batch (corpus cut into 64 KiB units) under every parse/emit kernel."""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")   # (the kernel switches: csrc/api.hip test_hooks_on)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
lib = m.load_library()
data = np.concatenate([corpus.by_name(x) for x in corpus.NAMES]); n = len(data)
lens = [min(65536, n - o) for o in range(0, n, 65536)]; in_off = np.arange(0, n, 65536, dtype=np.uint64)
caps = [m.max_compressed_size(3, l) + 2 for l in lens]; out_off, tot = m.pack_offsets(caps)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev); d_out = torch.empty(tot + 16, dtype=torch.uint8, device=dev)
d_len = torch.zeros(len(lens), dtype=torch.int64, device=dev); d_st = torch.zeros(len(lens), dtype=torch.int32, device=dev)
ref = None
for mode in (1, 2, 4):
    lib.mscomp_amd_debug_set_xpress_emit(mode)
    ctx = m.Context(); plan = m.Plan(ctx, 3, in_off, lens, out_off, caps)
    plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): plan.execute(d_in, d_out, d_len, d_st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    out = d_out.cpu().numpy().tobytes(); ln = d_len.cpu().numpy().tobytes()
    if ref is None: ref = (out, ln)
    print("mode %d: %.2f ms per pass (%.1f GB/s) %s" % (mode, dt * 1e3, n / dt / 1e9, "identical" if (out, ln) == ref else "DIFFERENT"))
lib.mscomp_amd_debug_set_xpress_emit(0)

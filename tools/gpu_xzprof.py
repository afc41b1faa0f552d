"""Dev helper (GPU box): the lazy Xpress finder per corpus member (64 KiB units): kernel ms next to the all-positions finder and,
with a -DXZ_PROFILE build, its counters (state-machine steps, positions claimed, candidate compares, matches stored)."""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")   # (the kernel switches: csrc/api.hip test_hooks_on)
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ms_compress_amd as m
from ms_compress_amd import corpus
ctx = m.Context(); lib = ctx.lib
has_prof = hasattr(lib, "mscomp_amd_debug_xz_prof")
dev = torch.device("cuda", 0)
names = sys.argv[1:] or corpus.NAMES
for name in names:
    data = corpus.by_name(name); n = len(data)
    lens = [min(65536, n - o) for o in range(0, n, 65536)]
    in_off = np.arange(0, n, 65536, dtype=np.uint64)
    caps = [m.max_compressed_size(3, l) + 2 for l in lens]
    out_off, out_total = m.pack_offsets(caps)
    d_in = torch.from_numpy(data).to(dev); d_out = torch.empty(out_total + 16, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(len(lens), dtype=torch.int64, device=dev); d_st = torch.zeros(len(lens), dtype=torch.int32, device=dev)
    res = {}
    for mode in (2, 1):
        lib.mscomp_amd_debug_set_finder(mode)
        plan = m.Plan(ctx, 3, in_off, lens, out_off, caps)
        plan.execute(d_in, d_out, d_len, d_st); torch.cuda.synchronize()
        if has_prof:
            buf = (C.c_ulonglong * 8)(); lib.mscomp_amd_debug_xz_prof(buf)

        ctx.profile_enable(True)
        for _ in range(3): plan.execute(d_in, d_out, d_len, d_st)
        torch.cuda.synchronize()
        prof = ctx.profile_read(); ctx.profile_enable(False)
        res[mode] = {k: v[0] / v[1] for k, v in prof.items()}
        if has_prof and mode == 1:
            lib.mscomp_amd_debug_xz_prof(buf); c = [x / 3 for x in buf]
            tot = (c[4] + c[5] + c[6]) or 1
            res["cnt"] = "wave-steps/64pos %.2f  claimed/pos %.3f  compares/pos %.2f  matches/pos %.3f  lane occupancy %.2f | wave cycles: staging %.2f walks %.2f waiting for the block %.2f (%.0f per 64 pos)" % (
                c[0] / n, c[1] / n, c[2] / n, c[3] / n, (c[2] + c[1]) / max(c[0], 1), c[4] / tot, c[5] / tot, c[6] / tot, tot / (n / 64))
        plan.close()
    lib.mscomp_amd_debug_set_finder(1)
    print("%-8s %9d B %5d units: find %.3f ms  lazy2 %.3f ms  (emit %.3f / %.3f) %s" % (name, n, len(lens), res[2].get("xp_find_kernel", 0), res[1].get("xp_lazy2_kernel", 0),
          res[2].get("xpress_emit_kernel", 0), res[1].get("xpress_emit_kernel", 0), res.get("cnt", "")), flush=True)

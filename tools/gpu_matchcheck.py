"""Dev helper (GPU box): compare the HIP match finder's per-position table with the oracle's (Xpress / XH windows)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cases
import ms_compress_amd as m
from ms_compress_amd import corpus
from oracle import loader

def check(ctx, data, max_off, clip):
    n = len(data)
    if n == 0:
        return 0
    lib = m.load_library(); orc = loader.load_oracle()
    d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    gl = np.zeros(n, dtype=np.uint16); go = np.zeros(n, dtype=np.uint16)
    st = lib.mscomp_amd_debug_xpress_matches(ctx._h, C.c_void_p(d.data_ptr()), n, max_off, clip, gl.ctypes.data, go.ctypes.data)
    assert st == 0, st
    ol = np.zeros(n, dtype=np.uint32); oo = np.zeros(n, dtype=np.uint32)
    orc.orc_xpress_match_table(data, n, max_off, ol.ctypes.data, oo.ctypes.data)
    # oracle: len (2 = none), uncapped. expected GPU: len3 = min(len,48)-3, off ; clip: positions with <3 bytes left in chunk -> none
    exp_l = np.where(ol >= 3, np.minimum(ol, 48) - 3, 0).astype(np.uint16)
    exp_o = np.where(ol >= 3, oo, 0).astype(np.uint16)
    if clip:
        pos = np.arange(n); rem = np.minimum((pos // 65536 + 1) * 65536, n) - pos
        exp_l[rem < 3] = 0; exp_o[rem < 3] = 0
    bad = np.nonzero((exp_l != gl) | (exp_o != go))[0]
    if len(bad):
        i = int(bad[0])
        print("  MISMATCH n=%d max_off=%x: %d bad, first @%d: gpu (len3 %d off %d) oracle (len %d off %d)" % (n, max_off, len(bad), i, gl[i], go[i], ol[i], oo[i]))
    return len(bad)

ctx = m.Context()
units = cases.edge_cases() + [corpus.file_bytes(i, 300_000).tobytes() for i in range(12)] + [cases.mixed_buffer()]
tot = 0
for mo, clip in ((0x2000, 0), (0xFFFF, 1)):
    b = sum(1 for u in units if check(ctx, u, mo, clip))
    print("max_off %x: units %d bad %d" % (mo, len(units), b)); tot += b

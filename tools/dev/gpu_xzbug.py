"""dev (GPU box): the units of tools/dev/realbug/*.in (64 KiB pieces of real files on which the lazy Xpress finder gave other bytes than the
reference) as batches of N copies, K times, under the library given by MSCOMP_AMD_LIB: how many copies differ from the oracle's bytes?"""
import os as _os; _os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ms_compress_amd as m
from oracle import loader
import cases, random
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = m.Context()
units = [(os.path.basename(f), open(f, "rb").read()) for f in sorted(glob.glob(os.path.join(ROOT, "tools", "dev", "realbug", "*.in")))]
rnd = random.Random(77)
text = b"".join(rnd.choice(cases.WORDS) for _ in range(30000))
units += [("periodic+mutations", cases.periodic_with_mutations()), ("periodic+mutations p=1000", cases.periodic_with_mutations(period=1000, seed=3, gap=(120, 300)))]
units += [("few_distances", cases.few_distances()), ("few_distances 2", cases.few_distances(dists=(3756, 7426), seed=12)), ("few_distances 8", cases.few_distances(dists=(3756, 7426, 3704, 486, 524, 3680, 503, 7576), seed=13, run=(113, 200)))]
units += [("few_distances long", cases.few_distances(seed=21, run=(600, 3000))), ("few_distances long2", cases.few_distances(dists=(3756, 7426), seed=22, run=(300, 2500))),
          ("few_distances mid", cases.few_distances(seed=23, run=(200, 1200))), ("periodic long gaps", cases.periodic_with_mutations(seed=5, gap=(500, 3000)))]
units += [("text65536", text[:65536]), ("lz65536", cases.family("lz", 65536, rnd)), ("two65536", cases.family("two", 65536, rnd))]
for finder in ((1,) if os.environ.get('XZBUG_F1') else (1, 2)):
    ctx.lib.mscomp_amd_debug_set_finder(finder)
    for name, u in units:
        exp = loader.oracle_compress(3, u)[1]
        bad = []
        for r in range(reps):
            got, st = m.compress_units(3, [u] * copies, ctx=ctx)
            bad.append(sum(1 for g in got if g != exp))
        print("finder %d %-22s copies %d: differing per run %s" % (finder, name, copies, bad), flush=True)
ctx.lib.mscomp_amd_debug_set_finder(1)

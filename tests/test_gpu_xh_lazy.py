"""The lazy Xpress+Huffman match finder (csrc/xhuff_lazy.hip, round 5: the chunk's chain links in LDS, candidate bytes gathered from L2) is a
measurement mode -- slower than the all-positions finder and off by default -- but it is a second, independent way to the same bytes:
this test keeps it exact. Both measurement modes live in the DEVELOPMENT flavour of the library only (libmscomp_amd_dev.so, `make dev`; the product
library has one finder per codec): the subprocesses load it through MSCOMP_AMD_LIB; the switch (MSCOMP_AMD_XH_LAZY=1) is read once per process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SNIPPET = r"""
import hashlib, sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import ms_compress_amd as m
from ms_compress_amd import corpus
import cases
units = [cases.mixed_buffer()] + [corpus.by_name(x, 1500000).tobytes() for x in ("mozilla", "nci", "dickens", "x-ray", "osdb")] + [u for u in cases.edge_cases() if len(u) > 60000][:6]
out, st = m.compress_units(4, units)
assert all(s == 0 for s in st)
print("DIGEST", hashlib.sha256(b"".join(out)).hexdigest(), sum(len(o) for o in out))
""" % (ROOT, ROOT)


DEV_LIB = os.path.join(ROOT, "ms_compress_amd", "libmscomp_amd_dev.so")


def _run(env_extra, dev=True):
    env = dict(os.environ); env.update(env_extra)
    if dev:
        assert os.path.exists(DEV_LIB), "libmscomp_amd_dev.so is missing: __graft_entry__.build() makes it"
        env["MSCOMP_AMD_LIB"] = DEV_LIB
    else:
        env.pop("MSCOMP_AMD_LIB", None)
    r = subprocess.run([sys.executable, "-c", SNIPPET], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0]


def test_lazy_xpress_huff_finder_gives_the_same_bytes():
    assert _run({"MSCOMP_AMD_XH_LAZY": "1"}) == _run({"MSCOMP_AMD_XH_LAZY": "1"}, dev=False)     # (the product library ignores the switch: its one finder)


def test_sorted_span_xpress_huff_finder_gives_the_same_bytes():
    """csrc/xpress_sort.hip (round 5): the chunk's positions sorted by (hash, position), a position's <= 11 candidates read as two spans of
    that array instead of a chain walk -- a third way to the same bytes, also a measurement mode (MSCOMP_AMD_XH_SORT=1)."""
    assert _run({"MSCOMP_AMD_XH_SORT": "1"}) == _run({"MSCOMP_AMD_XH_SORT": "1"}, dev=False)

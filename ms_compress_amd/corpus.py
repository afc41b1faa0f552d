"""Deterministic Silesia-shaped synthetic corpus -- bench/test INPUT only.

The real Silesia files are not available offline; ``SILESIA_DIR`` is honoured when it holds the 12 files.
Otherwise 12 pseudo-files of exactly the Silesia sizes come from csrc/corpus.c (integer-only splitmix64).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["dickens", "mozilla", "mr", "nci", "ooffice", "osdb", "reymont", "samba", "sao", "webster", "xml", "x-ray"]
SIZES = [10192446, 51220480, 9970564, 33553445, 6152192, 10085684, 6627202, 21606400, 7251944, 41458703, 5345280, 8474240]
TOTAL = sum(SIZES)  # 211 938 580
_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(os.path.join(HERE, "libmscorpus.so"))
        lib.mscorpus_generate.argtypes = [C.c_int, C.c_void_p, C.c_uint64]
        lib.mscorpus_generate.restype = None
        _lib = lib
    return _lib


def source():
    d = os.environ.get("SILESIA_DIR")
    if d and all(os.path.isfile(os.path.join(d, n)) and os.path.getsize(os.path.join(d, n)) == s for n, s in zip(NAMES, SIZES)):
        return "silesia:" + d
    return "synthetic"


def file_bytes(index, n=None):
    """First n bytes (default: the whole file) of corpus member `index` as a numpy uint8 array."""
    size = SIZES[index] if n is None else int(n)
    src = source()
    if src.startswith("silesia:"):
        with open(os.path.join(src[8:], NAMES[index]), "rb") as f:
            return np.frombuffer(f.read(size), dtype=np.uint8).copy()
    out = np.empty(size, dtype=np.uint8)
    _load().mscorpus_generate(index, out.ctypes.data, size)
    return out


def by_name(name, n=None):
    return file_bytes(NAMES.index(name), n)

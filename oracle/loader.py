"""ctypes loaders for the checker libraries -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

* ``load_oracle()``  -> oracle/liboracle.so   (our plain-C restatement, oracle/mscomp_oracle.c)
* ``load_ref()``     -> oracle/_ref/libMSCompression.so (the real reference compiled by oracle/Makefile
                        from /root/reference/src; prebuilt file travels to the GPU box) or None.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
NONE, LZNT1, XPRESS, XPRESS_HUFF = 0, 2, 3, 4
FORMATS = {"lznt1": LZNT1, "xpress": XPRESS, "xpress_huff": XPRESS_HUFF}

_oracle = None
_ref = None


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present). Building the checker is not using it."""
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "mscomp_oracle.c")
    need = force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
    need_ref = os.path.isdir("/root/reference/src") and not (os.path.exists(os.path.join(HERE, "_ref", "libMSCompression.so")) and
                                                             os.path.exists(os.path.join(HERE, "_ref", "libMSCompression_sa.so")))
    if need or need_ref:
        subprocess.run(["make", "-C", HERE, "all"], check=True, stdout=subprocess.DEVNULL)


def load_oracle():
    global _oracle
    if _oracle is None:
        so = os.path.join(HERE, "liboracle.so")
        try:
            build()                                   # (mtime check inside: a stale prebuilt library lacks newer symbols -- ADVICE r05)
        except Exception:
            if not os.path.exists(so):
                raise
        lib = C.CDLL(so)
        lib.orc_compress.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
        lib.orc_compress.restype = C.c_int
        lib.orc_decompress.argtypes = lib.orc_compress.argtypes
        lib.orc_decompress.restype = C.c_int
        lib.orc_last_undefined.restype = C.c_int
        lib.orc_max_compressed_size.argtypes = [C.c_int, C.c_size_t]
        lib.orc_max_compressed_size.restype = C.c_size_t
        lib.orc_set_lznt1_sa_dict.argtypes = [C.c_int]
        lib.orc_set_lznt1_sa_dict.restype = None
        lib.orc_huff_lengths.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_huff_lengths_slow.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_lznt1_match_table.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
        lib.orc_xpress_match_table.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.orc_compress_units.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int]
        lib.orc_compress_units.restype = C.c_int
        lib.orc_time_units.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.orc_time_units.restype = C.c_double
        lib.orc_time_units_ex.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.orc_time_units_ex.restype = C.c_double
        lib.orc_time_units_ex2.argtypes = lib.orc_time_units_ex.argtypes + [C.POINTER(C.c_double)]
        lib.orc_time_units_ex2.restype = C.c_double
        _oracle = lib
    return _oracle


def load_ref():
    """The compiled reference, or None when oracle/_ref was not built (no /root/reference at build time)."""
    global _ref
    if _ref is None:
        so = os.path.join(HERE, "_ref", "libMSCompression.so")
        if not os.path.exists(so):
            return None
        lib = C.CDLL(so)
        for f in (lib.ms_compress, lib.ms_decompress):
            f.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
            f.restype = C.c_int
        lib.ms_max_compressed_size.argtypes = [C.c_int, C.c_size_t]
        lib.ms_max_compressed_size.restype = C.c_size_t
        _ref = lib
    return _ref


_ref_sa = None


def load_ref_sa():
    """The reference compiled with MSCOMP_WITH_LZNT1_SA_DICT (its suffix-array LZNT1 flavour), or None."""
    global _ref_sa
    if _ref_sa is None:
        so = os.path.join(HERE, "_ref", "libMSCompression_sa.so")
        if not os.path.exists(so):
            return None
        lib = C.CDLL(so)
        lib.ms_compress.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
        lib.ms_compress.restype = C.c_int
        lib.ms_max_compressed_size.argtypes = [C.c_int, C.c_size_t]
        lib.ms_max_compressed_size.restype = C.c_size_t
        _ref_sa = lib
    return _ref_sa


def oracle_compress_sa(data, cap=None):
    """LZNT1 with the suffix-array dictionary flavour (orc_set_lznt1_sa_dict around one orc_compress call)."""
    lib = load_oracle()
    lib.orc_set_lznt1_sa_dict(1)
    try:
        return oracle_compress(LZNT1, data, cap)
    finally:
        lib.orc_set_lznt1_sa_dict(0)


def ref_compress_sa(data, cap=None):
    lib = load_ref_sa()
    if cap is None:
        cap = lib.ms_max_compressed_size(LZNT1, len(data))
    return _one_shot(lib.ms_compress, LZNT1, data, cap)


def _one_shot(fn, fmt, data, cap):
    data = bytes(data)
    out = C.create_string_buffer(cap + 64)
    n = C.c_size_t(cap)
    st = fn(fmt, data, len(data), out, C.byref(n))
    return st, (out.raw[: n.value] if st == 0 else b"")


def oracle_compress(fmt, data, cap=None):
    lib = load_oracle()
    if cap is None:
        cap = lib.orc_max_compressed_size(fmt, len(data))
    return _one_shot(lib.orc_compress, fmt, data, cap)


def oracle_decompress(fmt, data, out_len):
    return _one_shot(load_oracle().orc_decompress, fmt, data, out_len)


def oracle_decompress_ex(fmt, data, cap):
    """(status, bytes, undefined): ``undefined`` is True when the reference's behaviour on this stream is undefined
    (the restatement then says DATA_ERROR and the compiled reference must not be asked)."""
    lib = load_oracle()
    st, out = _one_shot(lib.orc_decompress, fmt, data, cap)
    return st, out, bool(lib.orc_last_undefined())


def time_units(fn, fmt, units, caps, threads, passes):
    """Seconds for `passes` passes of fn (a ctypes function with the one-shot signature, e.g. load_ref().ms_decompress; None = the
    oracle's compressor) over independent units, on `threads` C threads (no interpreter between the calls). Returns (seconds, statuses, lengths)."""
    import numpy as np
    lib = load_oracle()
    in_off = np.zeros(len(units) + 1, dtype=np.uint64); in_off[1:] = np.cumsum([len(u) for u in units])
    out_off = np.zeros(len(units) + 1, dtype=np.uint64); out_off[1:] = np.cumsum([int(c) for c in caps])
    blob = np.frombuffer(b"".join(bytes(u) for u in units) or b"\0", dtype=np.uint8)
    out = np.zeros(int(out_off[-1]) + 64, dtype=np.uint8)
    out_len = np.zeros(len(units), dtype=np.uint64); status = np.zeros(len(units), dtype=np.int32)
    fp = C.cast(fn, C.c_void_p) if fn is not None else None
    dt = lib.orc_time_units(fp, fmt, blob.ctypes.data, in_off.ctypes.data, len(units), out.ctypes.data, out_off.ctypes.data,
                            out_len.ctypes.data, status.ctypes.data, threads, passes)
    return dt, status, out_len


def time_units_ex(fn, fmt, blob, in_off, in_len, caps, threads, passes, keep_output=False, busy=None):
    """time_units for units given as (offset, length) into ONE numpy uint8 array (units may alias: replicas of the same files cost no
    memory on the input side), handed to the threads one at a time in the order given. Returns (seconds, statuses, lengths[, out, out_off]).
    busy: a list that receives the thread-seconds spent inside fn during this call (its own figure: concurrent calls do not mix)."""
    import numpy as np
    lib = load_oracle()
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64); in_len = np.ascontiguousarray(in_len, dtype=np.uint64)
    caps = np.ascontiguousarray(caps, dtype=np.uint64)
    out_off = np.zeros(len(caps), dtype=np.uint64); out_off[1:] = np.cumsum(caps[:-1] + np.uint64(64))
    out = np.empty(int(out_off[-1] + caps[-1]) + 64 if len(caps) else 64, dtype=np.uint8)
    out_len = np.zeros(len(caps), dtype=np.uint64); status = np.zeros(len(caps), dtype=np.int32)
    fp = C.cast(fn, C.c_void_p) if fn is not None else None
    b = C.c_double(0.0)
    dt = lib.orc_time_units_ex2(fp, fmt, blob.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, len(caps), out.ctypes.data, out_off.ctypes.data,
                                caps.ctypes.data, out_len.ctypes.data, status.ctypes.data, threads, passes, C.byref(b))
    if busy is not None:
        busy.append(b.value)
    return (dt, status, out_len, out, out_off) if keep_output else (dt, status, out_len)


def ref_compress(fmt, data, cap=None):
    lib = load_ref()
    if cap is None:
        cap = lib.ms_max_compressed_size(fmt, len(data))
    return _one_shot(lib.ms_compress, fmt, data, cap)


def ref_decompress(fmt, data, out_len):
    return _one_shot(load_ref().ms_decompress, fmt, data, out_len)

"""Host-side binding of libmscomp_amd.so (the C-ABI declared in include/mscomp_amd.h).

Mirrors the reference's own ctypes fixture (/root/reference/test/compressors.py:187-251: ``OpenSrc.Compress`` ->
``ms_compress``; default output buffer, status codes) so that the parity tests read like the reference's tests,
and adds the batch interface (device-resident units) the GPU path needs.

PyTorch is plumbing only: device memory (``torch.empty(..., device="cuda")``), the current HIP stream and
``torch.distributed``. There is NO CPU encoder behind these functions: if the HIP extension is missing or no
GPU is visible they raise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSCOMP_AMD_LIB") or os.path.join(HERE, "libmscomp_amd.so")   # (MSCOMP_AMD_LIB: a development build of the same library, e.g. with a *_PROFILE macro)

# enum values identical to /root/reference/include/mscomp/general.h:65-85
MSCOMP_NONE, MSCOMP_RESERVED, MSCOMP_LZNT1, MSCOMP_XPRESS, MSCOMP_XPRESS_HUFF = 0, 1, 2, 3, 4
MSCOMP_OK, MSCOMP_ERRNO, MSCOMP_ARG_ERROR, MSCOMP_DATA_ERROR, MSCOMP_MEM_ERROR, MSCOMP_BUF_ERROR = 0, -1, -2, -3, -4, -5
FORMATS = {"lznt1": MSCOMP_LZNT1, "xpress": MSCOMP_XPRESS, "xpress_huff": MSCOMP_XPRESS_HUFF}
CHUNK = {MSCOMP_LZNT1: 4096, MSCOMP_XPRESS: 65536, MSCOMP_XPRESS_HUFF: 65536}

EXPORTS = [  # every symbol include/mscomp_amd.h declares (tests check the library exports all of them)
    "ms_compress", "ms_max_compressed_size",
    "lznt1_compress", "lznt1_max_compressed_size", "xpress_compress", "xpress_max_compressed_size",
    "xpress_huff_compress", "xpress_huff_max_compressed_size",
    "mscomp_amd_ctx_create", "mscomp_amd_ctx_destroy", "mscomp_amd_plan_create", "mscomp_amd_plan_destroy",
    "mscomp_amd_plan_execute", "mscomp_amd_compress_batch", "mscomp_amd_profile_enable", "mscomp_amd_profile_read",
    "mscomp_amd_plan_layout", "mscomp_amd_compact_batch",
    "ms_inflate_init", "ms_inflate", "ms_inflate_end", "lznt1_inflate_init", "lznt1_inflate", "lznt1_inflate_end",
    "ms_deflate_init", "ms_deflate", "ms_deflate_end", "lznt1_deflate_init", "lznt1_deflate", "lznt1_deflate_end",
    "xpress_deflate_init", "xpress_deflate", "xpress_deflate_end", "xpress_inflate_init", "xpress_inflate", "xpress_inflate_end",
    "ms_decompress", "lznt1_decompress", "xpress_decompress", "xpress_huff_decompress", "mscomp_amd_plan_create_decompress", "mscomp_amd_decompress_batch",
    "mscomp_amd_version", "mscomp_amd_debug_xpress_matches", "mscomp_amd_debug_huff_lengths", "mscomp_amd_debug_lds_lane_order", "mscomp_amd_debug_set_xpress_emit", "mscomp_amd_debug_set_lznt1", "mscomp_amd_debug_set_serial_atomics", "mscomp_amd_compress_units_host", "mscomp_amd_decompress_units_host", "mscomp_amd_host_pool_release", "mscomp_amd_debug_set_finder", "mscomp_amd_debug_set_one_shot", "mscomp_amd_debug_set_xpress_decoder", "mscomp_amd_debug_lzg_open", "mscomp_amd_set_lznt1_sa_dict", "mscomp_amd_get_lznt1_sa_dict", "mscomp_amd_ctx_set_lznt1_sa_dict", "mscomp_amd_debug_hooks_enabled", "mscomp_amd_debug_lzd_walked",
]


class MSCompError(RuntimeError):
    def __init__(self, status, what=""):
        super().__init__("mscomp status %d %s" % (status, what))
        self.status = status


_lib = None


def load_library():
    """Load libmscomp_amd.so. Fails loudly when the HIP extension has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    try:                      # PyTorch bundles its own libamdhip64; this library links /opt/rocm's. Both live in one process, and the
        import torch  # noqa: F401   device is only visible to both when torch's runtime was loaded first (seen on the MI355X box).
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    one_shot = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    lib.ms_compress.argtypes = [C.c_int] + one_shot
    lib.ms_compress.restype = C.c_int
    lib.ms_max_compressed_size.argtypes = [C.c_int, C.c_size_t]
    lib.ms_max_compressed_size.restype = C.c_size_t
    for name in ("lznt1", "xpress", "xpress_huff"):
        f = getattr(lib, name + "_compress")
        f.argtypes = one_shot
        f.restype = C.c_int
        g = getattr(lib, name + "_max_compressed_size")
        g.argtypes = [C.c_size_t]
        g.restype = C.c_size_t
    lib.mscomp_amd_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.mscomp_amd_ctx_create.restype = C.c_int
    lib.mscomp_amd_ctx_destroy.argtypes = [C.c_void_p]
    lib.mscomp_amd_ctx_destroy.restype = None
    lib.mscomp_amd_plan_create.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_void_p)]
    lib.mscomp_amd_plan_create.restype = C.c_int
    lib.mscomp_amd_plan_destroy.argtypes = [C.c_void_p]
    lib.mscomp_amd_plan_destroy.restype = None
    lib.mscomp_amd_plan_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mscomp_amd_plan_execute.restype = C.c_int
    lib.mscomp_amd_compress_batch.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mscomp_amd_compress_batch.restype = C.c_int
    lib.ms_decompress.argtypes = [C.c_int] + one_shot
    lib.ms_decompress.restype = C.c_int
    for name in ("lznt1", "xpress", "xpress_huff"):
        f = getattr(lib, name + "_decompress")
        f.argtypes = one_shot
        f.restype = C.c_int
    lib.mscomp_amd_plan_create_decompress.argtypes = lib.mscomp_amd_plan_create.argtypes
    lib.mscomp_amd_plan_create_decompress.restype = C.c_int
    lib.mscomp_amd_decompress_batch.argtypes = lib.mscomp_amd_compress_batch.argtypes
    lib.mscomp_amd_decompress_batch.restype = C.c_int
    lib.mscomp_amd_plan_layout.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.mscomp_amd_plan_layout.restype = C.c_uint64
    lib.mscomp_amd_compact_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mscomp_amd_compact_batch.restype = C.c_int
    lib.mscomp_amd_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.mscomp_amd_profile_enable.restype = None
    lib.mscomp_amd_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.mscomp_amd_profile_read.restype = C.c_int
    lib.mscomp_amd_version.restype = C.c_char_p
    lib.mscomp_amd_debug_xpress_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    lib.mscomp_amd_debug_xpress_matches.restype = C.c_int
    lib.mscomp_amd_debug_huff_lengths.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.mscomp_amd_debug_huff_lengths.restype = C.c_int
    lib.mscomp_amd_debug_lds_lane_order.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.mscomp_amd_debug_lds_lane_order.restype = C.c_uint32
    lib.mscomp_amd_debug_set_xpress_emit.argtypes = [C.c_int]
    lib.mscomp_amd_debug_set_xpress_emit.restype = None
    lib.mscomp_amd_set_lznt1_sa_dict.argtypes = [C.c_int]
    lib.mscomp_amd_set_lznt1_sa_dict.restype = None
    lib.mscomp_amd_get_lznt1_sa_dict.argtypes = []
    lib.mscomp_amd_get_lznt1_sa_dict.restype = C.c_int
    lib.mscomp_amd_ctx_set_lznt1_sa_dict.argtypes = [C.c_void_p, C.c_int]
    lib.mscomp_amd_ctx_set_lznt1_sa_dict.restype = C.c_int
    lib.mscomp_amd_debug_hooks_enabled.argtypes = []
    lib.mscomp_amd_debug_hooks_enabled.restype = C.c_int
    lib.mscomp_amd_debug_set_xpress_decoder.argtypes = [C.c_int]
    lib.mscomp_amd_debug_set_xpress_decoder.restype = None
    lib.mscomp_amd_debug_set_finder.argtypes = [C.c_int]
    lib.mscomp_amd_debug_set_finder.restype = None
    lib.mscomp_amd_compress_units_host.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mscomp_amd_compress_units_host.restype = C.c_int
    lib.mscomp_amd_decompress_units_host.argtypes = lib.mscomp_amd_compress_units_host.argtypes
    lib.mscomp_amd_decompress_units_host.restype = C.c_int
    lib.mscomp_amd_host_pool_release.argtypes = []
    lib.mscomp_amd_host_pool_release.restype = None
    lib.mscomp_amd_debug_set_one_shot.argtypes = [C.c_int]
    lib.mscomp_amd_debug_set_one_shot.restype = None
    lib.mscomp_amd_debug_set_lznt1.argtypes = [C.c_int]
    lib.mscomp_amd_debug_set_lznt1.restype = None
    lib.mscomp_amd_debug_set_serial_atomics.argtypes = [C.c_int]
    lib.mscomp_amd_debug_set_serial_atomics.restype = None
    lib.mscomp_amd_debug_lzd_walked.argtypes = [C.c_void_p]
    lib.mscomp_amd_debug_lzd_walked.restype = C.c_uint32
    _lib = lib
    return lib


def max_compressed_size(fmt, n):
    """ms_max_compressed_size (/root/reference/src/mscomp.cpp:96-100)."""
    return load_library().ms_max_compressed_size(int(fmt), int(n))


def compress(fmt, data, out_capacity=None):
    """One-shot ``ms_compress`` with HOST buffers (the reference contract, mscomp.h:59): returns the compressed
    bytes or raises MSCompError(status). Default capacity follows compressors.py:248 / ms_max_compressed_size."""
    lib = load_library()
    data = bytes(data)
    cap = max_compressed_size(fmt, len(data)) if out_capacity is None else int(out_capacity)
    out = C.create_string_buffer(cap + 2)
    n = C.c_size_t(cap)
    st = lib.ms_compress(int(fmt), data, len(data), out, C.byref(n))
    if st != MSCOMP_OK:
        raise MSCompError(st, "ms_compress(format=%d, in_len=%d, capacity=%d)" % (fmt, len(data), cap))
    return out.raw[: n.value]


def decompress(fmt, data, out_capacity):
    """One-shot ``ms_decompress`` with HOST buffers (mscomp.h:79): ``out_capacity`` is what the caller passes in *out_len.
    Returns the decompressed bytes or raises MSCompError(status)."""
    lib = load_library()
    data = bytes(data)
    cap = int(out_capacity)
    out = C.create_string_buffer(cap + 1)
    n = C.c_size_t(cap)
    st = lib.ms_decompress(int(fmt), data, len(data), out, C.byref(n))
    if st != MSCOMP_OK:
        raise MSCompError(st, "ms_decompress(format=%d, in_len=%d, capacity=%d)" % (fmt, len(data), cap))
    return out.raw[: n.value]


def pack_offsets(lengths, align=16):
    """Start offsets (n entries, uint64) of units of the given byte lengths packed back to back with every start
    aligned, and the total size. """
    off = np.zeros(len(lengths), dtype=np.uint64)
    pos = 0
    for i, n in enumerate(lengths):
        off[i] = pos
        pos += (int(n) + align - 1) // align * align
    return off, pos


class Context:
    """mscomp_amd_ctx: one per (device, stream). Uses torch's current device/stream by default."""

    def __init__(self, device=None, stream=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: ms_compress_amd has no CPU fallback")
        self.lib = load_library()
        self.device = torch.cuda.current_device() if device is None else int(device)
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream() if stream is None else stream
        self.stream = s
        self._h = C.c_void_p()
        st = self.lib.mscomp_amd_ctx_create(self.device, C.c_void_p(s.cuda_stream), C.byref(self._h))
        if st != MSCOMP_OK:
            raise MSCompError(st, "mscomp_amd_ctx_create")

    def close(self):
        if self._h:
            self.lib.mscomp_amd_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_lznt1_sa_dict(self, on):
        """LZNT1 dictionary flavour of the plans this context creates from now on: True / False, None = the process default."""
        st = self.lib.mscomp_amd_ctx_set_lznt1_sa_dict(self._h, -1 if on is None else int(bool(on)))
        if st != MSCOMP_OK:
            raise MSCompError(st, "mscomp_amd_ctx_set_lznt1_sa_dict")

    def profile_enable(self, on=True):
        self.lib.mscomp_amd_profile_enable(self._h, 1 if on else 0)

    def profile_read(self):
        """{kernel name: (total ms, launches)} since the last read (synchronizes the stream)."""
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        cnt = (C.c_uint64 * cap)()
        n = self.lib.mscomp_amd_profile_read(self._h, names, ms, cnt, cap)
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n)}


class Plan:
    """mscomp_amd_plan: the unit layout of one batch (offset tables uploaded once, scratch sized once)."""

    def __init__(self, ctx, fmt, in_off, in_len, out_off, out_cap, decompress=False):
        self.ctx, self.fmt = ctx, int(fmt)
        arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (in_off, in_len, out_off, out_cap)]
        assert all(a.ndim == 1 and a.shape == arrs[0].shape for a in arrs)
        self.in_off, self.in_len, self.out_off, self.out_cap = arrs
        self.n_units = len(self.in_off)
        self._h = C.c_void_p()
        create = ctx.lib.mscomp_amd_plan_create_decompress if decompress else ctx.lib.mscomp_amd_plan_create
        st = create(ctx._h, self.fmt, self.n_units, *[a.ctypes.data for a in arrs], C.byref(self._h))
        if st != MSCOMP_OK:
            raise MSCompError(st, "mscomp_amd_plan_create")

    def execute(self, d_in, d_out, d_out_len, d_status):
        """Enqueue on the ctx stream. Arguments are torch CUDA tensors (uint8, uint8, int64/uint64[n], int32[n])."""
        st = self.ctx.lib.mscomp_amd_plan_execute(self._h, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                                  C.c_void_p(d_out_len.data_ptr()), C.c_void_p(d_status.data_ptr()))
        if st != MSCOMP_OK:
            raise MSCompError(st, "mscomp_amd_plan_execute")

    def close(self):
        if self._h:
            self.ctx.lib.mscomp_amd_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def compact_batch(ctx, out_off, out_cap, d_out, d_out_len):
    """mscomp_amd_compact_batch: the outputs of an executed batch packed back to back, in unit order, on the
    device. Returns (d_packed uint8 tensor, d_packed_off int64 tensor of n + 1 offsets); enqueued on the ctx stream."""
    import torch
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint64)
    n = len(out_off)
    dev = torch.device("cuda", ctx.device)
    d_packed = torch.empty(int(out_cap.sum()) + 16, dtype=torch.uint8, device=dev)
    d_poff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    st = ctx.lib.mscomp_amd_compact_batch(ctx._h, n, C.c_void_p(d_out.data_ptr()), out_off.ctypes.data, out_cap.ctypes.data,
                                          C.c_void_p(d_out_len.data_ptr()), C.c_void_p(d_packed.data_ptr()), C.c_void_p(d_poff.data_ptr()))
    if st != MSCOMP_OK:
        raise MSCompError(st, "mscomp_amd_compact_batch")
    return d_packed, d_poff


def decompress_units(fmt, units, capacities, ctx=None):
    """Decompress a list of independent compressed buffers on the GPU (each exactly as one ms_decompress call would, with
    *out_len = capacities[i] on entry). Returns (list of bytes, or None where the status is not MSCOMP_OK; list of status)."""
    return compress_units(fmt, units, ctx=ctx, capacities=capacities, decompress=True)


def compress_units(fmt, units, ctx=None, capacities=None, decompress=False):
    """Compress a list of independent byte strings on the GPU (each exactly as one ms_compress call would).
    ``capacities`` (optional) gives the exact output capacity of every unit (default: ms_max_compressed_size+2).
    Returns (list of compressed bytes, or None where the unit got MSCOMP_BUF_ERROR; list of status)."""
    import torch
    own = ctx is None
    ctx = ctx or Context()
    lens = [len(u) for u in units]
    in_off, in_total = pack_offsets(lens)
    if capacities is None:
        caps = [max_compressed_size(fmt, n) + 2 for n in lens]
        out_off, out_total = pack_offsets(caps)
    else:
        caps = [int(x) for x in capacities]
        out_off, out_total = pack_offsets(caps, align=1)
    blob = np.zeros(in_total + 16, dtype=np.uint8)
    for u, o in zip(units, in_off):
        if len(u):
            blob[int(o): int(o) + len(u)] = np.frombuffer(bytes(u), dtype=np.uint8)
    dev = torch.device("cuda", ctx.device)
    with torch.cuda.device(ctx.device), torch.cuda.stream(ctx.stream):
        d_in = torch.from_numpy(blob).to(dev)
        d_out = torch.zeros(out_total + 16, dtype=torch.uint8, device=dev)
        d_len = torch.zeros(max(1, len(units)), dtype=torch.int64, device=dev)
        d_st = torch.zeros(max(1, len(units)), dtype=torch.int32, device=dev)
        plan = Plan(ctx, fmt, in_off, lens, out_off, caps, decompress=decompress)
        plan.execute(d_in, d_out, d_len, d_st)
        ctx.stream.synchronize()
        h_out, h_len, h_st = d_out.cpu().numpy(), d_len.cpu().numpy(), d_st.cpu().numpy()
        plan.close()
    res = []
    for i in range(len(units)):
        o = int(out_off[i])
        res.append(bytes(h_out[o: o + int(h_len[i])]) if h_st[i] == MSCOMP_OK else None)
    if own:
        ctx.close()
    return res, [int(x) for x in h_st[: len(units)]]




def decompress_units_host(fmt, in_arrays, out_arrays, devices=(0,)):
    """mscomp_amd_decompress_units_host: like compress_units_host for the decoders (capacity of unit i = len(out_arrays[i]))."""
    return compress_units_host(fmt, in_arrays, out_arrays, devices=devices, decompress=True)


class HostViews:
    """n units as views of ONE numpy uint8 array: unit i = base[off[i] : off[i] + length[i]]. The pointer table of a call is then built
    by numpy (base address + offsets) instead of one Python attribute access per unit (milliseconds for thousands of units)."""

    def __init__(self, base, off, length):
        self.base = base
        self.off = np.ascontiguousarray(off, dtype=np.uint64)
        self.length = np.ascontiguousarray(length, dtype=np.uint64)
        assert len(self.off) == len(self.length) and (len(self.off) == 0 or int((self.off + self.length).max()) <= base.size)

    def __len__(self):
        return len(self.off)

    def tables(self):
        return (self.off + np.uint64(self.base.ctypes.data)), self.length


def _host_tables(arrays):
    if isinstance(arrays, HostViews):
        return arrays.tables()
    n = len(arrays)
    return (np.array([a.ctypes.data for a in arrays], dtype=np.uint64) if n else np.zeros(0, np.uint64),
            np.array([a.size for a in arrays], dtype=np.uint64) if n else np.zeros(0, np.uint64))


def compress_units_host(fmt, in_arrays, out_arrays, devices=(0,), decompress=False):
    """mscomp_amd_compress_units_host: units given as numpy uint8 arrays (host memory, any layout) or as HostViews of one array, outputs
    written into the numpy uint8 arrays / HostViews of out_arrays (capacity = their length), on the GPUs `devices` (a range per entry; an
    ordinal may repeat). Returns (status of the call, out_lens uint64 array, statuses int32 array). Nothing is copied on the Python side."""
    lib = load_library()
    n = len(in_arrays)
    assert len(out_arrays) == n
    ip, il = _host_tables(in_arrays)
    op, oc = _host_tables(out_arrays)
    pad = np.zeros(1, np.uint64)
    ip, il, op, oc = [(x if n else pad) for x in (ip, il, op, oc)]
    ol = np.zeros(max(1, n), dtype=np.uint64)
    st = np.full(max(1, n), -9, dtype=np.int32)
    dv = (C.c_int * len(devices))(*[int(d) for d in devices])
    fn = lib.mscomp_amd_decompress_units_host if decompress else lib.mscomp_amd_compress_units_host
    rc = fn(int(fmt), len(devices), dv, n, ip.ctypes.data, il.ctypes.data, op.ctypes.data, oc.ctypes.data, ol.ctypes.data, st.ctypes.data)
    return rc, ol[:n], st[:n]

"""CPU: the C-ABI library loads and exports every symbol include/mscomp_amd.h declares; host-only behaviour
(sizes, argument errors, the MSCOMP_NONE copy codec) matches the reference facade. No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mscomp_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:ms|lznt1|xpress|xpress_huff|mscomp_amd)_[a-z0-9_]+)\s*\(", txt)))


def test_exports_every_declared_symbol():
    import ms_compress_amd as m
    lib = m.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(lib, s), "libmscomp_amd.so does not export " + s
    assert sorted(m.api.EXPORTS) == syms, "api.EXPORTS and include/mscomp_amd.h disagree"


def test_max_compressed_size_matches_reference_table():
    import ms_compress_amd as m
    n = [0, 1, 4096, 65536, 1 << 20]
    assert [m.max_compressed_size(2, x) for x in n] == [3, 6, 4101, 65571, 1049091]
    assert [m.max_compressed_size(3, x) for x in n] == [4, 5, 4612, 73732, 1179652]
    assert [m.max_compressed_size(4, x) for x in n] == [292, 293, 4388, 66086, 1052996]
    assert [m.max_compressed_size(0, x) for x in n] == n
    assert m.max_compressed_size(1, 10) == 2 ** 64 - 1 and m.max_compressed_size(5, 10) == 2 ** 64 - 1   # mscomp.cpp:98
    lib = m.load_library()
    assert lib.lznt1_max_compressed_size(4096) == 4101 and lib.xpress_max_compressed_size(65536) == 73732
    assert lib.xpress_huff_max_compressed_size(65536) == 66086


def test_argument_errors_without_gpu():
    import ms_compress_amd as m
    lib = m.load_library()
    out = C.create_string_buffer(64)
    n = C.c_size_t(64)
    assert lib.ms_compress(1, b"abc", 3, out, C.byref(n)) == m.MSCOMP_ARG_ERROR      # MSCOMP_RESERVED (mscomp.cpp:115)
    assert lib.ms_compress(7, b"abc", 3, out, C.byref(n)) == m.MSCOMP_ARG_ERROR
    n = C.c_size_t(64)
    assert lib.ms_compress(0, b"abc", 3, out, C.byref(n)) == 0 and n.value == 3 and out.raw[:3] == b"abc"   # copy codec
    n = C.c_size_t(2)
    assert lib.ms_compress(0, b"abc", 3, out, C.byref(n)) == m.MSCOMP_BUF_ERROR
    ctx = C.c_void_p()
    assert lib.mscomp_amd_ctx_create(9999, None, C.byref(ctx)) == m.MSCOMP_ERRNO     # no such device: fails loudly


def test_no_cpu_fallback_without_gpu():
    """On a machine without a GPU the compress path must FAIL (never silently run on the CPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ms_compress_amd as m
    with pytest.raises(m.MSCompError) as e:
        m.compress(2, b"hello hello hello hello")
    assert e.value.status == m.MSCOMP_ERRNO
    with pytest.raises(RuntimeError):
        m.Context()


def test_product_does_not_touch_oracle():
    """Nothing under ms_compress_amd/ may import, link or mention the checker (oracle/)."""
    pkg = os.path.join(ROOT, "ms_compress_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "orc_compress" not in txt and "from oracle" not in txt, os.path.join(dirpath, f)


def test_pack_offsets():
    import ms_compress_amd as m
    off, total = m.pack_offsets([5, 0, 17, 16])
    assert off.tolist() == [0, 16, 16, 48] and total == 64
    off, total = m.pack_offsets([5, 0, 17], align=1)
    assert off.tolist() == [0, 5, 5] and total == 22


def test_plan_layout_is_the_max_compressed_size_layout():
    """mscomp_amd_plan_layout (host arithmetic, no GPU): capacities from ms_max_compressed_size, aligned running offsets"""
    import ctypes as C
    import numpy as np
    import ms_compress_amd as m
    lib = m.load_library()
    lens = np.array([0, 1, 4096, 65536, 70001, 1 << 20], dtype=np.uint64)
    for fmt in (0, 2, 3, 4):
        off = np.zeros(len(lens), dtype=np.uint64); cap = np.zeros(len(lens), dtype=np.uint64)
        total = lib.mscomp_amd_plan_layout(fmt, len(lens), lens.ctypes.data, 16, off.ctypes.data, cap.ctypes.data)
        pos = 0
        for i, n in enumerate(lens):
            want = m.max_compressed_size(fmt, int(n)) + (2 if fmt == 2 else 0)
            assert int(cap[i]) == want and int(off[i]) == pos and pos % 16 == 0
            pos += (want + 15) // 16 * 16
        assert total == pos
    assert lib.mscomp_amd_plan_layout(7, 1, lens.ctypes.data, 16, None, None) == 2 ** 64 - 1


def test_stream_object_has_the_reference_layout():
    """mscomp_stream as the streaming tests bind it (include/mscomp/general.h:95-121, default build with error + warning texts):
    576 bytes, the state pointer last"""
    import ctypes as C

    class Stream(C.Structure):
        _fields_ = [("format", C.c_int), ("compressing", C.c_bool), ("in_", C.c_void_p), ("in_avail", C.c_size_t), ("in_total", C.c_size_t),
                    ("out", C.c_void_p), ("out_avail", C.c_size_t), ("out_total", C.c_size_t), ("error", C.c_char * 256),
                    ("warning", C.c_char * 256), ("state", C.c_void_p)]
    assert C.sizeof(Stream) == 576 and Stream.state.offset == 568 and Stream.in_.offset == 8 and Stream.error.offset == 56
    import ms_compress_amd as m
    lib = m.load_library()
    s = Stream()
    lib.ms_deflate_init.argtypes = [C.c_int, C.POINTER(Stream)]
    lib.ms_deflate_end.argtypes = [C.POINTER(Stream)]
    assert lib.ms_deflate_init(0, C.byref(s)) == 0 and s.format == 0 and s.compressing and not s.state     # the copy codec needs no GPU
    assert lib.ms_deflate_end(C.byref(s)) == 0


def test_xpress_deflate_entry_points_answer_like_the_reference():
    """include/xpress.h:52-54: the unfinished Xpress streaming compressor. Same statuses as the compiled reference for a fresh stream,
    a stream with the Xpress format and no state, and one with a (fake) state -- directly and through the ms_deflate* facade. No GPU needed."""
    import ctypes as C
    from oracle import loader
    import ms_compress_amd as m

    class Stream(C.Structure):
        _fields_ = [("format", C.c_int), ("compressing", C.c_bool), ("in_", C.c_void_p), ("in_avail", C.c_size_t), ("in_total", C.c_size_t),
                    ("out", C.c_void_p), ("out_avail", C.c_size_t), ("out_total", C.c_size_t), ("error", C.c_char * 256),
                    ("warning", C.c_char * 256), ("state", C.c_void_p)]

    def answers(lib):
        res = []
        for f in (lib.xpress_deflate_init, lib.xpress_deflate_end, lib.ms_deflate_end):
            f.argtypes = [C.c_void_p]
        lib.xpress_deflate.argtypes = lib.ms_deflate.argtypes = [C.c_void_p, C.c_int]
        lib.ms_deflate_init.argtypes = [C.c_int, C.c_void_p]
        for setup in range(3):
            s = Stream()
            if setup >= 1:
                s.format, s.compressing = 3, True
            if setup == 2:
                s.state = 0x1000
            res.append(lib.xpress_deflate_init(C.byref(s)))
            res.append((s.format, bool(s.compressing), s.state))          # init must not touch the stream
            res.append(lib.xpress_deflate(C.byref(s), 0))
            res.append(lib.xpress_deflate_end(C.byref(s)))
            res.append(lib.ms_deflate_init(3, C.byref(s)))
            if setup >= 1:
                res.append(lib.ms_deflate(C.byref(s), 4))
                res.append(lib.ms_deflate_end(C.byref(s)))
        return res
    ours = answers(m.load_library())
    assert ours[:5] == [m.MSCOMP_MEM_ERROR, (0, False, None), m.MSCOMP_ARG_ERROR, m.MSCOMP_ARG_ERROR, m.MSCOMP_MEM_ERROR]
    ref = loader.load_ref()
    if ref is not None:
        assert ours == answers(ref)


def test_xpress_inflate_stub_says_why():
    """The streaming Xpress DEcompressor is not offloaded: the init call answers MSCOMP_MEM_ERROR, names the reason in
    stream->error and leaves the rest of the stream untouched -- directly and through ms_inflate_init. No GPU needed."""
    import ctypes as C
    import ms_compress_amd as m

    class Stream(C.Structure):
        _fields_ = [("format", C.c_int), ("compressing", C.c_bool), ("in_", C.c_void_p), ("in_avail", C.c_size_t), ("in_total", C.c_size_t),
                    ("out", C.c_void_p), ("out_avail", C.c_size_t), ("out_total", C.c_size_t), ("error", C.c_char * 256),
                    ("warning", C.c_char * 256), ("state", C.c_void_p)]
    lib = m.load_library()
    if not hasattr(lib, "xpress_inflate_init"):
        pytest.skip("built with -DMSCOMP_AMD_NO_XPRESS_INFLATE")
    lib.xpress_inflate_init.argtypes = [C.c_void_p]
    lib.ms_inflate_init.argtypes = [C.c_int, C.c_void_p]
    for call in (lambda s: lib.xpress_inflate_init(C.byref(s)), lambda s: lib.ms_inflate_init(3, C.byref(s))):
        s = Stream()
        s.in_avail, s.state = 77, 0x1000
        assert call(s) == m.MSCOMP_MEM_ERROR
        assert b"not offloaded" in s.error and s.in_avail == 77 and s.state == 0x1000 and s.format == 0
    assert lib.xpress_inflate_init(None) == m.MSCOMP_MEM_ERROR


def test_every_c_symbol_of_the_reference_is_exported():
    """A program linked against libMSCompression must link against the drop-in: every C-linkage function the compiled reference exports
    (include/mscomp.h, lznt1.h, xpress.h, xpress_huff.h) is exported by libmscomp_amd.so as well (C++-mangled internals aside)."""
    import subprocess
    ref = os.path.join(ROOT, "oracle", "_ref", "libMSCompression.so")
    if not os.path.exists(ref):
        pytest.skip("compiled reference not available")

    def syms(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T"}
    want = {s for s in syms(ref) if not s.startswith("_Z") and not s.startswith("_")}
    ours = syms(os.path.join(ROOT, "ms_compress_amd", "libmscomp_amd.so"))
    assert len(want) >= 25 and not (want - ours), sorted(want - ours)


def test_kernel_switches_are_inert_unless_the_process_asked_for_them():
    """mscomp_amd_debug_set_*: process-wide switches between bit-identical kernels, for tests. The library honours them only when
    MSCOMP_AMD_TEST_HOOKS=1 was in the environment when it loaded (tests/conftest.py sets it); any other process gets no-ops."""
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r); import ms_compress_amd as m; l = m.load_library(); l.mscomp_amd_debug_set_finder(2); print(l.mscomp_amd_debug_hooks_enabled())" % ROOT
    for val, want in ((None, "0"), ("1", "1"), ("0", "0")):
        env = {k: v for k, v in os.environ.items() if k != "MSCOMP_AMD_TEST_HOOKS"}
        if val is not None:
            env["MSCOMP_AMD_TEST_HOOKS"] = val
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == want, (val, r.stdout, r.stderr[-500:])

"""ms_compress_amd -- MI355X-native (gfx950, HIP) one-shot compressors for the MS-XCA codecs LZNT1, Xpress and
Xpress+Huffman, bit-exact with coderforlife/ms-compress, behind that library's ``mscomp.h`` C-ABI.

Only the compress hot path is here: ``csrc/`` holds the HIP kernels and the C-ABI
(``libmscomp_amd.so``, declared in ``include/mscomp_amd.h``); ``api.py`` is the host-side mirror of the
reference interface; ``corpus.py`` the synthetic Silesia-shaped bench input; ``sharding.py`` the per-GPU split.
"""
from .api import (MSCOMP_NONE, MSCOMP_LZNT1, MSCOMP_XPRESS, MSCOMP_XPRESS_HUFF, MSCOMP_OK, MSCOMP_ERRNO,  # noqa: F401
                  MSCOMP_ARG_ERROR, MSCOMP_DATA_ERROR, MSCOMP_MEM_ERROR, MSCOMP_BUF_ERROR, FORMATS, CHUNK,
                  MSCompError, load_library, max_compressed_size, compress, compress_units, compress_units_host, decompress_units_host, HostViews, pack_offsets, decompress, decompress_units, compact_batch,
                  Context, Plan)

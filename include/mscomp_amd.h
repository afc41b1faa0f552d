/* include/mscomp_amd.h -- C-ABI of libmscomp_amd.so, the MI355X-native drop-in for the one-shot
 * compressors of coderforlife/ms-compress.
 *
 * Part 1 re-exports, with identical names / argument meaning / error behaviour, exactly the symbols that
 * the reference's three compressor translation units and its facade export for this path:
 *
 *   reference symbol (file:line)                                              -> replaced by
 *   ms_compress               include/mscomp.h:59,  src/mscomp.cpp:113-117     -> ms_compress
 *   ms_max_compressed_size    include/mscomp.h:99,  src/mscomp.cpp:96-100      -> ms_max_compressed_size
 *   lznt1_compress            include/lznt1.h:49,   src/lznt1_compress.cpp:233 -> lznt1_compress
 *   lznt1_max_compressed_size include/lznt1.h:50,   src/lznt1_compress.cpp:27  -> lznt1_max_compressed_size
 *   xpress_compress           include/xpress.h:47,  src/xpress_compress.cpp:240-> xpress_compress
 *   xpress_max_compressed_size include/xpress.h:48, src/xpress_compress.cpp:30 -> xpress_max_compressed_size
 *   xpress_huff_compress      include/xpress_huff.h:46, src/xpress_huff_compress.cpp:247 -> xpress_huff_compress
 *   xpress_huff_max_compressed_size include/xpress_huff.h:47, src/xpress_huff_compress.cpp:46 -> (same name)
 *
 * These take HOST pointers (the reference contract). Every byte of output is produced by HIP kernels on
 * the current device; there is no CPU encoder in this library -- if no usable GPU/HIP runtime is present
 * the calls return MSCOMP_ERRNO and never fall back.
 *
 * Part 1 also holds the rows SURVEY.md 8f lists next, each declared below with the reference file:line it replaces:
 * the one-shot decompressors (ms_decompress, lznt1_decompress, xpress_decompress, xpress_huff_decompress) and the
 * LZNT1 streaming compressor / decompressor with the reference's stream object (ms_deflate*, ms_inflate*, lznt1_deflate*,
 * lznt1_inflate*). Same rule: the bytes come from the GPU, there is no CPU codec to fall back to.
 *
 * Part 2 is the additive batch interface the GPU needs (SURVEY.md 8b "batch extension"): many independent
 * units (buffers) already resident in HBM, compressed (or decompressed) in one pass, output written to HBM; plus
 * capacity planning and device-side compaction of a batch's outputs.
 *
 * Plain C types only (no torch / HIP types in any signature; a hipStream_t is passed as void*).
 */
#ifndef MSCOMP_AMD_H
#define MSCOMP_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums: values identical to include/mscomp/general.h:65-93 of the reference ---- */
typedef enum _MSCompFormat {
	MSCOMP_NONE = 0, MSCOMP_RESERVED = 1, MSCOMP_LZNT1 = 2, MSCOMP_XPRESS = 3, MSCOMP_XPRESS_HUFF = 4
} MSCompFormat;
typedef enum _MSCompStatus {
	MSCOMP_OK = 0, MSCOMP_STREAM_END = 1, MSCOMP_POSSIBLE_STREAM_END = 2,
	MSCOMP_ERRNO = -1, MSCOMP_ARG_ERROR = -2, MSCOMP_DATA_ERROR = -3, MSCOMP_MEM_ERROR = -4, MSCOMP_BUF_ERROR = -5
} MSCompStatus;

/* ================= Part 1: drop-in one-shot interface (host pointers) ================= */
MSCompStatus ms_compress(MSCompFormat format, const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len);
size_t       ms_max_compressed_size(MSCompFormat format, size_t in_len);

MSCompStatus lznt1_compress(const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len);
size_t       lznt1_max_compressed_size(size_t in_len);
MSCompStatus xpress_compress(const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len);
size_t       xpress_max_compressed_size(size_t in_len);
MSCompStatus xpress_huff_compress(const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len);
size_t       xpress_huff_max_compressed_size(size_t in_len);

/* Decompressors. Same contract as the reference: *out_len holds the capacity on entry and the number of
 * bytes produced on MSCOMP_OK; MSCOMP_BUF_ERROR when the output (or what is left of the input) does not fit,
 * MSCOMP_DATA_ERROR for a malformed stream.
 *   ms_decompress      include/mscomp.h:88,   src/mscomp.cpp:119-134               -> ms_decompress
 *   lznt1_decompress   include/lznt1.h:51,    src/lznt1_decompress.cpp:293 (the inflate wrapper, internal.h:616-630) -> lznt1_decompress
 *   xpress_decompress  include/xpress.h:50,   src/xpress_decompress.cpp:405                -> xpress_decompress
 *   xpress_huff_decompress include/xpress_huff.h:49, src/xpress_huff_decompress.cpp:130      -> xpress_huff_decompress */
MSCompStatus ms_decompress(MSCompFormat format, const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len);
MSCompStatus lznt1_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len);
MSCompStatus xpress_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len);
MSCompStatus xpress_huff_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len);

/* Streaming compression: the reference's stream object and its LZNT1 streaming compressor -- the only format
 * whose ms_deflate works in the reference. Layout of mscomp_stream = include/mscomp/general.h:95-121 in the default build
 * (MSCOMP_WITH_ERROR_MESSAGES and MSCOMP_WITH_WARNING_MESSAGES, include/mscomp/config.h:54-63). Same call protocol and statuses
 * (include/mscomp.h:102-158); the bytes are those of the reference for every way of slicing the input and the output windows.
 *   ms_deflate_init / ms_deflate / ms_deflate_end          include/mscomp.h:114,143,158, src/mscomp.cpp:136-165 (MSCOMP_NONE, MSCOMP_LZNT1)
 *   lznt1_deflate_init / lznt1_deflate / lznt1_deflate_end include/lznt1.h:55-57,       src/lznt1_compress.cpp:132-231 */
typedef enum _MSCompFlush { MSCOMP_NO_FLUSH = 0, MSCOMP_FLUSH = 2, MSCOMP_FINISH = 4 } MSCompFlush;
typedef struct _mscomp_internal_state mscomp_internal_state;
typedef struct _mscomp_stream {
	MSCompFormat format;
#ifdef __cplusplus
	bool compressing;
#else
	int compressing;
#endif
	const uint8_t* in;  size_t in_avail,  in_total;
	uint8_t*       out; size_t out_avail, out_total;
	char error[256];
	char warning[256];
	mscomp_internal_state* state;
} mscomp_stream;
MSCompStatus ms_deflate_init(MSCompFormat format, mscomp_stream* stream);
MSCompStatus ms_deflate(mscomp_stream* stream, MSCompFlush flush);
MSCompStatus ms_deflate_end(mscomp_stream* stream);
MSCompStatus lznt1_deflate_init(mscomp_stream* stream);
MSCompStatus lznt1_deflate(mscomp_stream* stream, MSCompFlush flush);
MSCompStatus lznt1_deflate_end(mscomp_stream* stream);
/* The Xpress streaming compressor is unfinished in the reference; its three entry points exist and return fixed statuses in the default
 * build (include/xpress.h:52-54; src/xpress_compress.cpp:52-73 MSCOMP_MEM_ERROR, :74-218 MSCOMP_ARG_ERROR, :219-235 stream check). Exported
 * with the same statuses so that a program naming them links against the drop-in and sees what it saw before. */
MSCompStatus xpress_deflate_init(mscomp_stream* stream);
MSCompStatus xpress_deflate(mscomp_stream* stream, MSCompFlush flush);
MSCompStatus xpress_deflate_end(mscomp_stream* stream);
/* ... and its streaming decompressor: ms_inflate_init / ms_inflate / ms_inflate_end (include/mscomp.h:174,198,213, src/mscomp.cpp:167-196;
 * MSCOMP_NONE and MSCOMP_LZNT1) and lznt1_inflate_init / lznt1_inflate / lznt1_inflate_end (include/lznt1.h:59-61,
 * src/lznt1_decompress.cpp:210-290). Every chunk is decoded on the GPU. */
/* xpress_inflate_init / xpress_inflate / xpress_inflate_end (include/xpress.h:56-58, src/xpress_decompress.cpp:45-403) are NOT offloaded (one
 * stream is a serial token chain, handed over piecewise): the symbols exist so that programs naming them link; xpress_inflate_init -- and
 * ms_inflate_init(MSCOMP_XPRESS) -- returns MSCOMP_MEM_ERROR without touching the stream, the other two MSCOMP_ARG_ERROR. Build with
 * -DMSCOMP_AMD_NO_XPRESS_INFLATE and keep the reference's xpress_decompress.cpp in the link where streaming Xpress decompression is needed. */
MSCompStatus xpress_inflate_init(mscomp_stream* stream);
MSCompStatus xpress_inflate(mscomp_stream* stream);
MSCompStatus xpress_inflate_end(mscomp_stream* stream);
MSCompStatus ms_inflate_init(MSCompFormat format, mscomp_stream* stream);
MSCompStatus ms_inflate(mscomp_stream* stream);
MSCompStatus ms_inflate_end(mscomp_stream* stream);
MSCompStatus lznt1_inflate_init(mscomp_stream* stream);
MSCompStatus lznt1_inflate(mscomp_stream* stream);
MSCompStatus lznt1_inflate_end(mscomp_stream* stream);

/* ================= Part 2: batch interface (device pointers) ================= */
typedef struct mscomp_amd_ctx  mscomp_amd_ctx;    /* one per (device, stream); owns scratch in HBM  */
typedef struct mscomp_amd_plan mscomp_amd_plan;   /* unit layout of one batch, uploaded once        */

/* device = HIP ordinal; hip_stream = hipStream_t to launch on (NULL = the null stream).
 * A context belongs to ONE host thread at a time (its scratch, its pinned table staging and its pool of table buffers are not locked): use one
 * context per thread -- the one-shot entries of Part 1 do that themselves (a thread-local context), mscomp_amd_compress_units_host keeps one per
 * sub-batch slot. mscomp_amd_plan_create must not run while the context's stream is being captured into a graph (it waits on an event of that
 * stream when its table staging is still in flight); capture mscomp_amd_plan_execute instead, with the plan made beforehand. */
MSCompStatus mscomp_amd_ctx_create(int device, void* hip_stream, mscomp_amd_ctx** ctx);
void         mscomp_amd_ctx_destroy(mscomp_amd_ctx* ctx);

/* A batch is n_units independent buffers resident in HBM. Unit i is (all four arrays: n_units entries, host memory)
 *   input    d_in  + in_off[i]  , in_len[i]  bytes
 *   output   d_out + out_off[i] , out_cap[i] bytes of capacity
 * and is compressed exactly as one ms_compress(format, ...) call would compress it:
 *   LZNT1        4 KiB chunks inside the unit, End_of_buffer 00 00 appended when capacity allows
 *   XPRESS       one Xpress stream per unit
 *   XPRESS_HUFF  64 KiB chunks inside the unit, matches reach into the previous chunk, EOS in the last
 * Units must not overlap on the output side. For best load/store width keep in_off[i] and out_off[i] multiples
 * of 16 (any alignment is accepted). */
MSCompStatus mscomp_amd_plan_create(mscomp_amd_ctx* ctx, MSCompFormat format, size_t n_units,
                                    const uint64_t* in_off, const uint64_t* in_len,
                                    const uint64_t* out_off, const uint64_t* out_cap, mscomp_amd_plan** plan);
void         mscomp_amd_plan_destroy(mscomp_amd_plan* plan);

/* Asynchronous on the ctx stream. d_out_len[i] (device, uint64) receives the compressed size of unit i,
 * d_status[i] (device, int32) MSCOMP_OK or MSCOMP_BUF_ERROR (unit did not fit its capacity; its output
 * bytes are then unspecified). Returns MSCOMP_OK when everything was enqueued, MSCOMP_ERRNO on a HIP error. */
MSCompStatus mscomp_amd_plan_execute(mscomp_amd_plan* plan, const uint8_t* d_in, uint8_t* d_out,
                                     uint64_t* d_out_len, int32_t* d_status);

/* SURVEY.md 8e "Multi-GPU" + 8f-3 "host pipeline" (no counterpart in the reference: ms_compress, mscomp.h:59 / src/mscomp.cpp:113-117, is one
 * buffer per call on one thread): n independent units given by HOST pointers, compressed on n_dev GPUs of this process. Unit i is exactly one
 * ms_compress(format, in_ptrs[i], in_lens[i], out_ptrs[i], &out_lens[i]) call with *out_len = out_caps[i] on entry: same bytes, statuses[i] =
 * MSCOMP_OK or MSCOMP_BUF_ERROR (out_lens[i] = 0 and nothing written behind the capacity then), the uncounted LZNT1 00 00 behind the stream
 * when the capacity has room. Bytes of a unit's capacity behind its stream are unspecified afterwards.
 * devices = n_dev device ordinals (NULL: 0 .. n_dev - 1; the same ordinal may appear twice: two ranges share that GPU). The units are cut into
 * n_dev contiguous ranges with near-equal input bytes (no exchange step, no collective: units are independent); every range runs on its own
 * host thread as a pipeline of sub-batches (MSCOMP_AMD_HOST_BATCH_MB MiB of input each, default 32 -- 96 for Xpress+Huffman --, up to
 * MSCOMP_AMD_HOST_SLOTS = 8 in flight, each with its own context and stream; an uploader and a downloader thread beside it); units
 * that lie back to back in the caller's memory travel as one copy. Returns MSCOMP_OK when every range ran (per-unit results in statuses),
 * MSCOMP_ARG_ERROR for a bad format / device / pointer, MSCOMP_MEM_ERROR / MSCOMP_ERRNO when a range could not run (statuses of units not
 * reached stay MSCOMP_ERRNO; a failed allocation or a thread the system refuses is MSCOMP_MEM_ERROR, never an abort: csrc/hostbatch.hip is built
 * with exceptions and catches at this boundary). ONE LZNT1 unit of 36 MiB or more on the calling thread's range takes the one-shot call's path
 * (caller buffers mapped into the GPU's address space; the two environment knobs do not apply to it). With outputs laid out capacity after
 * capacity, the bytes of a capacity behind its stream -- including the capacities of units that failed -- are unspecified afterwards (they come
 * down with the streams in one copy). Contexts and staging are kept per device between calls; mscomp_amd_host_pool_release() frees them. */
MSCompStatus mscomp_amd_compress_units_host(MSCompFormat format, int n_dev, const int* devices, size_t n_units,
                                            const uint8_t* const* in_ptrs, const size_t* in_lens, uint8_t* const* out_ptrs, const size_t* out_caps,
                                            size_t* out_lens, MSCompStatus* statuses);
/* The same for the decoders: unit i is one ms_decompress call (mscomp.h:88) with *out_len = out_caps[i] on entry; statuses[i] =
 * MSCOMP_OK / MSCOMP_BUF_ERROR / MSCOMP_DATA_ERROR exactly as the reference's one-shot decoder returns them (DESIGN_DECODERS.md). */
MSCompStatus mscomp_amd_decompress_units_host(MSCompFormat format, int n_dev, const int* devices, size_t n_units,
                                              const uint8_t* const* in_ptrs, const size_t* in_lens, uint8_t* const* out_ptrs, const size_t* out_caps,
                                              size_t* out_lens, MSCompStatus* statuses);
void         mscomp_amd_host_pool_release(void);

/* Convenience: create plan + execute + stream-synchronize + destroy. */
MSCompStatus mscomp_amd_compress_batch(mscomp_amd_ctx* ctx, MSCompFormat format, size_t n_units,
                                       const uint8_t* d_in, const uint64_t* in_off, const uint64_t* in_len,
                                       uint8_t* d_out, const uint64_t* out_off, const uint64_t* out_cap,
                                       uint64_t* d_out_len, int32_t* d_status);

/* The same for decompression: unit i holds one compressed buffer (what one ms_decompress call takes), out_cap[i] is the
 * capacity the caller passes in *out_len. d_status[i] is MSCOMP_OK / MSCOMP_BUF_ERROR / MSCOMP_DATA_ERROR exactly as the
 * reference's one-shot call returns them; d_out_len[i] is the decompressed size on MSCOMP_OK (0 otherwise; the output bytes
 * of a failed unit are unspecified). Executed with mscomp_amd_plan_execute. Units are limited to 4 GiB - 4096 of input. */
MSCompStatus mscomp_amd_plan_create_decompress(mscomp_amd_ctx* ctx, MSCompFormat format, size_t n_units,
                                               const uint64_t* in_off, const uint64_t* in_len,
                                               const uint64_t* out_off, const uint64_t* out_cap, mscomp_amd_plan** plan);
MSCompStatus mscomp_amd_decompress_batch(mscomp_amd_ctx* ctx, MSCompFormat format, size_t n_units,
                                         const uint8_t* d_in, const uint64_t* in_off, const uint64_t* in_len,
                                         uint8_t* d_out, const uint64_t* out_off, const uint64_t* out_cap,
                                         uint64_t* d_out_len, int32_t* d_status);

/* Batch helpers.
 * Capacity planning: out_cap[i] = what one ms_compress call needs at most for in_len[i] bytes (ms_max_compressed_size, + 2 for the LZNT1
 * End_of_buffer), out_off[i] = running offset rounded up to `align`; returns the total size of the output buffer ((uint64_t)-1: bad format).
 * Either output array may be NULL.
 * Compaction: the outputs of a batch sit at out_off[i] with gaps up to their capacities; this packs them back to back, in unit order,
 * into d_packed and writes the n_units + 1 offsets (uint64, device memory) to d_packed_off; out_off / out_cap are the host arrays given to
 * the plan, d_out_len the device array plan_execute filled (units with a status other than MSCOMP_OK have length 0). Enqueued on the ctx
 * stream after one small synchronous table upload. */
uint64_t     mscomp_amd_plan_layout(MSCompFormat format, size_t n_units, const uint64_t* in_len, uint64_t align, uint64_t* out_off, uint64_t* out_cap);
MSCompStatus mscomp_amd_compact_batch(mscomp_amd_ctx* ctx, size_t n_units, const uint8_t* d_out, const uint64_t* out_off, const uint64_t* out_cap,
                                      const uint64_t* d_out_len, uint8_t* d_packed, uint64_t* d_packed_off);

/* ---- measurement hooks (bench.py / profiles) ---- */
/* When enabled, every kernel launch of plan_execute is bracketed by hipEvents on the ctx stream. */
void         mscomp_amd_profile_enable(mscomp_amd_ctx* ctx, int on);
/* Synchronizes the stream, then returns the number of distinct kernels seen since the last reset and fills
 * up to cap entries: name (static string), accumulated milliseconds, launch count. Resets the counters. */
int          mscomp_amd_profile_read(mscomp_amd_ctx* ctx, const char** names, double* ms, uint64_t* launches, int cap);
/* Stage-level test hook: per-position matches (len-3 capped at 45, offset; 0 = no match) of ONE device-resident buffer as
 * found by the HIP hash-chain match finder. max_off = 0x2000 (Xpress) / 0xFFFF with clip=1 (Xpress+Huffman). */
MSCompStatus mscomp_amd_debug_xpress_matches(mscomp_amd_ctx* ctx, const uint8_t* d_in, size_t in_len, uint32_t max_off, int clip,
                                             uint16_t* h_len3, uint16_t* h_off);
/* Stage-level test hook: the code lengths HuffmanEncoder<15,512>::CreateCodes (include/mscomp/HuffmanEncoder.h:58-107, the heap build with
 * its > 15-bit rescale loop) gives for n histograms of 512 counts; h_counts (n x 512 uint32) and h_lens (n x 512 bytes) are host arrays. */
MSCompStatus mscomp_amd_debug_huff_lengths(mscomp_amd_ctx* ctx, const uint32_t* h_counts, size_t n, uint8_t* h_lens);
/* The mscomp_amd_debug_set_* hooks below choose between BIT-IDENTICAL kernels for the tests and are process-wide; the library ignores them unless
 * MSCOMP_AMD_TEST_HOOKS=1 was in the environment when it was loaded (1 = they work). A deployment never sets it. */
int          mscomp_amd_debug_hooks_enabled(void);
/* Test hook: the Xpress parse/emit stage has two bit-identical kernels (one wave per unit; four
 * or sixteen waves per unit with speculative segments). 0 = chosen by batch size (default), 1 / 2 / 3 = force. Process-wide. */
void         mscomp_amd_debug_set_xpress_emit(int mode);
/* The reference has two LZNT1 dictionaries, chosen when it is BUILT (/root/reference/include/mscomp/config.h:83-88): the default one and,
 * with -DMSCOMP_WITH_LZNT1_SA_DICT, a suffix-array one (/root/reference/include/mscomp/LZNT1Dictionary_SA.h) whose matches have the same
 * lengths but other offsets -- so the compressed bytes differ. A deployment that replaces such a build selects the same flavour here.
 * The flavour is a property of a PLAN, fixed when the plan is created: from its context's setting (mscomp_amd_ctx_set_lznt1_sa_dict: 1 / 0, -1 =
 * follow the process default) or else from the process default (mscomp_amd_set_lznt1_sa_dict, or MSCOMP_AMD_LZNT1_SA_DICT=1 in the environment
 * when the library loads). Changing a setting never touches a plan that exists. The entries without a context argument (ms_compress, ms_deflate,
 * mscomp_amd_compress_units_host) make their plans per call and follow the process default of that moment. Decompression is not affected. */
void         mscomp_amd_set_lznt1_sa_dict(int on);
int          mscomp_amd_get_lznt1_sa_dict(void);
MSCompStatus mscomp_amd_ctx_set_lznt1_sa_dict(mscomp_amd_ctx* ctx, int on);
/* Test hook: Xpress decompression has two bit-identical paths: 0 = default (32-bit tokens, a flag word per step, then the copy kernels that
 * Xpress+Huffman uses), 1 = one wave per stream taking a token per step and moving the bytes itself (round 1's kernel). Process-wide. */
void         mscomp_amd_debug_set_xpress_decoder(int mode);
/* Test hook: after a decompression whose large units (capacity >= 1 MiB) got their bytes from csrc/lzglobal.hip: out[0..32] = words still
 * pointing after each pointer pass (`words` = sum over those units of capacity + 64). 0 = read. */
int          mscomp_amd_debug_lzg_open(mscomp_amd_ctx* ctx, uint64_t words, uint32_t* out);
/* Test hook: how the Xpress match finder runs. 1 = default: batches whose units are at most 64 KiB run Find only where a greedy parse can
 * start a token (csrc/xpress_lazy.hip); 2 = Find for every position everywhere (xp_find_kernel, what longer streams and Xpress+Huffman
 * always use). The parse kernels get the same answers on every path they walk. Process-wide. */
void         mscomp_amd_debug_set_finder(int mode);
/* Test hook: how a host-pointer ms_compress(MSCOMP_LZNT1) of a large buffer (>= 36 MiB) runs. 0 = default (the caller's buffers are mapped
 * into the GPU's address space, one launch; falls back to 1 when the mapping fails), 1 = always in slices on three streams
 * (csrc/api.hip lznt1_compress_pipelined; the same as MSCOMP_AMD_ONE_ZEROCOPY=0 in the environment). Same bytes and statuses. Process-wide. */
void         mscomp_amd_debug_set_one_shot(int mode);
/* Test hook: the LZNT1 chunk stage has two bit-identical kernels (one wave / four waves per 4 KiB chunk). 0 = default, 1 / 2 = force. */
void         mscomp_amd_debug_set_lznt1(int mode);
/* Test hook: LZNT1 decompression finds the chunk headers by walking speculated chains per 48 KiB segment of the input; a segment
 * whose speculation held nothing usable is walked again by one lane. Returns how many segments that happened to since the last call
 * (synchronizes the stream). */
uint32_t     mscomp_amd_debug_lzd_walked(mscomp_amd_ctx* ctx);
/* Hardware self-check: the LZNT1 bucket sort and the Xpress chain links rely on gfx950 serving the returning
 * same-address LDS atomics of one wave instruction in lane order. Returns the number of lanes (over blocks x rounds x 64
 * lanes x {add, exchange}, keys drawn from nkeys <= 2048 values) that were served out of order: 0 on gfx950;
 * 0xFFFFFFFF if the check could not run. */
uint32_t     mscomp_amd_debug_lds_lane_order(mscomp_amd_ctx* ctx, uint32_t seed, uint32_t blocks, uint32_t rounds, uint32_t nkeys);
/* A device that serves them in another order is not refused: the two kernels then issue that atomic one lane at a time (the same bytes, a
 * slower sort; csrc/kernels.h). Test hook: 1 = run that order-independent form on every device, 0 = back to the default. Process-wide. */
void         mscomp_amd_debug_set_serial_atomics(int on);
/* Version / build string of the library (includes the gfx target it was compiled for). */
const char*  mscomp_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif

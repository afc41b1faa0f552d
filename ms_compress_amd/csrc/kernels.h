// kernels.h -- host-callable launchers of the gfx950 kernels (internal to libmscomp_amd.so).
#pragma once
#include "common.h"
#include <atomic>

namespace msc {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: every launcher keeps one of these per kernel and sets
// the attribute the first time it launches on a device (bit = device ordinal; two host threads racing set it twice, which is
// harmless). A process-wide `static bool` left the second GPU of a process without the attribute.
struct PerDeviceOnce {
	std::atomic<uint64_t> seen[4] = {};                      // 256 device ordinals
	bool needed() const { int d = 0; (void)hipGetDevice(&d); return !((seen[(d >> 6) & 3].load(std::memory_order_acquire) >> (d & 63)) & 1u); }
	void done() { int d = 0; (void)hipGetDevice(&d); seen[(d >> 6) & 3].fetch_or(1ull << (d & 63), std::memory_order_release); }
};

// ---- lane order of same-address LDS atomics (util.hip) ----
// The LZNT1 bucket sort and the Xpress chain links hand out slots / links with ONE returning LDS atomic per 64 positions and need the
// same-address atomics of that instruction served in lane order -- what gfx950 does, checked once per device (api.hip lane_order_verdict). On
// a device that fails the check (or when the test hook says so) the same kernels issue the atomic one lane at a time, in a 64-step loop: DS
// operations of a wave execute in program order, which IS architectural. Same bytes, slower sort.
bool serial_atomics_on_current_device();
void set_serial_atomics(int device, int on);           // on: 1 = one lane at a time. device >= 0: that device's self-check verdict; device < 0: the process-wide test hook (its 0 does NOT clear a verdict)

// ---- LZNT1 (lznt1.hip) ----
#ifdef LZ_TBL_GLOBAL         /* dev variant (round 6): a 12-bit hash whose bucket-end table (8 KiB) leaves LDS after the sort -- it lies behind the chunk's image in its slot */
#define LZNT1_SLOT (4352u + 8192u)
#else
#define LZNT1_SLOT 4352u     // scratch bytes per 4 KiB chunk image (2 B header + <=4096 B payload + emit slack)
#endif
void set_lznt1_mode(int mode);
void launch_lznt1_chunks(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint8_t* slots, uint32_t* slot_size);
void launch_lznt1_sa_chunks(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint8_t* slots, uint32_t* slot_size);   // lznt1_sa.hip: the suffix-array dictionary flavour

// ---- Xpress / Xpress+Huffman match finder (xpress_match.hip) ----
// links: u16 per position (64 KiB "link chunks", chunk-major), lasthead: 32768 u16 per link chunk,
// mlen3/moff: per position len-3 (capped at 48-3) and offset (0 = no match).
void launch_xp_links(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint16_t* links, uint16_t* lasthead);
void launch_xp_find(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* links, const uint16_t* lasthead,
                    uint16_t* mlen3, uint16_t* moff, uint32_t max_off, int clip);
// the same over chunks [chunk_base, chunk_base + chunk_count) only (the pipelined one-shot call: a range's kernels run while the next range is on its way up)
void launch_xp_links_range(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint16_t* links, uint16_t* lasthead, uint32_t chunk_base, uint32_t chunk_count);
void launch_xp_find_range(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* links, const uint16_t* lasthead,
                          uint16_t* mlen3, uint16_t* moff, uint32_t max_off, int clip, uint32_t chunk_base, uint32_t chunk_count);
// the lazy finder (xpress_lazy.hip): Find only where a greedy parse can start a token -- Xpress, every unit at most 64 KiB. Fills the same
// arrays for a superset of the true token starts; the offsets of all other positions are 0.
void launch_xp_lazy2(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* links, uint16_t* mlen3, uint16_t* moff);

#ifdef MSCOMP_AMD_DEV
// MEASUREMENT MODES, in the development flavour of the library only (libmscomp_amd_dev.so, `make dev`): two more Xpress+Huffman match finders that
// give the same bytes as xp_find_kernel and lose to it. The product library has ONE finder per codec and does not contain them.
// the lazy finder for Xpress+Huffman (xhuff_lazy.hip, round 5; a measurement mode, MSCOMP_AMD_XH_LAZY=1): the chunk's links in LDS, candidate bytes from L2
void launch_xh_lazy(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* links, const uint16_t* lasthead, uint16_t* mlen3);

// the finder without the dependent chain walk (xpress_sort.hip, round 5): positions sorted by (hash, position) per 64 KiB chunk, a position's
// candidates read as one span of that array. sorted: u16 per position behind XS_FRONT_PAD entries of padding (the caller passes the padded
// pointer); starts: XS_STARTS_STRIDE u32 per chunk (32768 bucket starts + the total); words: u32 per position -- xp_sort_kernel leaves
// index | rank << 16 there, xp_find2_kernel replaces it by the match word (the mlen3 / moff array of the other finders).
#define XS_STARTS_STRIDE 32832u
#define XS_FRONT_PAD 32u
void launch_xp_sort(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint16_t* sorted, uint32_t* words, uint32_t* starts);
void launch_xp_find2(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* sorted, const uint32_t* starts, uint32_t* words, uint32_t max_off, int clip);
#endif

// ---- Xpress stream emission (xpress_emit.hip): one wavefront per unit ----
void set_xpress_emit_mode(int mode);
int xpress_emit_mode_for(uint32_t n_units, uint32_t n_chunks);
// per-window / per-super-block records of the Xpress parse (see xpress_emit.hip); the last 10 only for the block-per-super-block kernels
struct XpressWinBufs { u64* wtok; u64* wmat; uint32_t* wfar; uint32_t* wecur; uint32_t* weF; uint32_t* wsum; uint32_t* wnr; uint32_t* ws0; uint32_t* ws1;
                       uint32_t* sbtot; u64* sbpre; uint32_t* seam; u64* seampos; uint32_t* used; };
void launch_xpress_emit(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint16_t* mlen3, const uint16_t* moff,
                        const XpressWinBufs& wb, uint8_t* d_out, u64* d_out_len, int32_t* d_status);

// ---- Xpress+Huffman chunk pipeline (xhuff.hip) ----
void launch_xh_parse(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint16_t* mlen3, const uint16_t* moff,
                     u64* tokbits, uint32_t* counts, uint32_t* extra);
void launch_xh_huff(hipStream_t st, const BatchTables& bt, const uint32_t* counts, const uint32_t* extra, uint8_t* lens, uint16_t* codes,
                    uint32_t* chunk_size, uint32_t* fb_list, uint32_t* fb_count, uint32_t* fbflag);
void launch_xh_huff_debug(hipStream_t st, const uint32_t* counts, uint8_t* lens, uint32_t n);
void launch_xh_fallback(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint32_t* fb_list, const uint32_t* fb_count,
                        uint32_t blocks, u64* tokbits, uint8_t* lens, uint16_t* codes, uint32_t* chunk_size);
void launch_xh_encode(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* mlen3, const uint16_t* moff,
                      const u64* tokbits, const uint8_t* lens, const uint16_t* codes, const uint32_t* fbflag, const u64* prefix, uint8_t* d_out);

// ---- decompressors (decompress.hip) ----
// LZNT1: the compressed input of a unit is cut into segments of LZD_SEG bytes (chunk_prefix[u] = first segment of unit u, n_chunks =
// segments of the batch). cin: header offsets, LZD_SLOTS per segment (u32); segL / segE / segcnt / segstop / segoff per speculated
// chain; selcnt / seloff per segment (the true chain); flat: u64 exclusive scan of selcnt (n_chunks + 1) = number of the first chunk of a segment; csize: decoded size or 0x8000 per
// chunk number (u16); stop / irregular (+1 global flag) per unit.
#define LZD_SEG   49152u
#define LZD_HEAD  12288u
#define LZD_SLOTS (LZD_SEG / 3u + 2u)
#define LZD_K     8u         // speculated chains kept per segment
struct LzdBufs { uint32_t* cin; uint16_t* csize; uint32_t* segL; uint32_t* segE; uint32_t* segcnt; uint32_t* segstop; uint32_t* segoff;   /* LZD_K per segment */
                 uint32_t* selcnt; uint32_t* seloff; uint32_t* stop; uint32_t* irregular; u64* flat; };
void launch_lzd_segments(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const LzdBufs& b);
uint32_t lzd_read_walked();
void launch_lzd_verify(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const LzdBufs& b);
void launch_lzd_chunks(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const LzdBufs& b, uint8_t* d_out, int exact);
void launch_lzd_finalize(hipStream_t st, const BatchTables& bt, const LzdBufs& b, u64* d_out_len, int32_t* d_status);

// Xpress: one wave per stream
void launch_xpress_decompress(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint8_t* d_out, u64* d_out_len, int32_t* d_status);
// large Xpress streams by segments (decompress.hip, xps_*): plan tables + scratch
#define XPS_SEG    16384u                                 // input bytes per segment (12 whole files, segment / warm-up KiB: 32/32 15.5 ms, 16/16 14.1, 8/8 19.6, 32/8 21.2)
#define XPS_WARM   16384u                                 // a speculative walk starts this far before its segment
#define XPS_SEG_BYTES 64u
#define XPS_MIN_IN (512u << 10)                           // streams with at least this much input
#define XPS_ROUNDS 8u                                     // rounds of "walk the segments again that do not hold"
struct XpsTables {
	const uint32_t* unit;          // n_big: the streams taken
	const u64* seg_prefix;         // n_big + 1: first segment of each
	void*     seg;                 // per segment: 64 bytes of state (XpsSeg)
	uint32_t* mode;                // per stream taken: 1 rounds running, 2 done by segments, 0 left to the one-wave walk
	uint32_t* done;                // per unit of the batch: XPS_DONE when the one-wave walk has nothing to do
	uint32_t  n_big, n_seg;
	uint32_t  seg_bytes, warm_bytes; // input bytes per segment; a speculative walk starts this far before its segment (<= seg_bytes)
};
// the same through 32-bit tokens: phase -1 = the segment kernels, 0 = xpt_parse_kernel (a flag word at a time), 1 = lz_copy_kernel, 2 = lz_copy_block_kernel
void launch_xpress_decompress_tokens(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const u64* tok_prefix, uint32_t* tok, u64* ntok,
                                     uint8_t* d_out, u64* d_out_len, int32_t* d_status, int phase, u64 lzg_min_cap, const XpsTables& x);

// Xpress+Huffman, in phases: 0 mark candidate chunk starts, 1 walk every candidate as one chunk, 2 chain check per buffer, 3 tokens of the
// accepted chunks, 4 serial walk of the buffers the speculation could not do, 5 tokens -> bytes. tok_prefix[u] = first token slot of unit u,
// cand_prefix[u] = first candidate slot (n_slots in all); per candidate slot: offset, next offset, reach, state, bytes, tokens, token offset.
#define XHC_TILE_BYTES 16384u
enum { XHC_SERIAL = 1, XHC_SPEC = 2 };
struct XhcBufs { uint32_t* cand_cnt; uint32_t* mode; uint32_t* cand_pos; uint32_t* res_end; uint32_t* res_reach; uint32_t* res_state;
                 u64* res_prod; u64* res_ntok; u64* tok_off;
                 const u64* scr_prefix; uint32_t* scr_tok; };   // token scratch: first scratch slot of every unit (n_units + 1; equal = none), XHC_SCR tokens per candidate
#define XHC_SCR 65600u                                    // a candidate gives up to 65536 tokens before its 65536th byte
void launch_xpress_huff_decompress(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const u64* tok_prefix, uint32_t* tok, u64* ntok,
                                   const u64* cand_prefix, uint32_t n_slots, const XhcBufs& xb,
                                   uint8_t* d_out, u64* d_out_len, int32_t* d_status, int phase, u64 lzg_min_cap);

// ---- tokens -> bytes for large units by all CUs (lzglobal.hip) ----
#define LZG_PASSES 33u                                    // pointer passes launched (chains halve at least: 2^32 bytes); a pass returns at once when the one before left nothing open
#ifndef LZG_TILE_SHIFT
#define LZG_TILE_SHIFT 13                                   // output bytes per tile of lzg_expand_kernel: 8 KiB (32 KiB tiles: expansion 0.78 -> 1.37 ms, passes 4.1 -> 4.6 ms on the 12 files)
#endif
#define LZG_MIN_CAP (1u << 20)                            // units with at least this much output capacity take this path (when the plan has the scratch for it)
struct LzgTables {
	const uint32_t* unit;          // n_big: the units taken
	const u64* tb_prefix;          // n_big + 1: first token block (8192 tokens) of each
	const u64* tile_prefix;        // n_big + 1: first tile (8192 output bytes) of each
	const u64* word_prefix;        // n_big + 1: first word of each in `words`
	u64*      bsum;                // per token block: bytes its tokens give; then the bytes before it
	uint32_t* dir_tok;             // per tile: the first token that starts in the tile or behind it ...
	uint32_t* dir_pos;             // ... and where
	uint32_t* words;               // per output byte: LZG_VAL | byte, or the index of its source
	uint32_t* open;                // LZG_PASSES counters: words still pointing after each pass
	uint8_t*  tile_pass;           // per tile: the pointer pass that has to look at it next (0xFF: none)
	uint32_t  n_big, n_tb, n_tiles;
};
// phase 0 = directory (3 kernels), 1 = lzg_expand_kernel, 2 = the pointer passes
void launch_lz_copy_global(hipStream_t st, const LzgTables& g, const BatchTables& bt, const u64* tok_prefix, const uint32_t* tok, const u64* ntok,
                           const u64* d_out_len, const int32_t* d_status, uint8_t* d_out, int phase);

// ---- utilities (util.hip) ----
// prefix[0..n] = exclusive scan of sizes[0..n) as u64 (prefix[n] = total). block_sums: scratch of ceil(n/1024)+1 u64.
void launch_scan_sizes(hipStream_t st, const uint32_t* sizes, u64* prefix, uint32_t n, u64* block_sums);
// Concatenate the chunk images of every unit into the caller's output (skips units that do not fit).
void launch_concat_slots(hipStream_t st, const uint8_t* slots, uint32_t slot_stride, const uint32_t* slot_size,
                         const u64* prefix, const BatchTables& bt, uint8_t* d_out);
// Per unit: out_len, status (OK / BUF_ERROR); LZNT1 also appends the uncounted 00 00 End_of_buffer when room.
// outputs of a batch packed back to back: packed_off[0..n] (device), bytes of unit u at d_packed + packed_off[u]
void launch_compact(hipStream_t st, const uint8_t* d_out, const u64* d_out_off, const uint32_t* d_tile_prefix, uint32_t n_units, uint32_t n_tiles,
                    const u64* d_out_len, u64* d_packed_off, uint8_t* d_packed);
uint32_t run_lds_lane_order_check(hipStream_t st, uint32_t seed, uint32_t blocks, uint32_t rounds, uint32_t nkeys, uint32_t* d_bad);
void launch_finalize_units(hipStream_t st, const u64* prefix, const BatchTables& bt, uint8_t* d_out,
                           u64* d_out_len, int32_t* d_status, int lznt1_eob);

} // namespace msc

// decompress.hip -- gfx950 decompressors, batch form: n independent units resident in HBM.
//
//   LZNT1           lzd_seg_kernel -> lzd_verify_kernel -> scan -> lzd_chunk_kernel<false> -> lzd_finalize_kernel -> lzd_chunk_kernel<true>
//                   (chunk-parallel; the header chain and the output offsets are speculated and verified)        [this comment]
//   Xpress          xpd_kernel            one wave walks and copies one stream (8 KiB window in an LDS ring)      [comment at the kernel]
//   Xpress+Huffman  xhc_mark_kernel -> xhc_parse_kernel<1> -> xhc_chain_kernel -> xhc_parse_kernel<2>: the chunks of a buffer found
//                   speculatively and walked in parallel, one wave per chunk, writing 32-bit tokens              [comments at the kernels]
//                   xhd_parse_kernel      the serial walk of a whole buffer, for what the speculation cannot do
//                   lz_copy_kernel        tokens -> bytes, 64 at a time (sources chased with ds_bpermute / LDS / HBM)
// Per unit, status and length are what the reference's one-shot call returns (MSCOMP_OK / MSCOMP_BUF_ERROR / MSCOMP_DATA_ERROR); the
// oracle restates those semantics (oracle/mscomp_oracle.c) and tests/test_gpu_decompress.py compares both with the compiled reference.
//
// LZNT1 follows the one-shot semantics of the reference, which is its streaming inflate driven once over the whole
// buffer (/root/reference/src/lznt1_decompress.cpp:122-290 through ALL_AT_ONCE_WRAPPER_DECOMPRESS,
// /root/reference/include/mscomp/internal.h:616-630):
//   * chunk headers are walked while output room AND >= 2 input bytes remain (:252-259); header 0 ends the stream and is
//     DATA_ERROR unless it is the last two bytes (:128-134); a chunk longer than the remaining input is kept as partial
//     state, so the one-shot call ends in BUF_ERROR (:136-143, wrapper :627); the signature is checked after that (:151);
//   * every compressed chunk is decoded against a 4096-byte limit (:158,:179), any chunk error becomes DATA_ERROR;
//   * a chunk may decode to fewer than 4096 bytes anywhere in the stream - the next chunk continues right behind it;
//   * output that does not fit (or input left when the output is full) gives BUF_ERROR; a single trailing byte is accepted
//     when it is 0 (:223-227).
// The chunks of a unit are independent once their headers are known, but the headers form a chain (each gives the distance
// to the next). The chain is walked in parallel, speculatively: (1) the compressed input of a unit is cut into segments of
// LZD_SEG bytes; the block of segment s tries every offset of the 4098 bytes that start LZD_HEAD bytes before the segment
// as a header and follows it: wrong guesses die at the first header without the 011 signature (7 of 8 random words), the
// ones that live until the segment begins all arrive at the same header - the landing L_s - which is then followed through
// the segment to the first header of the next one, E_s. (2) One wave per unit checks E_s == L_(s+1) for all s (segment 0
// starts at offset 0, so by induction every chain is the true one), walks the segments again where that fails, and
// counts. (3) One wave per chunk decodes in LDS and writes the chunk where it lands if every earlier chunk holds 4096
// bytes, (4) one wave per unit adds the sizes up, decides the status and notices units with short chunks in the middle,
// (5) whose chunks are decoded again to their exact places.
#include "kernels.h"

namespace msc {

// ===================================================================================================================
// (1) header chain, speculative per segment
// ===================================================================================================================
#define LZD_THREADS 1024u
#define LZD_WINDOW (LZD_HEAD + LZD_SEG + 16u)                      // candidate region + segment + the header that ends it
#define LZD_LDS (LZD_WINDOW + 48u)
#define LZD_NONE  0xFFFFFFFFu                                      // no (unique) landing
#define LZD_ENDED 0xFFFFFFFEu                                      // the chain ended (end of input or a stop) before the segment / inside it
enum { LZD_EOI0 = 0, LZD_EOI1 = 1, LZD_ZERO_OK = 2, LZD_ZERO_BAD = 3, LZD_TRUNC = 4, LZD_BADSIG = 5 };

// One step of the walk of :122-153 at offset pos of a unit of n bytes whose bytes are read through rd(pos):
// returns 0 and advances pos, or 1 + the reason the walk ends here.
template <typename RD>
__device__ __forceinline__ uint32_t lzd_step(uint32_t& pos, uint32_t n, RD rd, uint32_t& last)
{
	if (pos + 2u > n) { if (pos < n) { last = rd(pos); return 1u + LZD_EOI1; } return 1u + LZD_EOI0; }
	const uint32_t hdr = rd(pos) | (rd(pos + 1u) << 8);
	if (hdr == 0) { return 1u + (n - pos == 2u ? LZD_ZERO_OK : LZD_ZERO_BAD); }      // :128-134
	const uint32_t sz = (hdr & 0xFFFu) + 3u;
	if (sz > n - pos) { return 1u + LZD_TRUNC; }                                     // :136-143
	if ((hdr & 0x7000u) != 0x3000u) { return 1u + LZD_BADSIG; }                      // :151
	pos += sz;
	return 0;
}

// per segment g and chain k < LZD_K: segL (landing), segE (first header of the next segment, or LZD_ENDED), segcnt (headers inside the segment),
// segstop (what ended the chain | last byte << 8), segoff (where its header offsets start in cin[g * LZD_SLOTS ...]; LZD_NONE: not recorded)
__global__ __launch_bounds__(LZD_THREADS) void lzd_seg_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, uint32_t* __restrict__ cin,
                                                            uint32_t* __restrict__ segL, uint32_t* __restrict__ segE,
                                                            uint32_t* __restrict__ segcnt, uint32_t* __restrict__ segstop, uint32_t* __restrict__ segoff)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t s_win[];
	__shared__ uint32_t s_min, s_land[LZD_K], s_cnt[LZD_K];
	const uint32_t tid = threadIdx.x, g = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, g), s = g - bt.chunk_prefix[u];
	const uint32_t n = (uint32_t)bt.in_len[u];
	const uint8_t* base = d_in + bt.in_off[u];
	const uint32_t seg0 = s * LZD_SEG, seg1 = seg0 + LZD_SEG;           // the segment owns the headers in [seg0, seg1)
	const uint32_t w0 = s ? seg0 - LZD_HEAD : 0u;
	const uint32_t wend = n < seg1 + 2u ? n : seg1 + 2u;
	const uint32_t a0 = (uint32_t)((uintptr_t)(base + w0) & 15u);
	const uint8_t* ab = base + w0 - a0;
	const uint32_t nw = wend > w0 ? (a0 + (wend - w0) + 15u) >> 4 : 0u;
	for (uint32_t i = tid; i < nw; i += LZD_THREADS) { *reinterpret_cast<uint4*>(s_win + i * 16u) = *reinterpret_cast<const uint4*>(ab + (size_t)i * 16u); }
	if (tid < LZD_K) { s_land[tid] = LZD_NONE; s_cnt[tid] = 0; }
	if (tid == 0) { s_min = LZD_NONE; }
	__syncthreads();
	const uint8_t* wb = s_win + a0 - w0;                                // wb[pos] = byte at unit offset pos
	auto rd = [&](uint32_t p) -> uint32_t { return wb[p]; };
	if (s) {
		// every offset of the candidate region followed to the segment: where does it land (LZD_NONE: it died)
		uint32_t land[5];
		#pragma unroll
		for (int q = 0; q < 5; ++q) {
			const uint32_t c = (uint32_t)q * LZD_THREADS + tid;
			uint32_t pos = w0 + c, last = 0;
			land[q] = LZD_NONE;
			if (c < 4098u && pos <= n) {
				for (;;) {
					if (pos >= seg0) { land[q] = pos; break; }
					const uint32_t r = lzd_step(pos, n, rd, last);
					if (r) { break; }                                    // an end before the segment: nothing of this chain is ours
				}
			}
		}
		// the LZD_K smallest distinct landings
		uint32_t prev = 0; bool first = true;
		for (uint32_t k = 0; k < LZD_K; ++k) {
			#pragma unroll
			for (int q = 0; q < 5; ++q) { if (land[q] != LZD_NONE && (first || land[q] > prev)) { atomicMin(&s_min, land[q]); } }
			__syncthreads();
			const uint32_t m = s_min;
			__syncthreads();
			if (m == LZD_NONE) { break; }
			if (tid == 0) { s_land[k] = m; s_min = LZD_NONE; }
			prev = m; first = false;
			__syncthreads();
		}
		// more than LZD_K different landings: remember nothing (the verify kernel walks such a segment itself)
		bool more = false;
		#pragma unroll
		for (int q = 0; q < 5; ++q) { more |= land[q] != LZD_NONE && !first && land[q] > prev; }
		if (__syncthreads_or(more ? 1 : 0)) { if (tid < LZD_K) { s_land[tid] = LZD_NONE; } }
		__syncthreads();
	} else if (tid == 0) { s_land[0] = 0; }
	__syncthreads();
	// wave k follows chain k through the segment: first to count, then (offsets known) to record
	const uint32_t k = tid >> 6;
	uint32_t L = LZD_NONE, E = LZD_ENDED, count = 0, stopk = 0;
	if (k < LZD_K && (tid & 63u) == 0) {
		L = s_land[k];
		if (L != LZD_NONE) {
			uint32_t pos = L, last = 0;
			for (;;) {
				if (pos >= seg1) { E = pos; break; }
				const uint32_t r = lzd_step(pos, n, rd, last);
				if (r) { stopk = (r - 1u) | (last << 8); break; }
				++count;
			}
			s_cnt[k] = count;
		}
	}
	__syncthreads();
	if (k < LZD_K && (tid & 63u) == 0) {
		uint32_t off = 0;
		for (uint32_t q = 0; q < k; ++q) { off += s_cnt[q]; }
		if (L == LZD_NONE || off + count > LZD_SLOTS) { off = LZD_NONE; }
		else {
			uint32_t* __restrict__ my = cin + (size_t)g * LZD_SLOTS + off;
			uint32_t pos = L, last = 0;
			for (uint32_t i = 0; i < count; ++i) { my[i] = pos; (void)lzd_step(pos, n, rd, last); }
		}
		const size_t r = (size_t)g * LZD_K + k;
		segL[r] = L; segE[r] = E; segcnt[r] = count; segstop[r] = stopk; segoff[r] = off;
	}
}

__device__ unsigned int g_lzd_walked;                                // segments the verify kernel had to walk itself (test hook)
uint32_t lzd_read_walked()
{
	unsigned int v = 0xFFFFFFFFu, z = 0;
	if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_lzd_walked), 4) != hipSuccess) { return 0xFFFFFFFFu; }
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_lzd_walked), &z, 4);
	return v;
}

// (2) per unit, one wave: thread the true chain through the segments (segment 0 starts at offset 0; the chain of segment s + 1 is
// the one that lands where the chosen chain of segment s arrives); a segment without such a chain is walked here, by one lane
// from global memory. Leaves per segment: selcnt (chunks), seloff (where their header offsets are); per unit: stop.
__global__ __launch_bounds__(64) void lzd_verify_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, uint32_t* __restrict__ cin,
                                                       const uint32_t* __restrict__ segL, const uint32_t* __restrict__ segE,
                                                       const uint32_t* __restrict__ segcnt, const uint32_t* __restrict__ segstop, const uint32_t* __restrict__ segoff,
                                                       uint32_t* __restrict__ selcnt, uint32_t* __restrict__ seloff, uint32_t* __restrict__ stop)
{
	const uint32_t lane = threadIdx.x, u = blockIdx.x;
	const uint32_t g0 = bt.chunk_prefix[u], S = bt.chunk_prefix[u + 1] - g0;
	const uint32_t n = (uint32_t)bt.in_len[u];
	const uint8_t* base = d_in + bt.in_off[u];
	auto rd = [&](uint32_t p) -> uint32_t { return base[p]; };
	uint32_t pos = 0, stopk = 0, last = 0;
	bool ended = false;
	for (uint32_t s0 = 0; s0 < S; s0 += 64u) {
		// lane t holds the LZD_K chains of segment s0 + t (an unrecorded chain cannot be used), and the landings of the segment behind it
		const uint32_t sl = s0 + lane;
		uint32_t cl[LZD_K], ce[LZD_K], nl[LZD_K];
		#pragma unroll
		for (uint32_t k = 0; k < LZD_K; ++k) {
			const size_t r = (size_t)(g0 + (sl < S ? sl : S - 1u)) * LZD_K + k;
			const bool usable = sl < S && segoff[r] != LZD_NONE;
			cl[k] = usable ? segL[r] : LZD_NONE; ce[k] = usable ? segE[r] : LZD_NONE;
			nl[k] = (sl + 1u < S && segoff[r + LZD_K] != LZD_NONE) ? segL[r + LZD_K] : LZD_NONE;
		}
		uint32_t mycnt = 0, mysel = LZD_K + 1u;                          // what lane t learns about its segment (LZD_K + 1: not reached, LZD_K: walked here)
		const uint32_t tiles = S - s0 < 64u ? S - s0 : 64u;
		if (ended) { if (sl < S) { selcnt[g0 + sl] = 0; seloff[g0 + sl] = 0; } continue; }
		// ---- all 64 segments at once: chain k of a segment continues as chain map[k] of the next (0xE: the stream ends in it, 0xF: nowhere) ----
		uint32_t map = 0;
		#pragma unroll
		for (uint32_t k = 0; k < LZD_K; ++k) {
			uint32_t to = 0xFu;
			if (ce[k] == LZD_ENDED) { to = 0xEu; }
			else if (ce[k] != LZD_NONE) {
				#pragma unroll
				for (uint32_t q = LZD_K; q-- > 0;) { if (nl[q] == ce[k]) { to = q; } }
			}
			map |= to << (4u * k);
		}
		auto compose = [](uint32_t a, uint32_t b) -> uint32_t {          // chain k -> a[b[k]]
			uint32_t r = 0;
			#pragma unroll
			for (uint32_t k = 0; k < LZD_K; ++k) { const uint32_t bk = (b >> (4u * k)) & 0xFu; r |= (bk < LZD_K ? (a >> (4u * bk)) & 0xFu : bk) << (4u * k); }
			return r;
		};
		uint32_t P = map;
		#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)P, d, 64); if ((int)lane >= d) { P = compose(P, t); } }
		uint32_t carry = 0xFu;                                           // the chain of the tile's first segment that starts at pos
		#pragma unroll
		for (uint32_t k = LZD_K; k-- > 0;) { if ((uint32_t)__builtin_amdgcn_readlane((int)cl[k], 0) == pos) { carry = k; } }
		const uint32_t Pprev = (uint32_t)__shfl_up((int)P, 1, 64);
		const uint32_t sel = carry >= LZD_K ? 0xFu : (lane == 0 ? carry : (Pprev >> (4u * carry)) & 0xFu);
		const u64 fail = __ballot(sl < S && sel == 0xFu);
		if (!fail) {
			uint32_t nxt = 0xFu, e = LZD_NONE;
			#pragma unroll
			for (uint32_t k = 0; k < LZD_K; ++k) { if (sel == k) { nxt = (map >> (4u * k)) & 0xFu; e = ce[k]; } }
			if (sl < S && sel < LZD_K) { mysel = sel; }
			const u64 endm = __ballot(sl < S && sel < LZD_K && nxt == 0xEu);   // the segment in which the stream ends
			if (endm) {
				const uint32_t le = ctz64(endm);
				stopk = segstop[(size_t)(g0 + s0 + le) * LZD_K + (uint32_t)__builtin_amdgcn_readlane((int)sel, (int)le)];
				ended = true;
			} else { pos = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)(tiles - 1u)); }
		} else {
		for (uint32_t t = 0; t < tiles; ++t) {
			if (ended) { break; }
			uint32_t sel1 = LZD_K, e = LZD_NONE;
			#pragma unroll
			for (uint32_t k = 0; k < LZD_K; ++k) {
				const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)cl[k], (int)t), ek = (uint32_t)__builtin_amdgcn_readlane((int)ce[k], (int)t);
				if (sel1 == LZD_K && lk == pos && ek != LZD_NONE) { sel1 = k; e = ek; }
			}
			const uint32_t g = g0 + s0 + t;
			if (sel1 < LZD_K) {
				if (lane == t) { mysel = sel1; }
				if (e == LZD_ENDED) { stopk = segstop[(size_t)g * LZD_K + sel1]; ended = true; } else { pos = e; }
			} else {                                                     // no recorded chain starts here: walk the segment now
				uint32_t count = 0;
				uint32_t* __restrict__ my = cin + (size_t)g * LZD_SLOTS;
				const uint32_t seg1 = (s0 + t + 1u) * LZD_SEG;
				for (;;) {
					if (pos >= seg1) { break; }
					const uint32_t at = pos;
					const uint32_t r = lzd_step(pos, n, rd, last);
					if (r) { stopk = (r - 1u) | (last << 8); ended = true; break; }
					if (lane == 0 && count < LZD_SLOTS) { my[count] = at; }
					++count;
				}
				if (lane == t) { mycnt = count; mysel = LZD_K; }
				if (lane == 0) { atomicAdd(&g_lzd_walked, 1u); }
			}
		}
		}
		if (sl < S) {
			uint32_t myoff = 0;
			if (mysel < LZD_K) { const size_t r = (size_t)(g0 + sl) * LZD_K + mysel; mycnt = segcnt[r]; myoff = segoff[r]; }
			selcnt[g0 + sl] = mycnt; seloff[g0 + sl] = myoff;
		}
	}
	if (lane == 0) { stop[u] = stopk; }
}

// ===================================================================================================================
// (2)/(4) chunk decode: one wave per chunk
// ===================================================================================================================
#define LZD_ERR 0x8000u
struct LzdLds {
	__attribute__((aligned(16))) uint8_t in[4128];     // the chunk (header + data) at its 16-byte phase in global memory
	__attribute__((aligned(16))) uint8_t out[4096 + 64];   // literals at once; a match leaves offset - 1 in its first two bytes until its row is resolved
	u64      bm[64];                                   // token starts over the output positions
	u64      mb[64];                                   // ... those that are matches
	uint16_t gs[464];                                  // start (data offset) of every flag group
};

// size (<= 4096) or LZD_ERR. The decoded bytes are left in L.out.
__device__ __forceinline__ uint32_t lzd_decode_chunk(LzdLds& L, const uint8_t* __restrict__ src, uint32_t in_size, uint32_t lane)
{
	// ---- load: 16-byte words that hold at least one byte of the chunk ----
	const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
	const uint8_t* ab = src - a0;
	const uint32_t nw = (a0 + in_size + 15u) >> 4;                       // <= 258
	for (uint32_t i = lane; i < nw; i += 64u) { *reinterpret_cast<uint4*>(L.in + i * 16u) = *reinterpret_cast<const uint4*>(ab + i * 16u); }
	L.bm[lane] = 0; L.mb[lane] = 0;
	__syncthreads();
	const uint8_t* d = L.in + a0 + 2u;                                  // chunk data
	const uint32_t n = in_size - 2u;                                     // 1..4096
	// ---- flag groups: p -> p + 9 + popcount(flags) ----
	// 64 positions at a time: every lane knows the step that would follow if a group started at its byte; the chain through the
	// window is then followed with readlane (at most 8 steps of 9..17 bytes), and the visited lanes record themselves.
	uint32_t G = 0;
	{
		uint32_t e = 0;                                                  // where the chain enters the window
		for (uint32_t wbase = 0; wbase < n; wbase += 64u) {
			const uint32_t J = 9u + (uint32_t)__builtin_popcount(wbase + lane < n ? (uint32_t)d[wbase + lane] : 0u);
			u64 visited = 0; uint32_t q = e;
			while (q < 64u && wbase + q < n) { visited |= 1ull << q; q += (uint32_t)__builtin_amdgcn_readlane((int)J, (int)q); }
			if ((visited >> lane) & 1ull) { L.gs[G + popc_below(visited)] = (uint16_t)(wbase + lane); }
			G += (uint32_t)__builtin_popcountll(visited);
			e = q >= 64u ? q - 64u : 0u;
		}
	}
	__syncthreads();
	// ---- tokens, 64 at a time: input position, length (the offset/length split depends on the output position), output position ----
	uint32_t base_pos = 0, sh = 12, err = 0;                             // sh: per-lane copy of the (monotone) split
	const uint32_t nb = (G + 7u) >> 3;
	for (uint32_t tb = 0; tb < nb; ++tb) {
		const uint32_t t = tb * 64u + lane, g = t >> 3, k = t & 7u;
		uint32_t ipos = 0, flags = 0;
		if (g < G) { const uint32_t gp = L.gs[g]; flags = d[gp]; ipos = gp + 1u + k + (uint32_t)__builtin_popcount(flags & ((1u << k) - 1u)); }
		const bool valid = g < G && ipos < n;
		const bool is_match = valid && ((flags >> k) & 1u);
		if (is_match && ipos + 2u > n) { err = 1; }                     // :99 (two bytes needed)
		const uint32_t raw = valid ? ((uint32_t)d[ipos] | (is_match ? (uint32_t)d[ipos + 1u] << 8 : 0u)) : 0u;
		uint32_t len, pos;
		for (;;) {
			len = valid ? (is_match ? (raw & ((1u << sh) - 1u)) + 3u : 1u) : 0u;
			pos = base_pos + wave_incl_scan_add_u32(len) - len;
			const u64 m = __ballot(is_match && sh > 4u && pos > (16u << (12u - sh)));       // :100 (the split follows the position)
			if (!m) { break; }
			const uint32_t f = ctz64(m);
			const uint32_t pf = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)f);
			uint32_t ns = (uint32_t)__builtin_amdgcn_readlane((int)sh, (int)f);
			while (ns > 4u && pf > (16u << (12u - ns))) { --ns; }
			if (lane >= f) { sh = ns; }
		}
		const uint32_t off = (raw >> sh) + 1u;
		if (valid && (is_match ? (off > pos || pos + len > 4096u) : pos >= 4096u)) { err = 1; }   // :104-105; a literal beyond the chunk is our DATA_ERROR
		if (__ballot(err)) { return LZD_ERR; }
		if (valid) {
			L.out[pos] = (uint8_t)(is_match ? off - 1u : raw);
			atomicOr(reinterpret_cast<uint32_t*>(L.bm) + (pos >> 5), 1u << (pos & 31u));
			if (is_match) { L.out[pos + 1u] = (uint8_t)((off - 1u) >> 8); atomicOr(reinterpret_cast<uint32_t*>(L.mb) + (pos >> 5), 1u << (pos & 31u)); }
		}
		base_pos = (uint32_t)__builtin_amdgcn_readlane((int)(pos + len), 63);
		sh = (uint32_t)__builtin_amdgcn_readlane((int)sh, 63);
	}
	const uint32_t total = base_pos;
	__syncthreads();
	// ---- bytes, 64 at a time: every byte finds its token and its source; sources inside the row are chased with bpermute ----
	uint32_t carry = 0, carry_off = 0; bool carry_match = false;         // the token running when a row begins
	for (uint32_t rowbase = 0; rowbase < total; rowbase += 64u) {
		const u64 word = L.bm[rowbase >> 6], mword = L.mb[rowbase >> 6];
		const uint32_t i = rowbase + lane;
		const u64 mine = word & ((2ull << lane) - 1ull);
		const uint32_t s = mine ? rowbase + 63u - (uint32_t)__builtin_clzll(mine) : carry;
		const bool mat = mine ? ((mword >> (s - rowbase)) & 1ull) != 0 : carry_match;
		const uint32_t offm1 = mine ? ((uint32_t)L.out[s] | ((uint32_t)L.out[s + 1u] << 8)) : carry_off;     // only meaningful for a match
		uint32_t ptr = i, val = L.out[i];                                 // a literal is in place already
		if (word) {
			carry = rowbase + 63u - (uint32_t)__builtin_clzll(word);
			carry_match = ((mword >> (carry - rowbase)) & 1ull) != 0;
			carry_off = (uint32_t)L.out[carry] | ((uint32_t)L.out[carry + 1u] << 8);
		}
		bool resolved = !mat || i >= total;
		if (!resolved) {
			const uint32_t off = offm1 + 1u, dd = i - s;
			uint32_t rem = dd;
			if (dd >= off) {
				const uint32_t q = (uint32_t)((float)dd * __builtin_amdgcn_rcpf((float)off));
				int32_t rr = (int32_t)dd - (int32_t)(q * off);
				if (rr < 0) { rr += (int32_t)off; } else if (rr >= (int32_t)off) { rr -= (int32_t)off; }
				rem = (uint32_t)rr;
			}
			ptr = s - off + rem;
			if (ptr < rowbase) { val = L.out[ptr]; resolved = true; }
		}
		while (__ballot(!resolved)) {
			const uint32_t tl = resolved ? lane : ptr - rowbase;
			const uint32_t tv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tl << 2), (int)(val | (resolved ? 0x100u : 0u)));
			const uint32_t tp = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tl << 2), (int)ptr);
			if (!resolved) { if (tv & 0x100u) { val = tv & 0xFFu; resolved = true; } else { ptr = tp; if (ptr < rowbase) { val = L.out[ptr]; resolved = true; } } }
		}
		L.out[i] = (uint8_t)val;
		__syncthreads();
	}
	return total;
}

// LDS bytes [0, n) -> global dst (any alignment), one wave
__device__ __forceinline__ void lzd_store(uint8_t* __restrict__ dst, const uint8_t* lds, uint32_t n, uint32_t lane)
{
	uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u);
	if (head > n) { head = n; }
	if (lane < head) { dst[lane] = lds[lane]; }
	const uint32_t body = (n - head) >> 2;
	uint32_t* __restrict__ d32 = reinterpret_cast<uint32_t*>(dst + head);
	for (uint32_t i = lane; i < body; i += 64u) { d32[i] = lds_ld32(lds, head + i * 4u); }
	for (uint32_t i = head + body * 4u + lane; i < n; i += 64u) { dst[i] = lds[i]; }
}

// largest g with prefix[g] <= c (prefix: u64 exclusive scan of the per-segment chunk counts; empty segments share an entry with their successor)
__device__ __forceinline__ uint32_t seg_of_flat(const u64* __restrict__ prefix, uint32_t n_seg, u64 c)
{
	uint32_t lo = 0, hi = n_seg;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (prefix[mid] <= c) { lo = mid; } else { hi = mid; } }
	return lo;
}

// EXACT = false: every chunk, written where it lands if all earlier chunks hold 4096 bytes; records the size.
// EXACT = true: only units flagged irregular; chunks whose place differs are decoded again to the exact place.
// Chunks are numbered through the batch in stream order (flat[g] = number of the first chunk of segment g).
template <bool EXACT>
__global__ __launch_bounds__(64) void lzd_chunk_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, const uint32_t* __restrict__ cin,
                                                      const uint32_t* __restrict__ seloff, const u64* __restrict__ flat, uint16_t* __restrict__ csize,
                                                      const uint32_t* __restrict__ irregular, uint8_t* __restrict__ d_out)
{
	__shared__ LzdLds L;
	const uint32_t lane = threadIdx.x;
	const u64 total_chunks = flat[bt.n_chunks];
	if (EXACT && irregular[bt.n_units] == 0) { return; }
	for (u64 c = blockIdx.x; c < total_chunks; c += gridDim.x) {
		const uint32_t g = seg_of_flat(flat, bt.n_chunks, c);
		const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, g);
		if (EXACT && !irregular[u]) { continue; }
		const u64 ufirst = flat[bt.chunk_prefix[u]];
		const u64 j = c - ufirst;
		const size_t slot = (size_t)g * LZD_SLOTS + seloff[g] + (size_t)(c - flat[g]);
		u64 pos = j * 4096u;
		if (EXACT) {
			u64 acc = 0;                                                 // sum of the sizes of the chunks before this one
			for (u64 i = lane; i < j; i += 64u) { acc += csize[ufirst + i] & 0x1FFFu; }
			for (int o = 32; o; o >>= 1) { acc += __shfl_xor(acc, o, 64); }
			if (acc == pos) { continue; }
			pos = acc;
		}
		const uint8_t* src = d_in + bt.in_off[u] + cin[slot];
		const uint32_t hdr = (uint32_t)src[0] | ((uint32_t)src[1] << 8);
		const uint32_t in_size = (hdr & 0xFFFu) + 3u;
		uint32_t size;
		if (hdr & 0x8000u) {
			size = lzd_decode_chunk(L, src, in_size, lane);
		} else {                                                         // stored chunk (:192-209)
			size = in_size - 2u;
			for (uint32_t i = lane; i < size; i += 64u) { L.out[i] = src[2u + i]; }
			__syncthreads();
		}
		if (!EXACT && lane == 0) { csize[c] = (uint16_t)size; }
		const u64 cap = bt.out_cap[u];
		if (size != LZD_ERR && pos < cap) {
			const u64 room = cap - pos;
			lzd_store(d_out + bt.out_off[u] + pos, L.out, room < size ? (uint32_t)room : size, lane);
		}
		__syncthreads();
	}
}

// ===================================================================================================================
// (3) per unit: positions, status, out_len
// ===================================================================================================================
__global__ __launch_bounds__(256) void lzd_finalize_kernel(BatchTables bt, const u64* __restrict__ flat, const uint32_t* __restrict__ stop,
                                                          const uint16_t* __restrict__ csize, uint32_t* __restrict__ irregular,
                                                          u64* __restrict__ d_out_len, int32_t* __restrict__ d_status)
{
	__shared__ uint32_t s_wsum[4], s_wev[4], s_wirr[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, u = blockIdx.x;
	const u64 ufirst = flat[bt.chunk_prefix[u]], count = flat[bt.chunk_prefix[u + 1]] - ufirst;
	const uint32_t kind = stop[u] & 0xFFu, last = stop[u] >> 8;
	const u64 cap = bt.out_cap[u];
	const uint16_t* __restrict__ sz = csize + ufirst;
	u64 pos = 0; int32_t status = 1; bool irr = false;                   // status 1 = undecided
	for (u64 j0 = 0; j0 < count; j0 += 1024u) {                          // thread t: chunks j0 + 4t .. j0 + 4t + 3
		uint32_t raw[4], size[4], mine = 0;
		#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const u64 j = j0 + tid * 4u + q;
			raw[q] = j < count ? sz[j] : 0u;
			size[q] = raw[q] == LZD_ERR ? 0u : raw[q];
			mine += size[q];
			irr |= j + 1u < count && raw[q] != 4096u;
		}
		const uint32_t incl = wave_incl_scan_add_u32(mine);
		if (lane == 63u) { s_wsum[w] = incl; }
		__syncthreads();
		uint32_t before = incl - mine, total = 0;
		#pragma unroll
		for (uint32_t q = 0; q < 4u; ++q) { if (q < w) { before += s_wsum[q]; } total += s_wsum[q]; }
		// the walk of :252-259 at chunk j: no room left -> the caller sees BUF_ERROR; chunk error -> DATA_ERROR; chunk does not fit -> BUF_ERROR
		u64 p = pos + before; uint32_t ev = 0;
		#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const u64 j = j0 + tid * 4u + q;
			if (!ev && j < count) { ev = p >= cap ? 5u : raw[q] == LZD_ERR ? 3u : p + size[q] > cap ? 5u : 0u; }
			p += size[q];
		}
		const u64 m = __ballot(ev != 0);
		if (lane == 0) { s_wev[w] = m ? (uint32_t)__builtin_amdgcn_readlane((int)ev, (int)ctz64(m)) : 0u; }
		__syncthreads();
		uint32_t first = 0;
		#pragma unroll
		for (uint32_t q = 0; q < 4u; ++q) { if (!first) { first = s_wev[q]; } }
		__syncthreads();
		if (first) { status = -(int32_t)first; break; }
		pos += total;
	}
	const u64 mi = __ballot(irr);
	if (lane == 0) { s_wirr[w] = mi != 0; }
	__syncthreads();
	if (status == 1) {
		const bool room = pos < cap;
		switch (kind) {
		case LZD_EOI0:     status = 0; break;
		case LZD_EOI1:     status = last == 0 ? 0 : -5; break;                       // :223-227
		case LZD_ZERO_OK:  status = room ? 0 : -5; break;                            // the header is only read while there is room (:252)
		case LZD_ZERO_BAD: status = room ? -3 : -5; break;
		case LZD_TRUNC:    status = -5; break;
		default:           status = room ? -3 : -5; break;                           // LZD_BADSIG
		}
	}
	if (tid == 0) {
		const bool any = s_wirr[0] | s_wirr[1] | s_wirr[2] | s_wirr[3];
		d_status[u] = status; d_out_len[u] = status == 0 ? pos : 0;
		irregular[u] = any ? 1u : 0u;
		if (any) { atomicOr(&irregular[bt.n_units], 1u); }
	}
}

__global__ void lzd_clear_kernel(uint32_t* p) { *p = 0; }

void launch_lzd_segments(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const LzdBufs& b)
{
	if (bt.n_units == 0) { return; }
	static PerDeviceOnce attr;
	if (attr.needed()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lzd_seg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LZD_LDS); attr.done(); }
	hipLaunchKernelGGL(lzd_clear_kernel, dim3(1), dim3(1), 0, st, b.irregular + bt.n_units);
	hipLaunchKernelGGL(lzd_seg_kernel, dim3(bt.n_chunks), dim3(LZD_THREADS), LZD_LDS, st, d_in, bt, b.cin, b.segL, b.segE, b.segcnt, b.segstop, b.segoff);
}
void launch_lzd_verify(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const LzdBufs& b)
{
	if (bt.n_units == 0) { return; }
	hipLaunchKernelGGL(lzd_verify_kernel, dim3(bt.n_units), dim3(64), 0, st, d_in, bt, b.cin, b.segL, b.segE, b.segcnt, b.segstop, b.segoff, b.selcnt, b.seloff, b.stop);
}
void launch_lzd_chunks(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const LzdBufs& b, uint8_t* d_out, int exact)
{
	if (bt.n_units == 0) { return; }
	const u64 est = (u64)bt.n_chunks * (LZD_SEG / 2048u);            // the count lives on the device; typical chunks are 2-4 KiB
	const uint32_t grid = est < 16384u ? (uint32_t)est : 16384u;
	if (exact) { hipLaunchKernelGGL(lzd_chunk_kernel<true>, dim3(grid), dim3(64), 0, st, d_in, bt, b.cin, b.seloff, b.flat, b.csize, b.irregular, d_out); }
	else       { hipLaunchKernelGGL(lzd_chunk_kernel<false>, dim3(grid), dim3(64), 0, st, d_in, bt, b.cin, b.seloff, b.flat, b.csize, b.irregular, d_out); }
}
void launch_lzd_finalize(hipStream_t st, const BatchTables& bt, const LzdBufs& b, u64* d_out_len, int32_t* d_status)
{
	if (bt.n_units == 0) { return; }
	hipLaunchKernelGGL(lzd_finalize_kernel, dim3(bt.n_units), dim3(256), 0, st, bt, b.flat, b.stop, b.csize, b.irregular, d_out_len, d_status);
}

// ===================================================================================================================
// Xpress: one wave per stream
// ===================================================================================================================
// xpress_decompress (/root/reference/src/xpress_decompress.cpp:405-462, READ_SYMBOL :62-107): a stream is one chain of tokens - where
// a token starts depends on every token before it (32-bit flag words, 1 / 2 / 3 / 4 / 6 / 10-byte tokens, a length nibble shared
// by two matches) - so a stream is decoded by one wave, streams in parallel. All lanes run the token walk (it is uniform).
// The input is staged through a 2 KiB LDS ring (1 KiB blocks, loaded when the walk gets there), the output through a 10 KiB ring
// from which matches are copied (offsets reach 8192 bytes back) and which goes to HBM in 2 KiB pieces: 12.3 KiB of LDS, 13
// streams per CU. A literal run and a match are moved by the lanes together. The 8 bytes at the next token are fetched (3
// aligned dword reads) before the current token's bytes are moved, so a token costs about one LDS round trip.
#define XPD_INB  1024u
#define XPD_RING 10240u
#define XPD_PIECE 2048u
struct XpdLds { __attribute__((aligned(16))) uint8_t in[2u * XPD_INB]; __attribute__((aligned(16))) uint8_t out[XPD_RING]; };

__global__ __launch_bounds__(64) void xpd_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, uint8_t* __restrict__ d_out,
                                                u64* __restrict__ d_out_len, int32_t* __restrict__ d_status)
{
	__shared__ XpdLds S;
	const uint32_t lane = threadIdx.x, u = blockIdx.x;
	const uint32_t n = (uint32_t)bt.in_len[u];
	const u64 cap = bt.out_cap[u];
	const uint8_t* src = d_in + bt.in_off[u];
	uint8_t* dst = d_out + bt.out_off[u];
	int32_t status = -3; u64 op = 0;
	if (n < 5u) {                                                        // :414-418
		bool ok = n == 0;
		if (n == 4u) { ok = ((uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24)) != 0xFFFFFFFFu; }
		if (lane == 0) { d_status[u] = ok ? 0 : -3; d_out_len[u] = 0; }
		return;
	}
	// ---- input ring: q = offset from the 16-byte aligned base (32 bits: units are below 4 GiB - 4096); block b = q in [1024 b, 1024 (b+1)) ----
	const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
	const uint8_t* ab = src - a0;
	const uint32_t endq = a0 + n;
	uint32_t loaded = 0;                                                 // blocks of 1024 input bytes brought to LDS so far (the last two are resident)
	#define XPD_BLOCK() { __syncthreads(); { const u64 q_ = (u64)loaded * XPD_INB + lane * 16u; \
			*reinterpret_cast<uint4*>(S.in + (loaded & 1u) * XPD_INB + lane * 16u) = q_ < endq ? *reinterpret_cast<const uint4*>(ab + q_) : make_uint4(0, 0, 0, 0); } \
		++loaded; __syncthreads(); }
	XPD_BLOCK() XPD_BLOCK()
	auto rb = [&](uint32_t q) -> uint32_t { return S.in[q & (2u * XPD_INB - 1u)]; };
	const uint32_t* in32 = reinterpret_cast<const uint32_t*>(S.in);
	// bytes q .. q+7 as two dwords
	#define XPD_FETCH8(q, lo, hi) { const uint32_t i_ = ((q) >> 2) & (2u * XPD_INB / 4u - 1u), sh_ = (q) & 3u; \
		const uint32_t w0_ = in32[i_], w1_ = in32[(i_ + 1u) & (2u * XPD_INB / 4u - 1u)], w2_ = in32[(i_ + 2u) & (2u * XPD_INB / 4u - 1u)]; \
		lo = __builtin_amdgcn_alignbyte(w1_, w0_, sh_); hi = __builtin_amdgcn_alignbyte(w2_, w1_, sh_); }
	// ---- output ring: coordinate r = output offset + d0 (d0 = alignment of the destination): pieces of 2048 are 16-byte aligned in HBM ----
	const uint32_t d0 = (uint32_t)((uintptr_t)dst & 15u);
	uint8_t* db = dst - d0;
	u64 flushed = 0;                                                     // ring coordinate up to which the output is in HBM (multiple of XPD_PIECE)
	uint32_t fi = 0;                                                     // flushed mod XPD_RING
	#define XPD_FLUSH() { _Pragma("unroll") for (uint32_t i_ = 0; i_ < 2u; ++i_) { const uint32_t o_ = (i_ * 64u + lane) * 16u; const u64 r_ = flushed + o_; \
			if (r_ >= d0) { *reinterpret_cast<uint4*>(db + r_) = *reinterpret_cast<const uint4*>(S.out + fi + o_); } \
			else { for (uint32_t k_ = d0; k_ < 16u; ++k_) { db[r_ + k_] = S.out[fi + o_ + k_]; } } } \
		flushed += XPD_PIECE; fi += XPD_PIECE; if (fi == XPD_RING) { fi = 0; } }
	auto wrap = [](uint32_t x) -> uint32_t { return x >= XPD_RING ? x - XPD_RING : x; };
	uint32_t wi = d0;                                                    // ring index of output offset op
	u64 nextflush = XPD_PIECE - d0;                                      // output offset at which the next piece is complete
	uint32_t ip = a0;
	uint32_t half = 0; bool have_half = false;
	bool done = false;
	uint32_t lo = 0, hi = 0;
	XPD_FETCH8(ip, lo, hi)
	while (!done) {
		if (ip + 4u > endq) { status = -3; break; }                     // :461 the input ended at a flag word
		while ((u64)loaded * XPD_INB < (u64)ip + 352u && (u64)loaded * XPD_INB < endq) {   // a flag word and its 32 tokens take at most 4 + 32 * 10 bytes
			XPD_BLOCK()
			XPD_FETCH8(ip, lo, hi)
		}
		uint32_t flags = lo;
		uint32_t flagged = flags >> 31;
		flags = (flags << 1) | 1u; ip += 4u;
		lo = hi; XPD_FETCH8(ip, lo, hi)
		do {
			if (ip == endq) {                                            // :433-438
				const uint32_t x = ~flags;
				status = (flagged && !((x + 1u) & x)) ? 0 : -3; done = true; break;
			}
			if (flagged) {
				if (ip + 2u > endq) { status = -3; done = true; break; }
				const uint32_t sym = lo & 0xFFFFu;
				uint32_t used = 2u;                                      // bytes of this token; byte k of the token is (k < 4 ? lo : hi) >> 8 (k & 3)
				const uint32_t off = (sym >> 3) + 1u; uint32_t len = sym & 7u;
				if (len == 7u) {
					if (have_half) { len = half >> 4; have_half = false; }
					else if (ip + used == endq) { status = -3; done = true; break; }
					else { half = (lo >> 16) & 0xFFu; used = 3u; have_half = true; len = half & 0xFu; }
					if (len == 0xFu) {
						if (ip + used == endq) { status = -3; done = true; break; }
						len = used == 2u ? (lo >> 16) & 0xFFu : lo >> 24; ++used;
						if (len == 0xFFu) {
							if (ip + used + 2u > endq) { status = -3; done = true; break; }
							len = used == 3u ? ((lo >> 24) | ((hi & 0xFFu) << 8)) : (hi & 0xFFFFu); used += 2u;
							if (len == 0) {
								if (ip + used + 4u > endq) { status = -3; done = true; break; }
								const uint32_t q4 = ip + used;
								len = rb(q4) | (rb(q4 + 1u) << 8) | (rb(q4 + 2u) << 16) | (rb(q4 + 3u) << 24); used += 4u;
							}
							if (len < 0xFu + 0x7u) { status = -3; done = true; break; }
							len -= 0xFu + 0x7u;
						}
						len += 0xFu;
					}
					len += 0x7u;
				}
				len += 0x3u;
				ip += used;
				XPD_FETCH8(ip, lo, hi)                                  // the next token's bytes travel while this one is copied
				if (off > op) { status = -3; done = true; break; }       // :442
				if (len > cap - op) { status = -5; done = true; break; } // :443
				uint32_t left = len;
				if (off >= 64u) {
					uint32_t si = wi >= off ? wi - off : wi + XPD_RING - off;
					while (left) {
						const uint32_t step = left < 64u ? left : 64u;
						if (lane < step) { S.out[wrap(wi + lane)] = S.out[wrap(si + lane)]; }
						wi = wrap(wi + step); si = wrap(si + step); op += step; left -= step;
						if (op >= nextflush) { __syncthreads(); XPD_FLUSH() nextflush += XPD_PIECE; }
					}
				} else {
					const float ro = __builtin_amdgcn_rcpf((float)off);  // off < 64: (x + 0.5) / off is never within 1e-3 of an integer
					const uint32_t span = (uint32_t)(64.5f * ro) * off;  // whole periods per step
					const uint32_t lm = lane - (uint32_t)(((float)lane + 0.5f) * ro) * off;
					const uint32_t s0 = wi >= off ? wi - off : wi + XPD_RING - off;
					const uint32_t v = S.out[wrap(s0 + lm)];
					while (left) {
						const uint32_t step = left < span ? left : span;
						if (lane < step) { S.out[wrap(wi + lane)] = (uint8_t)v; }
						wi = wrap(wi + step); op += step; left -= step;
						if (op >= nextflush) { __syncthreads(); XPD_FLUSH() nextflush += XPD_PIECE; }
					}
				}
				flagged = flags >> 31; flags <<= 1;
			} else {
				// a run of literals: this token and the zero flags behind it, as far as input and room reach
				uint32_t run = (uint32_t)__builtin_clz(flags) + 1u;
				if (op == cap) { status = -5; done = true; break; }      // :455
				if (run > endq - ip) { run = endq - ip; }
				if ((u64)run > cap - op) { run = (uint32_t)(cap - op); }
				if (lane < run) { S.out[wrap(wi + lane)] = (uint8_t)rb(ip + lane); }
				op += run; ip += run; wi = wrap(wi + run);
				XPD_FETCH8(ip, lo, hi)
				if (op >= nextflush) { __syncthreads(); XPD_FLUSH() nextflush += XPD_PIECE; }
				flagged = (uint32_t)(((u64)flags << (run - 1u)) >> 31) & 1u;
				flags = (uint32_t)((u64)flags << run);
			}
		} while (flags);
	}
	#undef XPD_BLOCK
	#undef XPD_FETCH8
	// the rest of the ring
	__syncthreads();
	if (status == 0) {
		const u64 rend = op + d0;
		for (u64 r = flushed + lane; r < rend; r += 64u) { if (r >= d0) { db[r] = S.out[wrap(fi + (uint32_t)(r - flushed))]; } }
	}
	#undef XPD_FLUSH
	if (lane == 0) { d_status[u] = status; d_out_len[u] = status == 0 ? op : 0; }
}

void launch_xpress_decompress(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint8_t* d_out, u64* d_out_len, int32_t* d_status)
{
	if (bt.n_units == 0) { return; }
	hipLaunchKernelGGL(xpd_kernel, dim3(bt.n_units), dim3(64), 0, st, d_in, bt, d_out, d_out_len, d_status);
}

// ---- the same stream as TOKENS, a flag word at a time ---------------------------------------------------------------------------------
// xpd_kernel above takes one token per step (and moves its bytes): 450 cycles per token, 3.7 s for one 51 MB stream. But where the 32 tokens
// of a flag word start is known at once unless a match carries extra length bytes: a literal takes 1 byte, a match 2, so token r starts
// r + (matches before r) bytes behind the flag word, and only a match whose 3-bit length field is 7 (a match of 10 bytes or more) takes more.
// So lane r reads token r. Such a match takes its length from a nibble, two nibbles to a byte that the first of the two brings along: the
// tokens behind it move by one, and by one more behind a nibble of 15 with its length byte -- found by a short iteration (below). Only the
// longer forms (a match of 280 bytes or more) stop the lanes: the tokens before it are decoded, checked (READ_SYMBOL's and the copy's tests, :62-107 / :442-455, with the
// output offset from a wave scan of the lengths) and written as 32-bit tokens together, that match is decoded by all lanes as one, and
// the rest of the flag word goes the same way from behind it.
// The bytes are produced afterwards by lz_copy_kernel / lz_copy_block_kernel, as for Xpress+Huffman. Status and length are those of the
// serial walk: the first token, in order, that fails decides.
#define XPT_INB 1024u
#define LZT_MAXLEN 32766u                                       // longest match a 32-bit token holds (15 bits); longer ones are cut into pieces
#define XPS_DONE 0x5EC0D0E5u
// The walk of one wave from a flag word on: state in / state out. EMIT: tokens are written at mytok[tc ...] and the copy's tests (:442-443, :455)
// are made against the output offset `op` and the capacity; without it the walk only counts (tokens, bytes) -- for a stretch of a stream whose
// place in the output is not known yet (xps_* below). It ends in front of the first flag word at or behind `limit` (running = true) or where
// the reference's loop ends (status). Offsets are relative to the 16-byte aligned base `ab`, as in xpd_kernel.
struct XptWalk { uint32_t ip; u64 op, tc; uint32_t half, hp; bool have_half; int32_t status; bool running; };
#define XPT_BLOCK() { __syncthreads(); *reinterpret_cast<uint4*>(s_in + (loaded & 1u) * XPT_INB + lane * 16u) = nxt; ++loaded; \
	{ const u64 q_ = (u64)loaded * XPT_INB + lane * 16u; nxt = q_ < endq ? *reinterpret_cast<const uint4*>(ab + q_) : make_uint4(0, 0, 0, 0); } __syncthreads(); }
template <bool EMIT>
__device__ __forceinline__ void xpt_walk(uint8_t* s_in, const uint8_t* __restrict__ ab, const uint32_t endq, uint32_t& loaded, uint4& nxt, XptWalk& W,
                                         const uint32_t limit, const u64 cap, uint32_t* __restrict__ mytok, const uint32_t lane)
{
	auto rb = [&](uint32_t q) -> uint32_t { return s_in[q & (2u * XPT_INB - 1u)]; };
	const uint32_t* in32 = reinterpret_cast<const uint32_t*>(s_in);
	uint32_t ip = W.ip, half = W.half, hp = W.hp;
	u64 op = W.op, tc = W.tc;
	bool have_half = W.have_half, done = false, running = false;
	int32_t status = -3;
	while (!done) {
		if (ip >= limit) { running = true; break; }                     // a flag word at or behind the limit: the walk stops in front of it
		if (ip + 4u > endq) { status = -3; break; }                     // :461 the input ended at a flag word
		while ((u64)loaded * XPT_INB < (u64)ip + 352u && (u64)loaded * XPT_INB < endq) { XPT_BLOCK() }   // a flag word and its 32 tokens take at most 4 + 32 * 10 bytes
		uint32_t m;                                                      // bit i: token i is a match
		{ const uint32_t i_ = (ip >> 2) & (2u * XPT_INB / 4u - 1u); m = __builtin_bitreverse32(__builtin_amdgcn_alignbyte(in32[(i_ + 1u) & (2u * XPT_INB / 4u - 1u)], in32[i_], ip & 3u)); }
		ip += 4u;
		uint32_t j0 = 0;                                                 // first token of the flag word not taken yet
		while (j0 < 32u) {
			const uint32_t mm = m >> j0, cntl = 32u - j0, r = lane & 31u;
			const bool act = lane < cntl;
			const bool is_m = (mm >> r) & 1u;
			// Matches with length field 7 ("long") take their length from a nibble: every other one brings a byte for two of them (the
			// first, if none is pending, :88-95), which moves the tokens behind it by one. Which matches are long depends on where they
			// are, and that on the long ones before them: solved by iteration -- positions from the current set, the set from the symbols at
			// those positions, until it stands. A nibble of 15 is followed by a length byte (:96-99; 255 there: longer forms, not taken here), one
			// more byte to move by: a second set, iterated along. The sets are right up to one token further every round at least (token 0 is
			// always right) and the true sets are a fixed point, the only one; typically 1 + the number of shifting matches rounds.
			const uint32_t below = (1u << r) - 1u, mbefore = (uint32_t)__builtin_popcount(mm & below);
			const uint32_t hsel = have_half ? 1u : 0u;                   // which long matches bring the nibble byte: every other one, the first unless a nibble is pending
			uint32_t longs = 0, ext1 = 0, q, w, hbn, nib, extb;
			bool lng, brings, e1;
			for (;;) {
				const uint32_t kb = (uint32_t)__builtin_popcount(longs & below);
				q = ip + r + mbefore + ((kb + 1u - hsel) >> 1) + (uint32_t)__builtin_popcount(ext1 & below);
				{ const uint32_t i_ = (q >> 2) & (2u * XPT_INB / 4u - 1u); w = __builtin_amdgcn_alignbyte(in32[(i_ + 1u) & (2u * XPT_INB / 4u - 1u)], in32[i_], q & 3u); }   // bytes q .. q+3
				lng = act && is_m && (w & 7u) == 7u;
				const uint32_t now = (uint32_t)__ballot(lng);
				hbn = (w >> 16) & 0xFFu;
				if (now == 0) { brings = false; e1 = false; nib = 0; extb = 0; if (longs == 0 && ext1 == 0) { break; } longs = 0; ext1 = 0; continue; }   // no long match at all: most flag words of text
				const uint32_t kn = (uint32_t)__builtin_popcount(now & below);
				brings = lng && ((kn & 1u) == hsel);
				const uint32_t pv = now & below;
				const uint32_t hprev = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((31u - (uint32_t)__builtin_clz(pv ? pv : 1u)) << 2), (int)hbn);
				nib = brings ? (hbn & 0xFu) : ((kn ? hprev : half) >> 4);
				extb = brings ? w >> 24 : hbn;                          // the length byte behind a nibble of 15
				e1 = lng && nib == 0xFu && extb != 0xFFu;
				const uint32_t now1 = (uint32_t)__ballot(e1);
				if (now == longs && now1 == ext1) { break; }
				longs = now; ext1 = now1;
			}
			const uint32_t b0 = w & 0xFFu, sym = w & 0xFFFFu;
			const bool gone = q >= endq;
			const bool cut = is_m && (q + 2u > endq || (brings && q + 2u == endq) || (lng && nib == 0xFu && q + 2u + (brings ? 1u : 0u) >= endq));   // :64 / :92 / :97
			const u64 ev = __ballot(act && (gone || cut || (lng && nib == 0xFu && !e1)));
			const uint32_t first = ev ? ctz64(ev) : cntl;                // tokens j0 .. j0 + first - 1 are plain: literals and matches of up to 279 bytes
			const bool plain = lane < first;
			const uint32_t len = is_m ? (lng ? (e1 ? extb + 25u : nib + 10u) : (sym & 7u) + 3u) : 1u, off = (sym >> 3) + 1u;
			const uint32_t l = plain ? len : 0u, incl = wave_incl_scan_add_u32(l);
			const u64 opi = op + (incl - l);
			const bool bad_off = EMIT && plain && is_m && (u64)off > opi;                                   // :442
			const bool bad_cap = plain && (is_m ? (u64)len > cap - opi : opi >= cap);               // :443 / :455
			const u64 eb = __ballot(bad_off || bad_cap);
			if (eb) {
				const uint32_t e = ctz64(eb);
				status = ((__ballot(bad_off) >> e) & 1u) ? -3 : -5; done = true; break;
			}
			if (EMIT && plain) { mytok[tc + lane] = is_m ? ((len << 16) | off) : (0x80000000u | b0); }
			tc += first;
			op += first ? (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(first - 1u)) : 0u;
			{	// the nibble state behind the plain tokens
				const uint32_t pl = longs & (first < 32u ? (1u << first) - 1u : 0xFFFFFFFFu);
				if (pl) {
					const bool pend = have_half ^ (((uint32_t)__builtin_popcount(pl) & 1u) != 0);
					if (pend) { const uint32_t ll_ = 31u - (uint32_t)__builtin_clz(pl); half = (uint32_t)__builtin_amdgcn_readlane((int)hbn, (int)ll_); hp = (uint32_t)__builtin_amdgcn_readlane((int)q, (int)ll_) + 2u; }   // (the last long one brought it)
					have_half = pend;
				}
			}
			if (first == cntl) { ip = (uint32_t)__builtin_amdgcn_readlane((int)(q + (is_m ? 2u : 1u) + (brings ? 1u : 0u) + (e1 ? 1u : 0u)), (int)(cntl - 1u)); break; }
			ip = (uint32_t)__builtin_amdgcn_readlane((int)q, (int)first);
			if ((__ballot(gone) >> first) & 1u) {                        // :433-438: the input ends here: this flag and all behind it must be set
				const uint32_t k = j0 + first;
				status = (m >> k) == (0xFFFFFFFFu >> k) ? 0 : -3; done = true; break;
			}
			if ((__ballot(cut) >> first) & 1u) { status = -3; done = true; break; }
			// a match with extra length bytes, by all lanes as one (READ_SYMBOL :62-107)
			const uint32_t sy = (uint32_t)__builtin_amdgcn_readlane((int)sym, (int)first);
			const uint32_t loff = (sy >> 3) + 1u;
			uint32_t used = 2u, ll;
			if (have_half) { ll = half >> 4; have_half = false; }
			else if (ip + used == endq) { status = -3; done = true; break; }
			else { half = rb(ip + 2u); hp = ip + 2u; used = 3u; have_half = true; ll = half & 0xFu; }
			if (ll == 0xFu) {
				if (ip + used == endq) { status = -3; done = true; break; }
				ll = rb(ip + used); ++used;
				if (ll == 0xFFu) {
					if (ip + used + 2u > endq) { status = -3; done = true; break; }
					ll = rb(ip + used) | (rb(ip + used + 1u) << 8); used += 2u;
					if (ll == 0) {
						if (ip + used + 4u > endq) { status = -3; done = true; break; }
						ll = rb(ip + used) | (rb(ip + used + 1u) << 8) | (rb(ip + used + 2u) << 16) | (rb(ip + used + 3u) << 24); used += 4u;
					}
					if (ll < 0xFu + 0x7u) { status = -3; done = true; break; }
					ll -= 0xFu + 0x7u;
				}
				ll += 0xFu;
			}
			ll += 0x7u + 0x3u;
			if (EMIT && (u64)loff > op) { status = -3; done = true; break; }     // :442
			if ((u64)ll > cap - op) { status = -5; done = true; break; } // :443
			const uint32_t np = ll / LZT_MAXLEN + (ll % LZT_MAXLEN ? 1u : 0u);    // pieces with the same offset copy the same bytes
			for (uint32_t k = lane; EMIT && k < np; k += 64u) { mytok[tc + k] = ((k + 1u == np ? ll - (np - 1u) * LZT_MAXLEN : LZT_MAXLEN) << 16) | loff; }
			tc += np; op += ll; ip += used;
			j0 += first + 1u;
		}
	}
	W.ip = ip; W.op = op; W.tc = tc; W.half = half; W.hp = hp; W.have_half = have_half; W.status = status; W.running = running;
}

// ring start for a walk that begins at offset ip0 (from the aligned base): blocks ip0 / 1024 and the next one resident, the third on its way
#define XPT_RING_START(ip0) { loaded = (ip0) / XPT_INB; { const u64 q_ = (u64)loaded * XPT_INB + lane * 16u; nxt = q_ < endq ? *reinterpret_cast<const uint4*>(ab + q_) : make_uint4(0, 0, 0, 0); } XPT_BLOCK() XPT_BLOCK() }

__global__ __launch_bounds__(64) void xpt_parse_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, const u64* __restrict__ tok_prefix, uint32_t* __restrict__ tok,
                                                      u64* __restrict__ ntok, u64* __restrict__ d_out_len, int32_t* __restrict__ d_status, const uint32_t* __restrict__ spec_done)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_in[2u * XPT_INB];
	const uint32_t lane = threadIdx.x, u = blockIdx.x;
	if (spec_done && spec_done[u] == XPS_DONE) { return; }              // the segment-parallel walk has done this stream (xps_* below)
	const uint32_t n = (uint32_t)bt.in_len[u];
	const u64 cap = bt.out_cap[u];
	const uint8_t* src = d_in + bt.in_off[u];
	uint32_t* __restrict__ mytok = tok + tok_prefix[u];
	if (n < 5u) {                                                        // :414-418
		bool ok = n == 0;
		if (n == 4u) { ok = ((uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24)) != 0xFFFFFFFFu; }
		if (lane == 0) { d_status[u] = ok ? 0 : -3; d_out_len[u] = 0; ntok[u] = 0; }
		return;
	}
	// input ring as in xpd_kernel (two blocks of 1024 bytes); the block after the newest one is already on its way from HBM, in registers
	const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
	const uint8_t* ab = src - a0;
	const uint32_t endq = a0 + n;
	uint32_t loaded; uint4 nxt;
	XPT_RING_START(a0)
	XptWalk W = { a0, 0, 0, 0, 0, false, -3, false };
	xpt_walk<true>(s_in, ab, endq, loaded, nxt, W, 0xFFFFFFFFu, cap, mytok, lane);
	if (lane == 0) { d_status[u] = W.status; d_out_len[u] = W.status == 0 ? W.op : 0; ntok[u] = W.status == 0 ? W.tc : 0; }
}

// ---- ONE large Xpress stream by many waves (SURVEY.md 8f-1 / 8f-2a on the decoding side) ----------------------------------------------------
// Where a flag word starts is known only to a walk that comes from the start of the stream. But a walk that starts at a WRONG place falls
// into step with the right one sooner or later (tools/xp_sync_study.py: after 2 KB in the median, 27 KB at the 90th, 137 KB at the 99th
// percentile of 4 800 starts in the corpus: whenever it reaches a flag word of the right walk in the right state, it IS the right walk).
// So the input of a stream is cut into segments of XPS_SEG bytes (16 KiB) and
//   round 0: a wave per segment starts XPS_WARM bytes before its segment as if a flag word began there (no nibble pending), notes the state
//            in which it arrives at the first flag word at or behind the segment start ("landing": offset, pending nibble and where its byte
//            is) and counts tokens and bytes from there to the first flag word at or behind the segment end ("exit"). Segment 0 starts at the start.
//   check:   segment k holds if its landing is the exit of segment k - 1 (and that one ran on); by induction from segment 0 a chain of
//            holding segments is the true parse. The FIRST of a run of segments that do not hold is walked again from the exit of the segment
//            before it -- a segment that holds, so this is the true state -- and goes on through the run until it arrives where a later
//            segment had landed: one round per run, whatever its length (there are stretches of 100 KB and more that no speculative walk
//            enters: flag words 0xAAAAAAAA / 0x55555555 in a lattice of 52-54 bytes, DESIGN_DECODERS.md). XPS_ROUNDS rounds are launched; a segment
//            behind a run may stop holding when the run's exit changes, which is what the further rounds are for.
//   emit:    with the segments' token and byte counts summed up, every segment is walked once more from its true state, writing its tokens
//            at their place and making the tests that need the output offset.
// Whatever does not fit this picture -- a segment that ends in an error, more rounds needed, a test failing, output beyond the capacity --
// sends the stream to the one-wave walk above, which gives the reference's status to the letter.
struct XpsSeg { uint32_t l_ip, l_hp, e_ip, e_hp, kind, redo; u64 ntok, nout, tbase, obase, pad_; };
static_assert(sizeof(XpsSeg) == XPS_SEG_BYTES, "XpsSeg");   // l_hp / e_hp: 0xFFFFFFFF = no nibble pending; kind: 0 ran on, 1 the stream ended well, 2 anything else
__device__ __forceinline__ XpsSeg* xps_segs(const XpsTables& x, uint32_t b) { return reinterpret_cast<XpsSeg*>(static_cast<uint8_t*>(x.seg) + x.seg_prefix[b] * XPS_SEG_BYTES); }
__device__ __forceinline__ void xps_segment_of(const XpsTables& x, uint32_t flat, uint32_t& b, uint32_t& k)
{
	uint32_t lo = 0, hi = x.n_big;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (x.seg_prefix[mid] <= flat) { lo = mid; } else { hi = mid; } }
	b = lo; k = flat - (uint32_t)x.seg_prefix[lo];
}

__global__ void xps_init_kernel(XpsTables x) { const uint32_t b = blockIdx.x * 256u + threadIdx.x; if (b < x.n_big) { x.mode[b] = 1u; } }

// ROUND 0: every segment (speculative start); ROUND > 0: the segments the check marked, from the exit of the segment before
template <int ROUND>
__global__ __launch_bounds__(64) void xps_walk_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, XpsTables x)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_in[2u * XPT_INB];
	const uint32_t lane = threadIdx.x;
	uint32_t b, k;
	xps_segment_of(x, blockIdx.x, b, k);
	const uint32_t u = x.unit[b];
	XpsSeg* __restrict__ seg = xps_segs(x, b);
	const uint32_t nseg = (uint32_t)(x.seg_prefix[b + 1] - x.seg_prefix[b]);
	if (ROUND > 0 && (x.mode[b] != 1u || !seg[k].redo || seg[k - 1u].redo)) { return; }   // (mode 1: rounds still running.) Only the FIRST of a run of
	                                                                     // segments that do not hold walks: it starts from a segment that holds, and goes on below
	const uint32_t n = (uint32_t)bt.in_len[u];
	const uint8_t* src = d_in + bt.in_off[u];
	const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
	const uint8_t* ab = src - a0;
	const uint32_t endq = a0 + n;
	uint32_t loaded; uint4 nxt;
	XptWalk W = { a0, 0, 0, 0, 0, false, -3, false };
	if (ROUND == 0 && k > 0) {
		W.ip = a0 + k * x.seg_bytes - x.warm_bytes;
		XPT_RING_START(W.ip)
		xpt_walk<false>(s_in, ab, endq, loaded, nxt, W, a0 + k * x.seg_bytes, ~(u64)0, nullptr, lane);
		if (!W.running) {                                                  // it fell over before the segment: nothing to offer
			if (lane == 0) { seg[k].l_ip = 0xFFFFFFFFu; seg[k].l_hp = 0; seg[k].e_ip = 0; seg[k].e_hp = 0; seg[k].kind = 2u; seg[k].ntok = 0; seg[k].nout = 0; seg[k].redo = 0; }
			return;
		}
		W.op = 0; W.tc = 0;
	} else if (ROUND > 0) {
		if (seg[k - 1u].kind != 0u) { if (lane == 0) { seg[k].kind = 2u; seg[k].redo = 0; seg[k].l_ip = 0xFFFFFFFFu; } return; }   // nothing to start from
		W.ip = seg[k - 1u].e_ip;
		const uint32_t hp = seg[k - 1u].e_hp;
		W.have_half = hp != 0xFFFFFFFFu; W.hp = W.have_half ? hp : 0u; W.half = W.have_half ? ab[hp] : 0u;
		XPT_RING_START(W.ip)
	} else { XPT_RING_START(a0) }
	for (;;) {
		const uint32_t lim = k + 1u == nseg ? 0xFFFFFFFFu : a0 + (k + 1u) * x.seg_bytes;
		const uint32_t l_ip = W.ip, l_hp = W.have_half ? W.hp : 0xFFFFFFFFu;
		xpt_walk<false>(s_in, ab, endq, loaded, nxt, W, lim, ~(u64)0, nullptr, lane);
		const uint32_t e_hp = W.have_half ? W.hp : 0xFFFFFFFFu;
		if (lane == 0) {
			seg[k].l_ip = l_ip; seg[k].l_hp = l_hp; seg[k].e_ip = W.ip; seg[k].e_hp = e_hp;
			seg[k].kind = W.running ? 0u : (W.status == 0 ? 1u : 2u); seg[k].ntok = W.tc; seg[k].nout = W.op; seg[k].redo = 0;
		}
		// a walk that comes from a segment that holds goes on through the segments behind it until it arrives where one of them had landed:
		// a stretch that no speculative walk entered costs one round, however many segments it spans
		if (ROUND == 0 || !W.running || k + 1u == nseg) { break; }
		if (seg[k + 1u].l_ip == W.ip && seg[k + 1u].l_hp == e_hp) { break; }
		++k; W.op = 0; W.tc = 0;
	}
}

// which segments hold; LAST: sums, verdict (mode 2 = done by segments, 0 = the one-wave walk takes the stream) and the caller's results
template <bool LAST>
__global__ __launch_bounds__(256) void xps_check_kernel(BatchTables bt, XpsTables x, u64* __restrict__ ntok, u64* __restrict__ d_out_len, int32_t* __restrict__ d_status)
{
	__shared__ uint32_t s_bad;
	__shared__ u64 s_t[4], s_o[4];
	const uint32_t tid = threadIdx.x, b = blockIdx.x, u = x.unit[b];
	XpsSeg* __restrict__ seg = xps_segs(x, b);
	const uint32_t nseg = (uint32_t)(x.seg_prefix[b + 1] - x.seg_prefix[b]);
	if (x.mode[b] != 1u) { return; }
	if (tid == 0) { s_bad = 0; }
	__syncthreads();
	uint32_t bad = 0;                                                    // 1: some segment has to be walked again; 2: no use
	for (uint32_t k = tid; k < nseg; k += 256u) {
		bool redo = false;
		if (k > 0) {
			const XpsSeg p = seg[k - 1u];
			if (p.kind != 0u) { bad |= 2u; }                               // the stream ends (well or not) before its last segment: not our case
			else if (seg[k].l_ip != p.e_ip || seg[k].l_hp != p.e_hp) { redo = true; bad |= 1u; }
		}
		if (k + 1u == nseg && seg[k].kind != 1u && !redo) { bad |= seg[k].kind == 2u ? 2u : 2u; }   // the last segment must end the stream, and well
		seg[k].redo = redo ? 1u : 0u;
	}
	if (bad) { atomicOr(&s_bad, bad); }
	__syncthreads();
	const uint32_t verdict = s_bad;
	if (!LAST) {
		if (tid == 0 && (verdict & 2u) && !(verdict & 1u)) { x.mode[b] = 0u; }   // (while segments are still being redone, a bad kind may be that of a wrong walk)
		return;
	}
	if (verdict) { if (tid == 0) { x.mode[b] = 0u; } return; }
	// all segments hold: where their tokens and bytes go
	u64 tcar = 0, ocar = 0;
	for (uint32_t k0 = 0; k0 < nseg; k0 += 256u) {
		const uint32_t k = k0 + tid;
		const u64 t = k < nseg ? seg[k].ntok : 0, o = k < nseg ? seg[k].nout : 0;
		u64 ti = t, oi = o;
		#pragma unroll
		for (uint32_t d = 1; d < 64u; d <<= 1) { const u64 a = __shfl_up(ti, d, 64), c = __shfl_up(oi, d, 64); if ((tid & 63u) >= d) { ti += a; oi += c; } }
		if ((tid & 63u) == 63u) { s_t[tid >> 6] = ti; s_o[tid >> 6] = oi; }
		__syncthreads();
		u64 tb = tcar, ob = ocar, tt = 0, ot = 0;
		for (uint32_t w = 0; w < 4u; ++w) { if (w < (tid >> 6)) { tb += s_t[w]; ob += s_o[w]; } tt += s_t[w]; ot += s_o[w]; }
		if (k < nseg) { seg[k].tbase = tb + ti - t; seg[k].obase = ob + oi - o; }
		tcar += tt; ocar += ot;
		__syncthreads();
	}
	if (tid == 0) {
		const bool fits = ocar <= bt.out_cap[u];                           // beyond the capacity: the one-wave walk says where and how
		x.mode[b] = fits ? 2u : 0u;
		if (fits) { d_status[u] = 0; d_out_len[u] = ocar; ntok[u] = tcar; x.done[u] = XPS_DONE; }
	}
}

__global__ __launch_bounds__(64) void xps_emit_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, XpsTables x, const u64* __restrict__ tok_prefix, uint32_t* __restrict__ tok)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_in[2u * XPT_INB];
	const uint32_t lane = threadIdx.x;
	uint32_t b, k;
	xps_segment_of(x, blockIdx.x, b, k);
	if (x.mode[b] != 2u) { return; }
	const uint32_t u = x.unit[b];
	const XpsSeg* __restrict__ seg = xps_segs(x, b);
	const uint32_t nseg = (uint32_t)(x.seg_prefix[b + 1] - x.seg_prefix[b]);
	const uint32_t n = (uint32_t)bt.in_len[u];
	const uint8_t* src = d_in + bt.in_off[u];
	const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
	const uint8_t* ab = src - a0;
	const uint32_t endq = a0 + n;
	const uint32_t lim = k + 1u == nseg ? 0xFFFFFFFFu : a0 + (k + 1u) * x.seg_bytes;
	uint32_t loaded; uint4 nxt;
	XptWalk W = { seg[k].l_ip, seg[k].obase, seg[k].tbase, 0, 0, false, -3, false };
	if (seg[k].l_hp != 0xFFFFFFFFu) { W.have_half = true; W.hp = seg[k].l_hp; W.half = ab[W.hp]; }
	XPT_RING_START(W.ip)
	xpt_walk<true>(s_in, ab, endq, loaded, nxt, W, lim, bt.out_cap[u], tok + tok_prefix[u], lane);
	// the same walk as the one that was counted, now with the tests that need the output offset: anything else than the counted end is a failed test
	// (... including the pending length nibble the walk leaves with: it seeds the next segment's walk, and a record rewritten by a later round while
	// this segment was read could agree on everything else)
	const uint32_t e_hp = W.have_half ? W.hp : 0xFFFFFFFFu;
	const bool same = W.ip == seg[k].e_ip && W.tc == seg[k].tbase + seg[k].ntok && W.op == seg[k].obase + seg[k].nout && (W.running ? (seg[k].kind == 0u && e_hp == seg[k].e_hp) : (seg[k].kind == 1u && W.status == 0));
	if (!same && lane == 0) { x.done[u] = 0; }                            // the one-wave walk takes the stream after all
}

void launch_xpress_decompress_tokens(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const u64* tok_prefix, uint32_t* tok, u64* ntok,
                                     uint8_t* d_out, u64* d_out_len, int32_t* d_status, int phase, u64 lzg_min_cap, const XpsTables& x);

// ===================================================================================================================
// Xpress+Huffman: one wave per buffer
// ===================================================================================================================
// xpress_huff_decompress (/root/reference/src/xpress_huff_decompress.cpp:130-162, chunk loop :39-129; InputBitstream Bitstream.h:34-106;
// HuffmanDecoder<15,512> HuffmanDecoder.h:28-114). Where a chunk's 256-byte table starts is only known when the chunk before it
// has been decoded (no sizes are stored), and inside a chunk every symbol starts where the previous one ends, so the SYMBOLS of
// a buffer are walked by one wave, buffers in parallel (xhd_parse_kernel). The wave builds the decoding tables of a chunk
// together (counts and canonical ranks by ballots), then all lanes walk the symbols; codes of up to 9 bits resolve with one LDS
// read (symbol << 4 | length). The walk needs the output only as a running length, so it writes 32-bit tokens (a literal, or
// offset | length << 16; a match longer than 32766 is cut into matches with the same offset, which copy the same bytes) and
// keeps 4.1 KiB of LDS: 32 buffers per CU instead of the 2 that a 64 KiB output window allows. The bytes are produced afterwards
// by lz_copy_kernel, in parallel.
#define XHD_INB  1024u                 // input ring: two blocks of this size (the walk looks at most 320 bytes ahead); 4.1 KiB of LDS per wave -> 32 waves per CU
struct XhdLds {
	__attribute__((aligned(16))) uint8_t in[2u * XHD_INB];
	uint16_t fast[512];                 // 9-bit prefix -> symbol << 4 | length (0: longer code)
	uint16_t syms[512];                 // symbols in canonical order
	uint32_t lims[16], poss[16];
};

// tokens of unit u: tok[tok_prefix[u] ...], ntok[u]; d_out_len / d_status as the caller sees them (the bytes follow in lz_copy_kernel)
__global__ __launch_bounds__(64) void xhd_parse_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, const u64* __restrict__ tok_prefix,
                                                      uint32_t* __restrict__ tok, u64* __restrict__ ntok,
                                                      u64* __restrict__ d_out_len, int32_t* __restrict__ d_status, const uint32_t* __restrict__ mode)
{
	__shared__ XhdLds S;
	const uint32_t lane = threadIdx.x, u = blockIdx.x;
	if (mode[u] != XHC_SERIAL) { return; }                               // the chunk-parallel path has done this buffer
	const uint32_t n = (uint32_t)bt.in_len[u];
	const u64 cap = bt.out_cap[u];
	const uint8_t* src = d_in + bt.in_off[u];
	uint32_t* __restrict__ mytok = tok + tok_prefix[u];
	u64 nt = 0; uint32_t ns = 0, treg = 0;                               // tokens in HBM; tokens staged: token k of the batch waits in lane k
	#define XHD_EMIT(w) { treg = lane == ns ? (w) : treg; ++ns; if (ns == 64u) { mytok[nt + lane] = treg; nt += 64u; ns = 0; } }
	int32_t status = 1; u64 op = 0;                                      // 1 = running
	// ---- input ring (see xpd_kernel) ----
	const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
	const uint8_t* ab = src - a0;
	const uint32_t endq = a0 + n;                                        // units are below 4 GiB - 4096
	uint32_t loaded = 0;                                                 // blocks of XHD_INB input bytes brought to LDS so far (the last two are resident)
	// a block is loaded when the walk gets there (one HBM round trip per KiB of input: nothing next to ~2500 symbols); values that
	// live across the walk in registers (a prefetched block) made the compiler wait for memory and shuffle them on every symbol
	#define XHD_BLOCK() { uint8_t* b_ = S.in + (loaded & 1u) * XHD_INB; __syncthreads(); \
		_Pragma("unroll") for (int i_ = 0; i_ < (int)(XHD_INB / 1024u); ++i_) { const u64 q_ = (u64)loaded * XHD_INB + ((uint32_t)i_ * 64u + lane) * 16u; \
			*reinterpret_cast<uint4*>(b_ + ((uint32_t)i_ * 64u + lane) * 16u) = q_ < endq ? *reinterpret_cast<const uint4*>(ab + q_) : make_uint4(0, 0, 0, 0); } \
		++loaded; __syncthreads(); }
	// loaded * XHD_INB never lies behind the walk, so the distance to it is a plain 32-bit difference (units end below 4 GiB - 4096)
	#define XHD_NEED(q, margin) while (loaded * XHD_INB - (q) < (margin) && loaded * XHD_INB < endq) { XHD_BLOCK() }
	XHD_BLOCK() XHD_BLOCK()
	auto rb = [&](uint32_t q) -> uint32_t { return S.in[q & (2u * XHD_INB - 1u)]; };
	uint32_t ip = a0;
	while (status == 1) {
		// ---- a chunk: 256 bytes of code lengths, then its bit stream (:137-152) ----
		if (endq - ip < 260u) { status = (ip != endq) ? -3 : 0; break; } // :140-144
		XHD_NEED(ip, 320u)
		uint32_t cl[8];
		{
			const uint32_t w = rb(ip + 4u * lane) | (rb(ip + 4u * lane + 1u) << 8) | (rb(ip + 4u * lane + 2u) << 16) | (rb(ip + 4u * lane + 3u) << 24);
			#pragma unroll
			for (int k = 0; k < 8; ++k) { cl[k] = (w >> (4 * k)) & 0xFu; }   // symbols 8 lane .. 8 lane + 7
		}
		ip += 256u;
		__syncthreads();
		// SetCodeLengths (HuffmanDecoder.h:42-90): counts, limits, positions, canonical order
		uint32_t last = 0, pos_acc = 0, prevcnt = 0; bool bad = false;
		for (uint32_t i = lane; i < 512u; i += 64u) { S.syms[i] = 0xFFFFu; }
		if (lane == 0) { S.lims[0] = 0; S.poss[0] = 0; }
		for (uint32_t L = 1; L <= 15u; ++L) {
			u64 m[8]; uint32_t cnt = 0, before = 0;
			#pragma unroll
			for (int k = 0; k < 8; ++k) { m[k] = __ballot(cl[k] == L); cnt += (uint32_t)__builtin_popcountll(m[k]); before += popc_below(m[k]); }
			pos_acc += prevcnt; prevcnt = cnt;                           // poss[L] = poss[L-1] + cnts[L-1], cnts[0] = 0
			if (L < 15u) { const uint32_t inc = cnt << (15u - L); if (last + inc > 32768u) { bad = true; } last += inc; }
			else if (last + cnt > 32768u) { bad = true; }
			if (lane == 0) { S.lims[L] = L < 15u ? last : 32768u; S.poss[L] = pos_acc; }
			uint32_t mine = 0;
			#pragma unroll
			for (int k = 0; k < 8; ++k) { if (cl[k] == L) { const uint32_t at = pos_acc + before + mine; if (at < 512u) { S.syms[at] = (uint16_t)(lane * 8u + k); } ++mine; } }
		}
		if (bad) { status = -3; break; }                                 // :149
		__syncthreads();
		const uint32_t lims9 = S.lims[9];
		for (uint32_t i = lane; i < 512u; i += 64u) {
			uint32_t e = 0;
			const uint32_t x = i << 6;
			if (x < lims9) {
				uint32_t L = 1;
				while (x >= S.lims[L]) { ++L; }
				const uint32_t sidx = S.poss[L] + ((x - S.lims[L - 1u]) >> (15u - L));
				const uint32_t sym = sidx < 512u ? S.syms[sidx] : 0xFFFFu;
				e = sym == 0xFFFFu ? 0u : ((sym << 4) | L);
			}
			S.fast[i] = (uint16_t)e;
		}
		__syncthreads();
		// ---- the chunk's symbols (:87-127) ----
		XHD_NEED(ip, 320u)
		uint32_t mask = (rb(ip) << 16) | (rb(ip + 1) << 24) | rb(ip + 2) | (rb(ip + 3) << 8);   // Bitstream.h:44
		uint32_t bits = 32; ip += 4u;
		uint32_t prod = 0;                                               // bytes of this chunk so far, saturating
		bool stream_end = false;
		#define XHD_SKIP(k) { mask <<= (k); bits -= (k); if (bits < 16u && ip + 2u <= endq) { XHD_NEED(ip, 2u) mask |= (rb(ip) | (rb(ip + 1) << 8)) << (16u - bits); bits |= 16u; ip += 2u; } }
		#define XHD_MASK_ZERO() (bits == 0 || (mask >> (32u - bits)) == 0)
		#define XHD_DECODE(sym) { const uint32_t r_ = bits; const uint32_t x_ = r_ < 15u ? (((mask >> 16) >> (16u - r_)) << (15u - r_)) : (mask >> 17); \
			const uint32_t f_ = S.fast[x_ >> 6]; uint32_t n_; \
			if (f_) { n_ = f_ & 0xFu; sym = f_ >> 4; if (n_ > r_) { sym = 0xFFFFu; } else { XHD_SKIP(n_) } } \
			else { n_ = x_ >= lims9 ? 10u : 1u; while (x_ >= S.lims[n_]) { ++n_; } \
				if (n_ > r_) { sym = 0xFFFFu; } else { XHD_SKIP(n_) const uint32_t s_ = S.poss[n_] + ((x_ - S.lims[n_ - 1u]) >> (15u - n_)); sym = s_ >= 512u ? 0xFFFFu : S.syms[s_]; } } }
		while (prod < 65536u || !XHD_MASK_ZERO()) {
			uint32_t sym;
			XHD_DECODE(sym)
			if (sym < 0x100u) {
				if (op == cap) { status = -5; break; }
				XHD_EMIT(0x80000000u | sym)
				++op; ++prod;
			} else {
				if (sym == 0xFFFFu) { status = -3; break; }
				if (sym == 0x100u && ip == endq && XHD_MASK_ZERO()) { stream_end = true; break; }   // :91
				uint32_t len = sym & 0xFu;
				if (len == 0xFu) {
					XHD_NEED(ip, 8u)
					if (endq - ip < 1u) { status = -3; break; }
					len = rb(ip); ip += 1u;
					if (len == 0xFFu) {
						if (endq - ip < 2u) { status = -3; break; }
						len = rb(ip) | (rb(ip + 1) << 8); ip += 2u;
						if (len == 0) {
							if (endq - ip < 4u) { status = -3; break; }
							len = rb(ip) | (rb(ip + 1) << 8) | (rb(ip + 2) << 16) | (rb(ip + 3) << 24); ip += 4u;
						}
						if (len < 0xFu) { status = -3; break; }
						len -= 0xFu;
					}
					len += 0xFu;
				}
				len += 3u;
				const uint32_t ob = (sym >> 4) & 0xFu;
				if (ob > bits) { status = -3; break; }                   // :117
				const uint32_t off = ((mask >> 16) >> (16u - ob)) + (1u << ob);
				XHD_SKIP(ob)
				if (off > op) { status = -3; break; }                    // :120
				if (len > cap - op) { status = -5; break; }              // :121
				op += len; prod = prod + len < prod ? 0xFFFFFFFFu : prod + len;
				while (len > LZT_MAXLEN) { XHD_EMIT(off | (LZT_MAXLEN << 16)) len -= LZT_MAXLEN; }
				XHD_EMIT(off | (len << 16))
			}
		}
		if (status != 1) { break; }
		if (!stream_end) {                                               // :128-134: is the next symbol the end of the stream?
			const uint32_t ip_keep = ip;
			uint32_t sym;
			XHD_DECODE(sym)
			if (sym == 0x100u && ip == endq && XHD_MASK_ZERO()) { stream_end = true; } else { ip = ip_keep; }
		}
		if (stream_end) { status = 0; }
	}
	#undef XHD_BLOCK
	#undef XHD_NEED
	#undef XHD_SKIP
	#undef XHD_MASK_ZERO
	#undef XHD_DECODE
	if (lane < ns) { mytok[nt + lane] = treg; }
	#undef XHD_EMIT
	if (lane == 0) { d_status[u] = status; d_out_len[u] = status == 0 ? op : 0; ntok[u] = status == 0 ? nt + ns : 0; }
}

// ===================================================================================================================
// Xpress+Huffman: the chunks of ONE buffer in parallel (speculative chunk starts)
// ===================================================================================================================
// Where a chunk starts is not stored, but its first 256 bytes are a complete prefix code: the 512 nibbles l satisfy sum 2^(15-l) = 2^15
// (tools/xh_marker_study.py: true for every chunk of the corpus, and for 78 other offsets in 80 MB of streams). xhc_mark_kernel lists the
// offsets of a buffer with that property (plus offset 0), xhc_parse_kernel<1> walks every candidate as ONE chunk on its own, xhc_chain_kernel
// follows end(k) == start(k+1) from offset 0, places the chunks in the output and the token stream and checks that no match reaches in
// front of the buffer; xhc_parse_kernel<2> then writes the tokens of the accepted chunks. Anything unexpected (more candidates than room,
// a broken chain, an error inside a chunk, output beyond the capacity) sends the buffer to the serial walk, which reports the reference's status.
#define XHC_TILE 16384u                 // input bytes per block of xhc_mark_kernel
#define XHC_MAXC 8192u                  // candidates of a buffer the chain check can hold
__global__ __launch_bounds__(256) void xhc_mark_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, const u64* __restrict__ cand_prefix, XhcBufs xb)
{
	__shared__ uint8_t s_b[XHC_TILE + 256u + 16u];
	__shared__ uint32_t s_k[256];
	const uint32_t tile = blockIdx.x, tid = threadIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, tile), t = tile - bt.chunk_prefix[u];
	const uint32_t n = (uint32_t)bt.in_len[u];
	const uint8_t* __restrict__ src = d_in + bt.in_off[u];
	const uint32_t room = (uint32_t)(cand_prefix[u + 1] - cand_prefix[u]);
	uint32_t* __restrict__ my = xb.cand_pos + cand_prefix[u];
	const u64 T0 = (u64)t * XHC_TILE;
	if (t == 0 && tid == 0) { const uint32_t i = atomicAdd(&xb.cand_cnt[u], 1u); if (i < room) { my[i] = 0; } }   // chunk 0 starts at offset 0
	if (T0 >= n) { return; }
	const uint32_t avail = n - T0 < XHC_TILE + 255u ? (uint32_t)(n - T0) : XHC_TILE + 255u;
	for (uint32_t i = tid; i < avail; i += 256u) { s_b[i] = src[T0 + i]; }
	{ const uint32_t lo = tid & 15u, hi = tid >> 4; s_k[tid] = (lo ? 1u << (15u - lo) : 0u) + (hi ? 1u << (15u - hi) : 0u); }
	__syncthreads();
	const uint32_t base = tid * 64u;                                     // this thread: offsets T0 + base .. + 63
	uint32_t sum = 0;
	for (uint32_t i = 0; i < 64u; ++i) {
		const u64 p = T0 + base + i;
		if (p + 260u > n) { break; }                                     // a chunk is a table and at least 4 bytes (:140)
		if (i == 0) { for (uint32_t j = 0; j < 256u; ++j) { sum += s_k[s_b[base + j]]; } }
		else { sum += s_k[s_b[base + i + 255u]] - s_k[s_b[base + i - 1u]]; }
		if (sum == 32768u && p != 0) { const uint32_t k = atomicAdd(&xb.cand_cnt[u], 1u); if (k < room) { my[k] = (uint32_t)p; } }
	}
}

// ONE chunk per wave, wherever a candidate says a chunk starts (speculative: see xhc_mark_kernel / xhc_chain_kernel). PASS 1: every candidate
// is measured (where the next chunk would start, bytes produced, how far its matches reach in front of the chunk, tokens; 2 = not a chunk);
// the candidate at offset 0 is chunk 0 for sure and writes its tokens at once. PASS 2: the chunks the chain check accepted write their
// tokens at their place in the unit's token stream.
#ifdef XHC_PROFILE   // make EXTRA=-DXHC_PROFILE: steps / symbols per step / symbols one at a time, summed and the maximum per chunk (tools/dev/gpu_xhcprof.py)
__device__ unsigned long long g_xhc_prof[8];
extern "C" void mscomp_amd_debug_xhc_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xhc_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xhc_prof), z, 64); }
#define XHC_CN(i, v) { if (lane == 0) { atomicAdd(&g_xhc_prof[i], (unsigned long long)(v)); } }
#define XHC_LOC(i) { ++xhc_loc[i]; }
#define XHC_END() { if (lane == 0) { atomicMax(&g_xhc_prof[4], (unsigned long long)xhc_loc[0]); atomicMax(&g_xhc_prof[5], (unsigned long long)xhc_loc[1]); atomicMax(&g_xhc_prof[6], (unsigned long long)(xhc_loc[0] * 3u + xhc_loc[1])); atomicAdd(&g_xhc_prof[7], 1ull); } }
#else
#define XHC_CN(i, v)
#define XHC_LOC(i)
#define XHC_END()
#endif
#ifndef XHC_WIDE2
#define XHC_WIDE2 1                                            // two windows of 64 bit offsets per step (0: one)
#endif
template <int PASS>
__global__ __launch_bounds__(64) void xhc_parse_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, const u64* __restrict__ tok_prefix,
                                                      const u64* __restrict__ cand_prefix, XhcBufs xb, uint32_t* __restrict__ tok)
{
	__shared__ XhdLds S;
	const uint32_t lane = threadIdx.x, slot = blockIdx.x;
	const uint32_t u = seg_of_flat(cand_prefix, bt.n_units, slot);
	const uint32_t idx = slot - (uint32_t)cand_prefix[u];
	const uint32_t have = xb.cand_cnt[u], room = (uint32_t)(cand_prefix[u + 1] - cand_prefix[u]);
	if (idx >= (have < room ? have : room)) { return; }
	// a buffer of several chunks may have token scratch: its candidates keep their tokens there in PASS 1 (xhc_gather_kernel moves those of the
	// accepted chunks to their place), and PASS 2 is left with the chunks whose tokens did not fit (state bit 4)
	const bool has_scr = xb.scr_prefix != nullptr && xb.scr_prefix[u + 1] > xb.scr_prefix[u];
	if (PASS == 2 && (xb.mode[u] != XHC_SPEC || xb.tok_off[slot] == ~(u64)0 || xb.cand_pos[slot] == 0 || (has_scr && !(xb.res_state[slot] & 4u)))) { return; }
	const bool writing = PASS == 2 || xb.cand_pos[slot] == 0;
	const bool scr = PASS == 1 && !writing && has_scr;
	bool scr_ok = true;
	const uint32_t at = xb.cand_pos[slot];
	const uint32_t n = (uint32_t)bt.in_len[u];
	const uint8_t* src = d_in + bt.in_off[u];
	const u64 tok_at = PASS == 2 ? xb.tok_off[slot] : (u64)0;
	uint32_t* __restrict__ mytok = scr ? xb.scr_tok + (xb.scr_prefix[u] + idx) * (u64)XHC_SCR : tok + tok_prefix[u] + tok_at;
	const u64 tokcap = scr ? (u64)XHC_SCR : tok_prefix[u + 1] - tok_prefix[u] - tok_at;   // chunk 0 writes before the capacity is judged: never beyond the unit's slots
	const bool storing = writing || scr;
#ifdef XHC_PROFILE
	uint32_t xhc_loc[2] = {0, 0};
#endif
	u64 reach = 0;
	u64 nt = 0;                                                          // tokens so far
	#define XHD_EMIT(w) { if (storing && lane == 0 && nt < tokcap) { mytok[nt] = (w); } ++nt; }
	int32_t status = 1; u64 op = 0;                                      // 1 = running
	// ---- input ring (see xpd_kernel) ----
	const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
	const uint8_t* ab = src - a0;
	const uint32_t endq = a0 + n;                                        // units are below 4 GiB - 4096
	uint32_t loaded = 0;                                                 // blocks of XHD_INB input bytes brought to LDS so far (the last two are resident)
	// a block is loaded when the walk gets there (one HBM round trip per KiB of input: nothing next to ~2500 symbols); values that
	// live across the walk in registers (a prefetched block) made the compiler wait for memory and shuffle them on every symbol
	#define XHD_BLOCK() { uint8_t* b_ = S.in + (loaded & 1u) * XHD_INB; __syncthreads(); \
		_Pragma("unroll") for (int i_ = 0; i_ < (int)(XHD_INB / 1024u); ++i_) { const u64 q_ = (u64)loaded * XHD_INB + ((uint32_t)i_ * 64u + lane) * 16u; \
			*reinterpret_cast<uint4*>(b_ + ((uint32_t)i_ * 64u + lane) * 16u) = q_ < endq ? *reinterpret_cast<const uint4*>(ab + q_) : make_uint4(0, 0, 0, 0); } \
		++loaded; __syncthreads(); }
	// loaded * XHD_INB never lies behind the walk, so the distance to it is a plain 32-bit difference (units end below 4 GiB - 4096)
	#define XHD_NEED(q, margin) while (loaded * XHD_INB - (q) < (margin) && loaded * XHD_INB < endq) { XHD_BLOCK() }
	loaded = (a0 + at) / XHD_INB;
	XHD_BLOCK() XHD_BLOCK()
	auto rb = [&](uint32_t q) -> uint32_t { return S.in[q & (2u * XHD_INB - 1u)]; };
	uint32_t ip = a0 + at;
	uint32_t state = 2, next_at = 0;                                      // 0 chunk done, 1 the stream ends with it, 2 not a chunk
	while (status == 1) {
		// ---- a chunk: 256 bytes of code lengths, then its bit stream (:137-152) ----
		if (endq - ip < 260u) { if (ip == endq && at == 0) { state = 1; next_at = 0; } break; }   // an empty buffer is an empty stream (:140-144)
		XHD_NEED(ip, 320u)
		uint32_t cl[8];
		{
			const uint32_t w = rb(ip + 4u * lane) | (rb(ip + 4u * lane + 1u) << 8) | (rb(ip + 4u * lane + 2u) << 16) | (rb(ip + 4u * lane + 3u) << 24);
			#pragma unroll
			for (int k = 0; k < 8; ++k) { cl[k] = (w >> (4 * k)) & 0xFu; }   // symbols 8 lane .. 8 lane + 7
		}
		ip += 256u;
		__syncthreads();
		// SetCodeLengths (HuffmanDecoder.h:42-90): counts, limits, positions, canonical order
		uint32_t last = 0, pos_acc = 0, prevcnt = 0; bool bad = false;
		for (uint32_t i = lane; i < 512u; i += 64u) { S.syms[i] = 0xFFFFu; }
		if (lane == 0) { S.lims[0] = 0; S.poss[0] = 0; }
		for (uint32_t L = 1; L <= 15u; ++L) {
			u64 m[8]; uint32_t cnt = 0, before = 0;
			#pragma unroll
			for (int k = 0; k < 8; ++k) { m[k] = __ballot(cl[k] == L); cnt += (uint32_t)__builtin_popcountll(m[k]); before += popc_below(m[k]); }
			pos_acc += prevcnt; prevcnt = cnt;                           // poss[L] = poss[L-1] + cnts[L-1], cnts[0] = 0
			if (L < 15u) { const uint32_t inc = cnt << (15u - L); if (last + inc > 32768u) { bad = true; } last += inc; }
			else if (last + cnt > 32768u) { bad = true; }
			if (lane == 0) { S.lims[L] = L < 15u ? last : 32768u; S.poss[L] = pos_acc; }
			uint32_t mine = 0;
			#pragma unroll
			for (int k = 0; k < 8; ++k) { if (cl[k] == L) { const uint32_t at = pos_acc + before + mine; if (at < 512u) { S.syms[at] = (uint16_t)(lane * 8u + k); } ++mine; } }
		}
		if (bad) { status = -3; break; }                                 // :149
		__syncthreads();
		const uint32_t lims9 = S.lims[9];
		const uint32_t lim10 = S.lims[10], lim11 = S.lims[11], lim12 = S.lims[12], lim13 = S.lims[13], lim14 = S.lims[14];   // (for the many-symbols step)
		for (uint32_t i = lane; i < 512u; i += 64u) {
			uint32_t e = 0;
			const uint32_t x = i << 6;
			if (x < lims9) {
				uint32_t L = 1;
				while (x >= S.lims[L]) { ++L; }
				const uint32_t sidx = S.poss[L] + ((x - S.lims[L - 1u]) >> (15u - L));
				const uint32_t sym = sidx < 512u ? S.syms[sidx] : 0xFFFFu;
				e = sym == 0xFFFFu ? 0u : ((sym << 4) | L);
			}
			S.fast[i] = (uint16_t)e;
		}
		__syncthreads();
		// ---- the chunk's symbols (:87-127) ----
		XHD_NEED(ip, 320u)
		uint32_t mask = (rb(ip) << 16) | (rb(ip + 1) << 24) | rb(ip + 2) | (rb(ip + 3) << 8);   // Bitstream.h:44
		uint32_t bits = 32; ip += 4u;
		uint32_t prod = 0;                                               // bytes of this chunk so far, saturating
		bool stream_end = false, skip_wide = false;
		#define XHD_SKIP(k) { mask <<= (k); bits -= (k); if (bits < 16u && ip + 2u <= endq) { XHD_NEED(ip, 2u) mask |= (rb(ip) | (rb(ip + 1) << 8)) << (16u - bits); bits |= 16u; ip += 2u; } }
		#define XHD_MASK_ZERO() (bits == 0 || (mask >> (32u - bits)) == 0)
		#define XHD_DECODE(sym) { const uint32_t r_ = bits; const uint32_t x_ = r_ < 15u ? (((mask >> 16) >> (16u - r_)) << (15u - r_)) : (mask >> 17); \
			const uint32_t f_ = S.fast[x_ >> 6]; uint32_t n_; \
			if (f_) { n_ = f_ & 0xFu; sym = f_ >> 4; if (n_ > r_) { sym = 0xFFFFu; } else { XHD_SKIP(n_) } } \
			else { n_ = x_ >= lims9 ? 10u : 1u; while (x_ >= S.lims[n_]) { ++n_; } \
				if (n_ > r_) { sym = 0xFFFFu; } else { XHD_SKIP(n_) const uint32_t s_ = S.poss[n_] + ((x_ - S.lims[n_ - 1u]) >> (15u - n_)); sym = s_ >= 512u ? 0xFFFFu : S.syms[s_]; } } }
		while (prod < 65536u || !XHD_MASK_ZERO()) {
			// a chunk of an encoder ends with its 65536th byte and an empty bit buffer; what still has bits then runs on in the reference
			// (:87) -- possibly to the end of the buffer. A candidate is not followed there: it counts as "not a chunk", and a buffer
			// whose chain does not close without it goes to the serial walk, which follows the reference to the letter.
			if (!writing && prod >= 65536u) { status = -3; break; }
#if XHC_WIDE2
			if (!skip_wide && prod < 65536u && bits >= 16u && endq - ip >= 24u) {
				// ---- many symbols per step: lane b decodes the symbols that would start b and 64 + b bits from here (the code through the same
				// tables, a match's offset bits behind it); the symbols that really follow each other are then a walk b -> b + bits taken from
				// bit 0, by readlane: ~6 (matches) to 16 (8-bit literals) of them. Stops in front of a match with length bytes (they sit in the byte
				// stream, where the next 16 bits would be pulled from: the symbol-at-a-time code below takes that one) and in front of anything
				// invalid; the bit buffer is rebuilt as Bitstream.h would hold it. (One window of 64 bit offsets per step: 8 191 steps for a chunk of
				// literals, 7.9 ms for the slowest chunk of the bench; with two windows half the steps.)
				XHD_NEED(ip, 24u)
				uint32_t dw[5];                                            // bytes ip .. ip + 19: ten 16-bit words, each the next 16 bits of the stream
				{
					const uint32_t* in32 = reinterpret_cast<const uint32_t*>(S.in);
					const uint32_t i_ = (ip >> 2) & (2u * XHD_INB / 4u - 1u), sh_ = ip & 3u, M_ = 2u * XHD_INB / 4u - 1u;
					uint32_t r_[6];
					#pragma unroll
					for (uint32_t k_ = 0; k_ < 6u; ++k_) { r_[k_] = in32[(i_ + k_) & M_]; }
					#pragma unroll
					for (uint32_t k_ = 0; k_ < 5u; ++k_) { dw[k_] = __builtin_amdgcn_alignbyte(r_[k_ + 1u], r_[k_], sh_); }
				}
				#define XHD_SWAP16(x) (((x) << 16) | ((x) >> 16))              /* word k of the stream first: (w0 << 16) | w1 */
				const u64 t0 = ((u64)XHD_SWAP16(dw[0]) << 32) | XHD_SWAP16(dw[1]), t1 = ((u64)XHD_SWAP16(dw[2]) << 32) | XHD_SWAP16(dw[3]), t2 = (u64)XHD_SWAP16(dw[4]) << 32;
				#undef XHD_SWAP16
				const u64 sq0 = ((u64)mask << 32) | (t0 >> bits), sq1 = (t0 << (64u - bits)) | (t1 >> bits), sq2 = (t1 << (64u - bits)) | (t2 >> bits);   // the next 192 bits (bits + 160 real)
				uint32_t vw[2];
				vw[0] = (uint32_t)((lane ? (sq0 << lane) | (sq1 >> (64u - lane)) : sq0) >> 32);
				vw[1] = (uint32_t)((lane ? (sq1 << lane) | (sq2 >> (64u - lane)) : sq1) >> 32);
				uint32_t stp[2], tokw[2], mln[2], mof[2]; bool ism[2]; u64 evm[2];
				#pragma unroll
				for (uint32_t h_ = 0; h_ < 2u; ++h_) {
					const uint32_t view = vw[h_];
					const uint32_t x15 = view >> 17;
					const uint32_t f = S.fast[x15 >> 6];
					uint32_t n, sy;
					if (f) { n = f & 0xFu; sy = f >> 4; }
					else if (x15 < lims9) { n = 1; sy = 0xFFFFu; }           // a short code that no symbol has (the table says 0 for it too)
					else {                                                   // a code of 10 to 15 bits: its length from the limits (wave-uniform, in registers), no loop
						n = 10u + (x15 >= lim10 ? 1u : 0u) + (x15 >= lim11 ? 1u : 0u) + (x15 >= lim12 ? 1u : 0u) + (x15 >= lim13 ? 1u : 0u) + (x15 >= lim14 ? 1u : 0u);
						const uint32_t s_ = S.poss[n] + ((x15 - S.lims[n - 1u]) >> (15u - n)); sy = s_ >= 512u ? 0xFFFFu : S.syms[s_];
					}
					const bool lit = sy < 0x100u, mat = !lit && sy != 0xFFFFu;
					const uint32_t ob = (sy >> 4) & 0xFu;
					const uint32_t moff = ob ? ((view << n) >> (32u - ob)) + (1u << ob) : 1u;
					const uint32_t mlen = lit ? 1u : (sy & 0xFu) + 3u;
					const bool evl = !lit && (!mat || (sy & 0xFu) == 0xFu);
					evm[h_] = __ballot(evl);
					stp[h_] = evl ? 128u : n + (mat ? ob : 0u);              // (the walk ends on a symbol it must not take, which is then dropped again)
					tokw[h_] = lit ? (0x80000000u | sy) : (moff | (mlen << 16)); mln[h_] = mlen; mof[h_] = moff; ism[h_] = mat;
				}
				u64 mk0 = 0, mk1 = 0; uint32_t b = 0;
				while (b < 64u) { mk0 |= (u64)1 << b; b += (uint32_t)__builtin_amdgcn_readlane((int)stp[0], (int)b); }
				if (mk0 & evm[0]) { b = ctz64(mk0 & evm[0]); mk0 &= ~evm[0]; skip_wide = true; }
				else {
					while (b < 128u) { mk1 |= (u64)1 << (b - 64u); b += (uint32_t)__builtin_amdgcn_readlane((int)stp[1], (int)(b - 64u)); }
					if (mk1 & evm[1]) { b = 64u + ctz64(mk1 & evm[1]); mk1 &= ~evm[1]; skip_wide = true; }
				}
				bool on0 = (mk0 >> lane) & 1u, on1 = (mk1 >> lane) & 1u;
				const uint32_t l0 = on0 ? mln[0] : 0u, in0 = wave_incl_scan_add_u32(l0), tot0 = (uint32_t)__builtin_amdgcn_readlane((int)in0, 63);
				const uint32_t l1 = on1 ? mln[1] : 0u, in1 = tot0 + wave_incl_scan_add_u32(l1);
				const uint32_t bf0 = in0 - l0, bf1 = in1 - l1;
				{	// the chunk is full in front of a symbol: the loop condition decides there
					const u64 ov0 = __ballot(on0 && prod + bf0 >= 65536u), ov1 = __ballot(on1 && prod + bf1 >= 65536u);
					if (ov0) { const uint32_t sl = ctz64(ov0); mk0 &= ((u64)1 << sl) - 1u; mk1 = 0; b = sl; skip_wide = false; }
					else if (ov1) { const uint32_t sl = ctz64(ov1); mk1 &= ((u64)1 << sl) - 1u; b = 64u + sl; skip_wide = false; }
					on0 = (mk0 >> lane) & 1u; on1 = (mk1 >> lane) & 1u;
				}
				if (mk0) {
					const uint32_t adv = mk1 ? (uint32_t)__builtin_amdgcn_readlane((int)in1, (int)(63u - (uint32_t)__builtin_clzll(mk1)))
					                         : (uint32_t)__builtin_amdgcn_readlane((int)in0, (int)(63u - (uint32_t)__builtin_clzll(mk0)));
					const u64 op0 = op + bf0, op1 = op + bf1;
					const uint32_t rc0 = (on0 && ism[0] && (u64)mof[0] > op0) ? (uint32_t)((u64)mof[0] - op0) : 0u;   // how far a match reaches in front of the chunk (offsets are below 65536 + 32768)
					const uint32_t rc1 = (on1 && ism[1] && (u64)mof[1] > op1) ? (uint32_t)((u64)mof[1] - op1) : 0u;
					if (__ballot((rc0 | rc1) != 0)) { const uint32_t rmax = wave_max_u32(rc0 > rc1 ? rc0 : rc1); if (rmax > reach) { reach = rmax; } }
					const uint32_t c0 = (uint32_t)__builtin_popcountll(mk0), c1 = (uint32_t)__builtin_popcountll(mk1);
					const u64 ti0 = nt + popc_below(mk0), ti1 = nt + c0 + popc_below(mk1);
					if (storing && on0 && ti0 < tokcap) { mytok[ti0] = tokw[0]; }
					if (storing && on1 && ti1 < tokcap) { mytok[ti1] = tokw[1]; }
					nt += c0 + c1;
					XHC_CN(0, 1) XHC_CN(1, c0 + c1) XHC_LOC(0)
					op += adv; prod += adv;
					// Bitstream.h:61-75: a word is pulled whenever fewer than 16 bits are left
					const int32_t avail = (int32_t)bits - (int32_t)b;
					const uint32_t pulls = avail < 16 ? (uint32_t)(16 - avail + 15) >> 4 : 0u;
					const uint32_t nb = (uint32_t)(avail + 16 * (int32_t)pulls);
					u64 x64;
					if (b < 64u) { x64 = b ? (sq0 << b) | (sq1 >> (64u - b)) : sq0; }
					else if (b < 128u) { const uint32_t c_ = b - 64u; x64 = c_ ? (sq1 << c_) | (sq2 >> (64u - c_)) : sq1; }
					else { x64 = sq2 << (b - 128u); }
					mask = (uint32_t)(x64 >> 32) & (nb >= 32u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> nb));
					bits = nb; ip += 2u * pulls;
					continue;
				}
			}
#else
			if (!skip_wide && prod < 65536u && bits >= 16u && endq - ip >= 16u) {
				// ---- many symbols per step: lane b decodes the symbol that would start b bits from here (the code through the same tables, a match's
				// offset bits behind it); the symbols that really follow each other are then a walk b -> b + bits taken from lane 0, by readlane.
				// Stops in front of a match with length bytes (they sit in the byte stream, where the next 16 bits would be pulled from: the
				// symbol-at-a-time code below takes that one) and in front of anything invalid; the bit buffer is rebuilt as Bitstream.h would hold it.
				XHD_NEED(ip, 16u)
				uint32_t d0, d1, d2;                                       // bytes ip .. ip + 11: six 16-bit words, each the next 16 bits of the stream
				{
					const uint32_t* in32 = reinterpret_cast<const uint32_t*>(S.in);
					const uint32_t i_ = (ip >> 2) & (2u * XHD_INB / 4u - 1u), sh_ = ip & 3u, M_ = 2u * XHD_INB / 4u - 1u;
					const uint32_t a_ = in32[i_], b_ = in32[(i_ + 1u) & M_], c_ = in32[(i_ + 2u) & M_], e_ = in32[(i_ + 3u) & M_];
					d0 = __builtin_amdgcn_alignbyte(b_, a_, sh_); d1 = __builtin_amdgcn_alignbyte(c_, b_, sh_); d2 = __builtin_amdgcn_alignbyte(e_, c_, sh_);
				}
				#define XHD_SWAP16(x) (((x) << 16) | ((x) >> 16))              /* word k of the stream first: (w0 << 16) | w1 */
				const u64 t0 = ((u64)XHD_SWAP16(d0) << 32) | XHD_SWAP16(d1), t1 = (u64)XHD_SWAP16(d2) << 32;
				#undef XHD_SWAP16
				const u64 s_hi = ((u64)mask << 32) | (t0 >> bits), s_lo = (t0 << (64u - bits)) | (t1 >> bits);   // the next 128 bits of the stream (bits + 96 of them real)
				const uint32_t view = (uint32_t)((lane ? (s_hi << lane) | (s_lo >> (64u - lane)) : s_hi) >> 32);
				const uint32_t x15 = view >> 17;
				const uint32_t f = S.fast[x15 >> 6];
				uint32_t n, sy;
				if (f) { n = f & 0xFu; sy = f >> 4; }
				else if (x15 < lims9) { n = 1; sy = 0xFFFFu; }               // a short code that no symbol has (the table says 0 for it too)
				else {                                                       // a code of 10 to 15 bits: its length from the limits (wave-uniform, in registers), no loop
					n = 10u + (x15 >= lim10 ? 1u : 0u) + (x15 >= lim11 ? 1u : 0u) + (x15 >= lim12 ? 1u : 0u) + (x15 >= lim13 ? 1u : 0u) + (x15 >= lim14 ? 1u : 0u);
					const uint32_t s_ = S.poss[n] + ((x15 - S.lims[n - 1u]) >> (15u - n)); sy = s_ >= 512u ? 0xFFFFu : S.syms[s_];
				}
				const bool lit = sy < 0x100u, mat = !lit && sy != 0xFFFFu;
				const uint32_t ob = (sy >> 4) & 0xFu;
				const uint32_t cons = n + (mat ? ob : 0u);
				const uint32_t moff = ob ? ((view << n) >> (32u - ob)) + (1u << ob) : 1u;
				const uint32_t mlen = lit ? 1u : (sy & 0xFu) + 3u;
				const bool evl = !lit && (!mat || (sy & 0xFu) == 0xFu);
				const u64 evm = __ballot(evl);
				const uint32_t step = evl ? 64u : cons;                      // (the walk ends on a symbol it must not take, which is then dropped again)
				u64 mark = 0; uint32_t b = 0;
				while (b < 64u) { mark |= (u64)1 << b; b += (uint32_t)__builtin_amdgcn_readlane((int)step, (int)b); }
				if (mark & evm) { b = ctz64(mark & evm); mark &= ~evm; skip_wide = true; }   // (the next symbol is for the code below: no point in looking at it from 64 lanes again)
				bool on = (mark >> lane) & 1u;
				const uint32_t l = on ? mlen : 0u, incl = wave_incl_scan_add_u32(l), before = incl - l;
				const u64 over = __ballot(on && prod + before >= 65536u);                    // the chunk is full in front of this symbol: the loop condition decides there
				if (over) { const uint32_t sl = ctz64(over); mark &= ((u64)1 << sl) - 1u; b = sl; on = (mark >> lane) & 1u; }
				if (mark) {
					const uint32_t lastl = 63u - (uint32_t)__builtin_clzll(mark);
					const uint32_t adv = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)lastl);
					const u64 opi = op + before;
					const uint32_t rch = (on && mat && (u64)moff > opi) ? (uint32_t)((u64)moff - opi) : 0u;    // how far the match reaches in front of the chunk (offsets are below 65536 + 32768)
					if (__ballot(rch != 0)) { const uint32_t rmax = wave_max_u32(rch); if (rmax > reach) { reach = rmax; } }
					const u64 ti = nt + popc_below(mark);
					if (storing && on && ti < tokcap) { mytok[ti] = lit ? (0x80000000u | sy) : (moff | (mlen << 16)); }
					nt += (uint32_t)__builtin_popcountll(mark);
					XHC_CN(0, 1) XHC_CN(1, (uint32_t)__builtin_popcountll(mark)) XHC_LOC(0)
					op += adv; prod += adv;
					// Bitstream.h:61-75: a word is pulled whenever fewer than 16 bits are left
					const int32_t avail = (int32_t)bits - (int32_t)b;
					const uint32_t pulls = avail < 16 ? (uint32_t)(16 - avail + 15) >> 4 : 0u;
					const uint32_t nb = (uint32_t)(avail + 16 * (int32_t)pulls);
					const u64 x64 = b == 0 ? s_hi : (b < 64u ? (s_hi << b) | (s_lo >> (64u - b)) : s_lo << (b - 64u));
					mask = (uint32_t)(x64 >> 32) & (nb >= 32u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> nb));
					bits = nb; ip += 2u * pulls;
					continue;
				}
			}
#endif
			skip_wide = false;
			XHC_CN(2, 1) XHC_LOC(1)
			uint32_t sym;
			XHD_DECODE(sym)
			if (sym < 0x100u) {
				XHD_EMIT(0x80000000u | sym)
				++op; ++prod;
			} else {
				if (sym == 0xFFFFu) { status = -3; break; }
				if (sym == 0x100u && ip == endq && XHD_MASK_ZERO()) { stream_end = true; break; }   // :91
				uint32_t len = sym & 0xFu;
				if (len == 0xFu) {
					XHD_NEED(ip, 8u)
					if (endq - ip < 1u) { status = -3; break; }
					len = rb(ip); ip += 1u;
					if (len == 0xFFu) {
						if (endq - ip < 2u) { status = -3; break; }
						len = rb(ip) | (rb(ip + 1) << 8); ip += 2u;
						if (len == 0) {
							if (endq - ip < 4u) { status = -3; break; }
							len = rb(ip) | (rb(ip + 1) << 8) | (rb(ip + 2) << 16) | (rb(ip + 3) << 24); ip += 4u;
						}
						if (len < 0xFu) { status = -3; break; }
						len -= 0xFu;
					}
					len += 0xFu;
				}
				len += 3u;
				const uint32_t ob = (sym >> 4) & 0xFu;
				if (ob > bits) { status = -3; break; }                   // :117
				const uint32_t off = ((mask >> 16) >> (16u - ob)) + (1u << ob);
				XHD_SKIP(ob)
				if (off > op && off - op > reach) { reach = off - op; }   // :120 is judged when the chunk's place in the output is known
				op += len; prod = prod + len < prod ? 0xFFFFFFFFu : prod + len;
				if (writing || (scr && len <= 4u * LZT_MAXLEN)) { while (len > LZT_MAXLEN) { XHD_EMIT(off | (LZT_MAXLEN << 16)) len -= LZT_MAXLEN; } }
				else if (len > LZT_MAXLEN) { nt += (len - 1u) / LZT_MAXLEN; len = LZT_MAXLEN; scr_ok = false; }   // only counted: a candidate that is no chunk may "hold" gigabyte matches
				XHD_EMIT(off | (len << 16))
			}
		}
		if (status != 1) { break; }
		if (!stream_end) {                                               // :128-134: is the next symbol the end of the stream?
			const uint32_t ip_keep = ip;
			uint32_t sym;
			XHD_DECODE(sym)
			if (sym == 0x100u && ip == endq && XHD_MASK_ZERO()) { stream_end = true; } else { ip = ip_keep; }
		}
		state = stream_end ? 1u : 0u; next_at = ip - a0;
		break;                                                           // one chunk
	}
	#undef XHD_BLOCK
	#undef XHD_NEED
	#undef XHD_SKIP
	#undef XHD_MASK_ZERO
	#undef XHD_DECODE
	#undef XHD_EMIT
	XHC_END()
	if (PASS == 1 && lane == 0) {
		xb.res_state[slot] = (status == 1 ? state : 2u) | ((scr && (!scr_ok || nt > XHC_SCR)) ? 4u : 0u); xb.res_end[slot] = next_at; xb.res_prod[slot] = op; xb.res_ntok[slot] = nt;
		xb.res_reach[slot] = reach > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)reach;
	}
}

// one wave per buffer: the chain of chunks from offset 0 through the measured candidates
__global__ __launch_bounds__(64) void xhc_chain_kernel(BatchTables bt, const u64* __restrict__ cand_prefix, XhcBufs xb, u64* __restrict__ ntok,
                                                      u64* __restrict__ d_out_len, int32_t* __restrict__ d_status)
{
	__shared__ uint32_t s_pos[XHC_MAXC];
	const uint32_t lane = threadIdx.x, u = blockIdx.x;
	const uint32_t n = (uint32_t)bt.in_len[u];
	const u64 cap = bt.out_cap[u], base = cand_prefix[u];
	const uint32_t room = (uint32_t)(cand_prefix[u + 1] - base), have = xb.cand_cnt[u];
	bool ok = have >= 1u && have <= room && have <= XHC_MAXC;
	const uint32_t cnt = ok ? have : 0u;
	for (uint32_t i = lane; i < cnt; i += 64u) { s_pos[i] = xb.cand_pos[base + i]; xb.tok_off[base + i] = ~(u64)0; }
	__syncthreads();
	uint32_t pos = 0, steps = 0; u64 out = 0, nt = 0; bool success = false;
	while (ok) {
		uint32_t found = 0xFFFFFFFFu;
		for (uint32_t i0 = 0; i0 < cnt; i0 += 64u) {
			const u64 m = __ballot(i0 + lane < cnt && s_pos[i0 + lane] == pos);
			if (m) { found = i0 + ctz64(m); break; }
		}
		if (found == 0xFFFFFFFFu) { break; }
		const u64 sl = base + found;
		const uint32_t st = xb.res_state[sl] & 3u, reach = xb.res_reach[sl], end = xb.res_end[sl];
		if (st == 2u || reach == 0xFFFFFFFFu || (u64)reach > out) { break; }     // not a chunk, or a match reaches in front of the buffer (:120)
		if (lane == 0) { xb.tok_off[sl] = nt; }
		out += xb.res_prod[sl]; nt += xb.res_ntok[sl];
		if (st == 1u) { success = end == n; break; }
		if (end <= pos || ++steps > cnt) { break; }
		pos = end;
	}
	success = success && out <= cap;                                     // beyond the capacity: the serial walk says where and how
	if (lane == 0) {
		xb.mode[u] = success ? XHC_SPEC : XHC_SERIAL;
		if (success) { d_status[u] = 0; d_out_len[u] = out; ntok[u] = nt; }
	}
}

// the tokens of the accepted chunks of a buffer with token scratch: from where PASS 1 left them to their place in the buffer's token stream
__global__ __launch_bounds__(256) void xhc_gather_kernel(BatchTables bt, const u64* __restrict__ tok_prefix, const u64* __restrict__ cand_prefix, XhcBufs xb, uint32_t* __restrict__ tok)
{
	const uint32_t slot = blockIdx.x;
	const uint32_t u = seg_of_flat(cand_prefix, bt.n_units, slot);
	const uint32_t idx = slot - (uint32_t)cand_prefix[u];
	const uint32_t have = xb.cand_cnt[u], room = (uint32_t)(cand_prefix[u + 1] - cand_prefix[u]);
	if (idx >= (have < room ? have : room)) { return; }
	if (xb.scr_prefix == nullptr || xb.scr_prefix[u + 1] <= xb.scr_prefix[u]) { return; }
	if (xb.mode[u] != XHC_SPEC || xb.tok_off[slot] == ~(u64)0 || xb.cand_pos[slot] == 0 || (xb.res_state[slot] & 4u)) { return; }
	const uint32_t* __restrict__ from = xb.scr_tok + (xb.scr_prefix[u] + idx) * (u64)XHC_SCR;
	uint32_t* __restrict__ to = tok + tok_prefix[u] + xb.tok_off[slot];
	const uint32_t cnt = (uint32_t)xb.res_ntok[slot];
	for (uint32_t i = threadIdx.x; i < cnt; i += 256u) { to[i] = from[i]; }
}

// ===================================================================================================================
// tokens -> bytes (one wave per unit)
// ===================================================================================================================
// The output of a unit is produced 2048 bytes at a time. The tokens that start in the window set a bit per start and leave their
// word at that position; then 64 bytes at a time find their token (highest start at or below them; the token running when the
// row begins is carried in registers), hence their source: a literal, or byte (i - s) mod off of the match's first period. A
// source in an earlier window is read back from HBM, one in an earlier row of the window from LDS, one in the same row is chased
// with ds_bpermute pointer jumping (as in the LZNT1 chunk kernel).
#define LZC_W 2048u
struct LzcLds { __attribute__((aligned(16))) uint8_t win[LZC_W + 64]; uint32_t info[LZC_W]; u64 bm[LZC_W / 64u]; };

__global__ __launch_bounds__(64) void lz_copy_kernel(BatchTables bt, const u64* __restrict__ tok_prefix, const uint32_t* __restrict__ tok,
                                                    const u64* __restrict__ ntok, const u64* __restrict__ d_out_len, const int32_t* __restrict__ d_status,
                                                    uint8_t* __restrict__ d_out, uint32_t lzb_min, u64 lzg_min_cap)
{
	__shared__ LzcLds L;
	const uint32_t lane = threadIdx.x, u = blockIdx.x;
	if (d_status[u] != 0 || bt.out_cap[u] >= lzg_min_cap) { return; }   // (units with that much room: lzglobal.hip)
	const u64 total = d_out_len[u], nt = ntok[u];
	if (total >= lzb_min) { return; }                                    // larger units: lz_copy_block_kernel
	const uint32_t* __restrict__ mytok = tok + tok_prefix[u];
	uint8_t* dst = d_out + bt.out_off[u];
	u64 t = 0, tpos = 0;                                                 // next token to place, its output offset
	u64 cur_s = 0; uint32_t cur_w = 0x80000000u;                         // the token running at the current position
	for (u64 w0 = 0; w0 < total; w0 += LZC_W) {
		const uint32_t wlen = total - w0 < LZC_W ? (uint32_t)(total - w0) : LZC_W;
		if (lane < LZC_W / 64u) { L.bm[lane] = 0; }
		__syncthreads();
		// ---- the tokens that start in this window ----
		while (t < nt && tpos < w0 + wlen) {
			const u64 ti = t + lane;
			const uint32_t w = ti < nt ? mytok[ti] : 0x80000000u;
			const uint32_t len = ti < nt ? ((w & 0x80000000u) ? 1u : (w >> 16) & 0x7FFFu) : 0u;
			const uint32_t incl = wave_incl_scan_add_u32(len);
			const u64 p = tpos + incl - len;
			const bool in = ti < nt && p < w0 + wlen;
			if (in) {
				const uint32_t q = (uint32_t)(p - w0);
				L.info[q] = w;
				atomicOr(reinterpret_cast<uint32_t*>(L.bm) + (q >> 5), 1u << (q & 31u));
			}
			const uint32_t k = (uint32_t)__builtin_popcountll(__ballot(in));     // a prefix of the lanes
			t += k;
			tpos += k == 64u ? (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) : (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(k ? k - 1u : 0u)) * (k ? 1u : 0u);
			if (k < 64u) { break; }
		}
		__syncthreads();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");               // earlier windows of this unit are read back from HBM
		// ---- bytes ----
		for (uint32_t rowbase = 0; rowbase < wlen; rowbase += 64u) {
			const u64 word = L.bm[rowbase >> 6];
			const u64 i = w0 + rowbase + lane;
			const u64 mine = word & ((2ull << lane) - 1ull);
			const u64 s = mine ? w0 + rowbase + 63u - (uint32_t)__builtin_clzll(mine) : cur_s;
			const uint32_t inf = mine ? L.info[(uint32_t)(s - w0)] : cur_w;
			if (word) { cur_s = w0 + rowbase + 63u - (uint32_t)__builtin_clzll(word); cur_w = L.info[(uint32_t)(cur_s - w0)]; }
			const bool lit = (inf & 0x80000000u) != 0;
			uint32_t val = inf & 0xFFu;
			const uint32_t rowend = rowbase + 64u < wlen ? 64u : wlen - rowbase;
			bool resolved = lit || lane >= rowend;
			uint32_t ptr = lane;                                         // in-row source lane while unresolved
			if (!resolved) {
				const uint32_t off = inf & 0xFFFFu, dd = (uint32_t)(i - s);
				uint32_t rem = dd;
				if (dd >= off) {
					const uint32_t q = (uint32_t)((float)dd * __builtin_amdgcn_rcpf((float)off));
					int32_t rr = (int32_t)dd - (int32_t)(q * off);
					if (rr < 0) { rr += (int32_t)off; } else if (rr >= (int32_t)off) { rr -= (int32_t)off; }
					rem = (uint32_t)rr;
				}
				const u64 sp = s - off + rem;                            // absolute source position, < s
				if (sp >= w0 + rowbase) { ptr = (uint32_t)(sp - (w0 + rowbase)); }
				else if (sp >= w0) { val = L.win[(uint32_t)(sp - w0)]; resolved = true; }
				else { val = dst[sp]; resolved = true; }
			}
			while (__ballot(!resolved)) {
				const uint32_t tl = resolved ? lane : ptr;
				const uint32_t tv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tl << 2), (int)(val | (resolved ? 0x100u : 0u)));
				const uint32_t tp = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tl << 2), (int)ptr);
				if (!resolved) { if (tv & 0x100u) { val = tv & 0xFFu; resolved = true; } else { ptr = tp; } }
			}
			L.win[rowbase + lane] = (uint8_t)val;
			__syncthreads();
		}
		lzd_store(dst + w0, L.win, wlen, lane);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		__syncthreads();
	}
}


// ---- the same for a LARGE unit: one block of 1024 threads, 8 KiB of output at a time -------------------------------------------------
// lz_copy_kernel walks the output of a unit with one wave (0.4 GB/s: 539 of the 665 ms the 12 files of the bench corpus took as 12 buffers).
// Here a tile of 8192 output bytes is resolved by the whole block: the tokens that start in the tile are placed 1024 at a time (block scan
// of their lengths), every byte finds its token (start bits + a per-word "last start so far" from a block max-scan) and its source one
// match offset back -- in the last 64 KiB of output, kept in an LDS ring (offsets reach at most 65535 back: resolved), or inside the tile
// (a pointer); pointers are then jumped (ptr = ptr[ptr], at most 13 rounds, usually 2-4) until every byte has its value. One word per byte
// holds "value" or "pointer", so a racing read sees one or the other, both of which are right.
#define LZB_MIN_KB 32                                             // (a wave per unit, lz_copy_kernel, takes 3.8 ms for a 64 KiB unit; the block 57 us: 3 239 units 4.0 -> 1.6 ms)
#define LZB_T    8192u
#ifndef LZB_NT
#define LZB_NT   1024u
#endif
#define LZB_RING 73728u                                            // 65536 + LZB_T, a multiple of LZB_T: a tile never wraps
static uint32_t lzb_min_bytes()                                    // units with at least this much output take the block kernel (MSCOMP_AMD_LZB_MIN_KB overrides, for measurements)
{
	static const uint32_t v = [] { const char* e = getenv("MSCOMP_AMD_LZB_MIN_KB"); const long k = e ? atol(e) : 0; return (uint32_t)((k > 0 && k < (1 << 20) ? k : LZB_MIN_KB) << 10); }();
	return v;
}
struct LzbLds {
	__attribute__((aligned(16))) uint8_t ring[LZB_RING];
	uint32_t info[LZB_T];                                          // token word at its start position; later: 0x80000000 | value, or the in-tile source
	uint32_t bm[LZB_T / 32u];
	uint32_t last[LZB_T / 32u];                                    // highest token start in the words before this one (LZB_T = none in this tile)
	uint32_t wsum[16];
	uint32_t carry[4];                                             // tpos (2 words), token running at the tile start: its position relative to the tile (biased), its word
	uint32_t tokbuf[LZB_T + LZB_T / 64u];                          // the next LZB_T tokens of the unit (a tile cannot start more): fetched while the tile before is resolved
};

#ifdef LZB_PROFILE
__device__ unsigned long long g_lzb_prof[8];
extern "C" void mscomp_amd_debug_lzb_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lzb_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lzb_prof), z, 64); }
#define LZB_TM(i) { const unsigned long long t_ = __builtin_readcyclecounter(); if (tid == 0) { atomicAdd(&g_lzb_prof[i], t_ - lzb_prev); } lzb_prev = t_; }
#define LZB_CN(i, v) { if (tid == 0) { atomicAdd(&g_lzb_prof[i], (unsigned long long)(v)); } }
#else
#define LZB_TM(i)
#define LZB_CN(i, v)
#endif
static PerDeviceOnce g_lzb_attr;                                          // the dynamic-LDS attribute of lz_copy_block_kernel, per device (two launch sites)
__global__ __launch_bounds__(LZB_NT) void lz_copy_block_kernel(BatchTables bt, const u64* __restrict__ tok_prefix, const uint32_t* __restrict__ tok,
                                                              const u64* __restrict__ ntok, const u64* __restrict__ d_out_len, const int32_t* __restrict__ d_status,
                                                              uint8_t* __restrict__ d_out, uint32_t lzb_min, u64 lzg_min_cap)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lzb_smem[];
	LzbLds& L = *reinterpret_cast<LzbLds*>(lzb_smem);
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, u = blockIdx.x;
	if (d_status[u] != 0 || bt.out_cap[u] >= lzg_min_cap) { return; }   // (units with that much room: lzglobal.hip)
	const u64 total = d_out_len[u], nt = ntok[u];
	if (total < lzb_min) { return; }
	const uint32_t* __restrict__ mytok = tok + tok_prefix[u];
	uint8_t* __restrict__ dst = d_out + bt.out_off[u];
	u64 t = 0, tpos = 0;                                             // next token to place, its output offset (uniform)
	uint32_t run_w = 0x80000000u;                                    // the token running at the tile start
	uint32_t rbase = 0;                                              // the tile's place in the ring (w0 mod LZB_RING)
	uint32_t pre[LZB_T / LZB_NT];                                    // tokens t + r * 1024 + tid, on their way from HBM
	#pragma unroll
	for (uint32_t r = 0; r < LZB_T / LZB_NT; ++r) { const u64 ti = (u64)r * LZB_NT + tid; pre[r] = ti < nt ? mytok[ti] : 0x80000000u; }
#ifdef LZB_PROFILE
	unsigned long long lzb_prev = __builtin_readcyclecounter();
#endif
	for (u64 w0 = 0; w0 < total; w0 += LZB_T) {
		const uint32_t wlen = total - w0 < LZB_T ? (uint32_t)(total - w0) : LZB_T;
		LZB_CN(6, 1)
		if (tid < LZB_T / 32u) { L.bm[tid] = 0; }
		if (tid == 0) { L.carry[0] = 0; L.carry[3] = 0; }
		#pragma unroll
		for (uint32_t r = 0; r < LZB_T / LZB_NT; ++r) { const uint32_t i_ = r * LZB_NT + tid; L.tokbuf[i_ + (i_ >> 6)] = pre[r]; }   // (one word of padding per 64: the reads below, 8 words apart from lane to lane, meet no bank twice)
		__syncthreads();
		// ---- the tokens that start in this tile: a tile cannot start more than LZB_T, thread j looks at tokens 8 j .. 8 j + 7 of the buffer ----
		if (t < nt && tpos < w0 + wlen) {
			constexpr uint32_t TPT = LZB_T / LZB_NT;                     // tokens per thread
			uint32_t w[TPT], len[TPT], sum = 0;
			#pragma unroll
			for (uint32_t r = 0; r < TPT; ++r) {
				w[r] = L.tokbuf[tid * TPT + r + ((tid * TPT + r) >> 6)];
				len[r] = (t + tid * TPT + r < nt) ? ((w[r] & 0x80000000u) ? 1u : (w[r] >> 16) & 0x7FFFu) : 0u;
				sum += len[r];
			}
			const uint32_t incl = wave_incl_scan_add_u32(sum);
			if (lane == 63u) { L.wsum[wv] = incl; }
			__syncthreads();
			uint32_t base = 0;
			for (uint32_t k = 0; k < wv; ++k) { base += L.wsum[k]; }
			u64 p = tpos + base + incl - sum;
			uint32_t cnt = 0; u64 after = 0;
			uint32_t accw = 0xFFFFFFFFu, accb = 0;                        // start bits of my tokens, collected per 32-position word (8 atomics on a shared word cost 15 000 cycles per tile)
			#pragma unroll
			for (uint32_t r = 0; r < TPT; ++r) {
				if (t + tid * TPT + r < nt && p < w0 + wlen) {
					const uint32_t q = (uint32_t)(p - w0);
					L.info[q] = w[r];
					if ((q >> 5) != accw) { if (accb) { atomicOr(&L.bm[accw], accb); } accw = q >> 5; accb = 0; }
					accb |= 1u << (q & 31u);
					++cnt; after = p + len[r];
				}
				p += len[r];
			}
			if (accb) { atomicOr(&L.bm[accw], accb); }
			// the tokens placed are a prefix of the buffer; the end of the last one is where the next token starts
			{	// one update per wave (a thousand updates of one word are served one after the other)
				const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_add_u32(cnt), 63);
				const uint32_t aw = wave_max_u32(cnt ? (uint32_t)(after - w0) : 0u);
				if (lane == 0 && cw) { atomicAdd(&L.carry[3], cw); atomicMax(&L.carry[0], aw); }
			}
			__syncthreads();
			const uint32_t placed = L.carry[3];
			if (placed) { t += placed; tpos = w0 + L.carry[0]; }
		}
		__syncthreads();
		#pragma unroll
		for (uint32_t r = 0; r < LZB_T / LZB_NT; ++r) { const u64 ti = t + (u64)r * LZB_NT + tid; pre[r] = ti < nt ? mytok[ti] : 0x80000000u; }   // (for the next tile)
		__syncthreads();
		LZB_TM(0)
		// ---- per 32-position word: the highest token start before it (block max-scan over 256 words) ----
		if (tid < LZB_T / 32u) {
			const uint32_t wd = L.bm[tid];
			const uint32_t hi = wd ? tid * 32u + 31u - (uint32_t)__builtin_clz(wd) + 1u : 0u;      // biased by 1: 0 = no start in this word
			const uint32_t incl = wave_incl_scan_max(hi);
			if (lane == 63u) { L.wsum[wv] = incl; }
			L.last[tid] = incl;                                         // (inclusive for now)
		}
		__syncthreads();
		if (tid < LZB_T / 32u) {
			uint32_t before = 0;
			for (uint32_t k = 0; k < wv; ++k) { before = before > L.wsum[k] ? before : L.wsum[k]; }
			const uint32_t incl = L.last[tid] > before ? L.last[tid] : before;
			const uint32_t mine = L.bm[tid] ? tid * 32u + 31u - (uint32_t)__builtin_clz(L.bm[tid]) + 1u : 0u;
			// exclusive: the maximum over the words before this one
			const uint32_t prev_lane = (uint32_t)__shfl_up((int)incl, 1, 64);
			uint32_t excl = lane ? prev_lane : before;
			(void)mine;
			L.wsum[4u + 0u] = 0;                                        // (keeps the slot initialised)
			L.last[tid] = excl;                                         // biased start position, 0 = none before this word in the tile
			if (tid == LZB_T / 32u - 1u) { L.carry[2] = incl; }        // the last start of the tile (biased; 0 = none)
		}
		__syncthreads();
		LZB_TM(1)
		// ---- bytes: value, or where in the tile the value comes from (32-bit, relative to the tile) ----
		uint32_t myw[LZB_T / LZB_NT];
		#pragma unroll
		for (uint32_t r = 0; r < LZB_T / LZB_NT; ++r) {
			const uint32_t q = r * LZB_NT + tid;
			uint32_t word = 0x80000000u;
			if (q < wlen) {
				const uint32_t bits = L.bm[q >> 5] & (0xFFFFFFFFu >> (31u - (q & 31u)));
				const uint32_t sb = bits ? (q & ~31u) + 32u - (uint32_t)__builtin_clz(bits) : L.last[q >> 5];   // biased start of my token (0: it runs since before the tile)
				const uint32_t inf = sb ? L.info[sb - 1u] : run_w;
				if (inf & 0x80000000u) { word = 0x80000000u | (inf & 0xFFu); }
				else {
					const int32_t rel = (int32_t)q - (int32_t)(inf & 0xFFFFu);     // one offset back: the same byte
					if (rel >= 0) { word = (uint32_t)rel; }
					else { const int32_t ri = (int32_t)rbase + rel; word = 0x80000000u | L.ring[ri < 0 ? ri + (int32_t)LZB_RING : ri]; }
				}
			}
			myw[r] = word;
		}
		// the token running when the next tile begins
		const uint32_t lastb = L.carry[2];
		if (lastb) { run_w = L.info[lastb - 1u]; }
		__syncthreads();
		#pragma unroll
		for (uint32_t r = 0; r < LZB_T / LZB_NT; ++r) { L.info[r * LZB_NT + tid] = myw[r]; }
		__syncthreads();
		LZB_TM(2)
		for (;;) {
			LZB_CN(7, 1)
			bool open = false;
			#pragma unroll
			for (uint32_t r = 0; r < LZB_T / LZB_NT; ++r) {
				const uint32_t q = r * LZB_NT + tid;
				uint32_t mine = myw[r];
				if (!(mine & 0x80000000u)) {
					uint32_t tw = __hip_atomic_load(&L.info[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					if (!(tw & 0x80000000u)) { tw = __hip_atomic_load(&L.info[tw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // (two hops per round: half the barriers)
					mine = tw;                                              // its value, or where IT looks
					myw[r] = mine;
					__hip_atomic_store(&L.info[q], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					open |= !(mine & 0x80000000u);
				}
			}
			if (!__syncthreads_or(open ? 1 : 0)) { break; }
		}
		LZB_TM(3)
		// ---- the tile: into the ring and out ----
		#pragma unroll
		for (uint32_t r = 0; r < LZB_T / LZB_NT; ++r) { const uint32_t q = r * LZB_NT + tid; if (q < wlen) { L.ring[rbase + q] = (uint8_t)myw[r]; } }
		__syncthreads();
		{
			uint8_t* __restrict__ o = dst + w0;
			const uint8_t* src = L.ring + rbase;
			uint32_t head = (uint32_t)((4u - ((uintptr_t)o & 3u)) & 3u);
			if (head > wlen) { head = wlen; }
			if (tid < head) { o[tid] = src[tid]; }
			const uint32_t body = (wlen - head) >> 2;
			uint32_t* __restrict__ o32 = reinterpret_cast<uint32_t*>(o + head);
			for (uint32_t k = tid; k < body; k += LZB_NT) { o32[k] = lds_ld32(src, head + k * 4u); }
			for (uint32_t k = head + body * 4u + tid; k < wlen; k += LZB_NT) { o[k] = src[k]; }
		}
		rbase += LZB_T; if (rbase >= LZB_RING) { rbase -= LZB_RING; }
		__syncthreads();
		LZB_TM(4)
	}
}

void launch_xpress_huff_decompress(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const u64* tok_prefix, uint32_t* tok, u64* ntok,
                                   const u64* cand_prefix, uint32_t n_slots, const XhcBufs& xb,
                                   uint8_t* d_out, u64* d_out_len, int32_t* d_status, int phase, u64 lzg_min_cap)
{
	if (bt.n_units == 0) { return; }
	switch (phase) {
	case 0: (void)hipMemsetAsync(xb.cand_cnt, 0, ((size_t)bt.n_units + 1) * sizeof(uint32_t), st);
	        hipLaunchKernelGGL(xhc_mark_kernel, dim3(bt.n_chunks), dim3(256), 0, st, d_in, bt, cand_prefix, xb); break;
	case 1: hipLaunchKernelGGL(xhc_parse_kernel<1>, dim3(n_slots), dim3(64), 0, st, d_in, bt, tok_prefix, cand_prefix, xb, tok); break;
	case 2: hipLaunchKernelGGL(xhc_chain_kernel, dim3(bt.n_units), dim3(64), 0, st, bt, cand_prefix, xb, ntok, d_out_len, d_status); break;
	case 3: if (xb.scr_prefix) { hipLaunchKernelGGL(xhc_gather_kernel, dim3(n_slots), dim3(256), 0, st, bt, tok_prefix, cand_prefix, xb, tok); }
	        hipLaunchKernelGGL(xhc_parse_kernel<2>, dim3(n_slots), dim3(64), 0, st, d_in, bt, tok_prefix, cand_prefix, xb, tok); break;
	case 4: hipLaunchKernelGGL(xhd_parse_kernel, dim3(bt.n_units), dim3(64), 0, st, d_in, bt, tok_prefix, tok, ntok, d_out_len, d_status, xb.mode); break;
	default: {
		if (g_lzb_attr.needed()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz_copy_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LzbLds)); g_lzb_attr.done(); }
		hipLaunchKernelGGL(lz_copy_kernel, dim3(bt.n_units), dim3(64), 0, st, bt, tok_prefix, tok, ntok, d_out_len, d_status, d_out, lzb_min_bytes(), lzg_min_cap);
		hipLaunchKernelGGL(lz_copy_block_kernel, dim3(bt.n_units), dim3(LZB_NT), sizeof(LzbLds), st, bt, tok_prefix, tok, ntok, d_out_len, d_status, d_out, lzb_min_bytes(), lzg_min_cap);
		break;
	}
	}
}

void launch_xpress_decompress_tokens(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const u64* tok_prefix, uint32_t* tok, u64* ntok,
                                     uint8_t* d_out, u64* d_out_len, int32_t* d_status, int phase, u64 lzg_min_cap, const XpsTables& x)
{
	if (bt.n_units == 0) { return; }
	if (phase < 0) {                                                     // large streams by segments (nothing to do without any): -1 speculative walks, -2 one round of check + walk again, -3 verdict + tokens
		if (!x.n_big) { return; }
		if (phase == -1) {
			(void)hipMemsetAsync(x.done, 0, (size_t)bt.n_units * 4u, st);
			hipLaunchKernelGGL(xps_init_kernel, dim3((x.n_big + 255u) / 256u), dim3(256), 0, st, x);
			hipLaunchKernelGGL(xps_walk_kernel<0>, dim3(x.n_seg), dim3(64), 0, st, d_in, bt, x);
		} else if (phase == -2) {
			hipLaunchKernelGGL(xps_check_kernel<false>, dim3(x.n_big), dim3(256), 0, st, bt, x, ntok, d_out_len, d_status);
			hipLaunchKernelGGL(xps_walk_kernel<1>, dim3(x.n_seg), dim3(64), 0, st, d_in, bt, x);
		} else {
			hipLaunchKernelGGL(xps_check_kernel<true>, dim3(x.n_big), dim3(256), 0, st, bt, x, ntok, d_out_len, d_status);
			hipLaunchKernelGGL(xps_emit_kernel, dim3(x.n_seg), dim3(64), 0, st, d_in, bt, x, tok_prefix, tok);
		}
		return;
	}
	if (phase == 0) { hipLaunchKernelGGL(xpt_parse_kernel, dim3(bt.n_units), dim3(64), 0, st, d_in, bt, tok_prefix, tok, ntok, d_out_len, d_status, (const uint32_t*)(x.n_big ? x.done : nullptr)); return; }
	if (g_lzb_attr.needed()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz_copy_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LzbLds)); g_lzb_attr.done(); }
	if (phase == 1) { hipLaunchKernelGGL(lz_copy_kernel, dim3(bt.n_units), dim3(64), 0, st, bt, tok_prefix, tok, ntok, d_out_len, d_status, d_out, lzb_min_bytes(), lzg_min_cap); }
	else { hipLaunchKernelGGL(lz_copy_block_kernel, dim3(bt.n_units), dim3(LZB_NT), sizeof(LzbLds), st, bt, tok_prefix, tok, ntok, d_out_len, d_status, d_out, lzb_min_bytes(), lzg_min_cap); }
}

} // namespace msc

// xlazy.hip -- the Xpress-family match finder run LAZILY: Find only at positions a greedy parse can start a token at.
//
// Replaces XpressDictionary<MaxOffset,...>::Find / GetMatchLength (/root/reference/include/mscomp/XpressDictionary.h:145-183, :72-94)
// as called by xpress_compress (/root/reference/src/xpress_compress.cpp:269-271) and xh_compress_lz77
// (/root/reference/src/xpress_huff_compress.cpp:60,90), like xp_find_kernel -- which evaluates EVERY position. The reference runs Find
// at token starts only; a token covers 3-17 bytes of the bench corpus, so the all-positions finder does 3-17 x the reference's work
// (xp_find_kernel: 3.0 of the 5.5 ms of an Xpress pass, 4.4 of the 8.4 ms of an Xpress+Huffman pass).
//
// The walk from token to token is serial, but it forgets: started at ANY position as if a token began there, the greedy parse meets
// the true parse within a few tokens and is identical from there on (Find(p) depends on p alone). The parse kernels behind this one
// (xpress_emit*, xh_parse_kernel) walk the per-position (len-3, offset) arrays themselves and only ever look at positions on the
// path they walk, so it is enough to fill those arrays for a SUPERSET of the true token starts, everything else being "no match":
//   0. the block (1024 threads = one 64 KiB chunk) stages the chunk and its match window in LDS (Xpress+Huffman: 64 KiB + 64 KiB;
//      Xpress, units up to 64 KiB: the unit) and clears the chunk's offset array;
//   1. SPECULATE: lane s parses the 64-byte segment s like a CPU thread would -- Find at its position (<= 11 chain links from
//      xp_links_kernel, 16-byte LDS compares), extend, next token -- as if a token started at the segment's first byte, and marks its
//      token starts in an LDS bit mask;
//   2. CONTINUE: every lane walks on from where its parse left its segment until it stands on a speculative token start of a later
//      segment (typically 1-3 tokens). By induction over the true parse -- its first token is segment 0's; a true token that is a
//      speculative token of segment t is followed by segment t's tokens up to t's exit, where lane t's continuation takes over until it
//      meets a later segment's speculation -- every true token start has been visited by some lane.
//   3. (Xpress only) the ONE-lazy-Fill-per-token rule (xpress_compress.cpp:269) makes up to 7 positions after a match longer than
//      8 KiB forced literals and resumes the parse behind them: the (at most 8) possible resume points behind every such match get
//      a continuation walk of their own.
// What a lane writes for a position is what the all-positions finder would write (length capped at 48, offset), so both finders
// give the parse kernels the same answers on every path they can take.
#include "common.h"
#include "kernels.h"

namespace msc {

#define XC_NT   1024u
#define XC_SEG  64u
#define XC_PAD  80u                       // bytes staged behind the chunk (16-byte reads of a 48-byte compare at the chunk's last positions)
#define XC_TAB  64u                       // Xpress: ends of matches longer than 8 KiB (distinct values)

__device__ __forceinline__ uint32_t xc_hash3(uint32_t w)      // XpressDictionary.h:57-60 closed form
{
	return (((w & 0x1Fu) << 10) ^ (((w >> 8) & 0xFFu) << 5) ^ ((w >> 16) & 0xFFu)) & 0x7FFFu;
}
__device__ __forceinline__ uint32_t xc_diff16(const uint4 a, const uint4 b) { return first_nz_byte16(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }

struct XcChunk {
	u64 n, cbase;                // unit length, unit position of the chunk
	uint32_t cn, k, crel;        // bytes in the chunk, chunk number inside the unit, window-relative position of the chunk start
	const uint16_t* lkc;         // links of this chunk (the previous chunk's array directly precedes it)
	const uint16_t* lh_prev;     // final head table of the previous chunk
	uint32_t prev_last_hash;     // hash of the previous chunk's last position (65535 as a head is "none" otherwise)
	uint16_t* mlen3c;
	uint16_t* moffc;
};

// The token that starts at chunk position p: stores what the all-positions finder stores for p (len-3 capped at 45, offset;
// offset 0 = no match) and returns the token's real length (1 = literal), which the lane needs to go on.
//   XH:  Find only with >= 3 bytes left in the chunk (xpress_huff_compress.cpp:90), window 0xFFFF reaching into the previous
//        chunk, token clipped to the chunk (:93).
//   !XH: Find for p < n - 2 (xpress_compress.cpp:266), window 0x2000, no clipping.
template <bool XH>
__device__ __forceinline__ uint32_t xc_token(const XcChunk& c, const uint8_t* __restrict__ s_data, uint32_t p)
{
	constexpr uint32_t MAXOFF = XH ? 0xFFFFu : 0x2000u;
	const u64 P = c.cbase + p;
	const uint32_t rem = c.cn - p;                                 // bytes left in the chunk (Xpress: in the unit)
	const bool can = XH ? (rem >= 3u) : (P + 2u < c.n);
	if (can) {
		const uint32_t pw = c.crel + p;                                // window-relative position
		const uint4 own = lds_ld128(s_data, pw);
		const uint32_t h = xc_hash3(own.x);
		const u64 lim64 = c.n - P - 1u;                            // never count the buffer's final byte (XpressDictionary.h:88-93)
		const uint32_t lim = lim64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)lim64;
		const uint32_t cap = lim < 48u ? lim : 48u;
		const bool prev_last = (c.prev_last_hash == h);
		uint32_t best = 2, boff = 0, chain = 11;
		// first candidate: my own link (an offset inside the chunk), else the previous chunk's last position with my hash
		uint32_t x = c.lkc[p];
		int32_t base = 0;
		bool alive = true;
		if (x == 0xFFFFu) {
			if (c.k == 0) { alive = false; }
			else { x = c.lh_prev[h]; base = -65536; alive = (x != 0xFFFFu) || prev_last; }
		}
		int32_t xr = base + (int32_t)x;
		alive = alive && (uint32_t)((int32_t)p - xr) <= MAXOFF;
		while (alive) {
			const uint32_t dist = (uint32_t)((int32_t)p - xr);
			uint32_t xn = c.lkc[xr];                                   // the link of xr (L2), consumed after the compare
			const uint32_t xw = pw - dist;
			uint32_t l = xc_diff16(lds_ld128(s_data, xw), own);
			if (l == 16u && cap > 16u) {                               // on, 16 bytes at a time, up to the cap
				uint32_t dl = 16;
				for (;;) {
					const uint32_t la = xc_diff16(lds_ld128(s_data, xw + dl), lds_ld128(s_data, pw + dl));
					l = dl + la;
					if (la < 16u || l >= cap) { break; }
					dl += 16u;
				}
			}
			l = l < cap ? l : cap;
			if (l > best) { best = l; boff = dist; }                   // strictly longer only: the nearer one wins ties
			bool more = (best < 48u) && (--chain != 0u);
			const bool incur = xr >= 0;
			base = incur ? 0 : -65536;
			x = xn;
			if (x == 0xFFFFu) {
				if (incur && c.k != 0) { x = c.lh_prev[h]; base = -65536; more = more && ((x != 0xFFFFu) || prev_last); }
				else { more = false; }
			}
			xr = base + (int32_t)x;
			alive = more && (uint32_t)((int32_t)p - xr) <= MAXOFF;
		}
		if (best >= 3u) {
			c.mlen3c[p] = (uint16_t)(best - 3u);
			c.moffc[p] = (uint16_t)boff;
			uint32_t len = best;
			const uint32_t elim = (XH && rem < lim) ? rem : lim;       // Xpress+Huffman clips the token to the chunk (:93)
			if (best >= 48u && elim > 48u) {                           // the finder's cap: how long is it really
				const uint32_t xw = pw - boff;
				uint32_t dl = 48;
				for (;;) {
					const uint32_t la = xc_diff16(lds_ld128(s_data, xw + dl), lds_ld128(s_data, pw + dl));
					len = dl + la;
					if (la < 16u || len >= elim) { break; }
					dl += 16u;
				}
			}
			return len < elim ? len : elim;
		}
	}
	return 1u;                                                       // (the offset array was cleared: no match)
}

// Xpress: remember the end of a match longer than 8 KiB (distinct values; open addressing; [XC_TAB + 1] = table overflow)
__device__ __forceinline__ void xc_note_long(uint32_t* s_tab, uint32_t e)
{
	uint32_t slot = (e * 2654435761u) >> 26;
	for (uint32_t k = 0; k < XC_TAB; ++k) {
		const uint32_t old = atomicCAS(&s_tab[slot], 0u, e + 1u);
		if (old == 0u || old == e + 1u) { return; }
		slot = (slot + 1u) & (XC_TAB - 1u);
	}
	s_tab[XC_TAB + 1u] = 1u;
}

// LDS behind the data window: spec bits (2048 words) | Xpress: table of long-match ends (XC_TAB values, 2 words, XC_TAB done flags)
template <uint32_t WIN, bool XH>
__global__ __launch_bounds__(XC_NT) void xp_lazy_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                       const uint16_t* __restrict__ links, const uint16_t* __restrict__ lasthead,
                                                       uint16_t* __restrict__ mlen3, uint16_t* __restrict__ moff, uint32_t chunk_base)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint8_t* const s_data = smem;
	uint32_t* const s_spec = reinterpret_cast<uint32_t*>(smem + WIN + 65536u + XC_PAD);
	uint32_t* const s_tab = s_spec + 2048u;                        // [0..XC_TAB): end + 1 of a long match (0 = free), [XC_TAB + 1]: overflow, then XC_TAB done flags (bytes)
	const uint32_t tid = threadIdx.x;
	const uint32_t lc = chunk_base + blockIdx.x;
	XcChunk c;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	c.k = lc - bt.chunk_prefix[u];
	c.n = bt.in_len[u];
	c.cbase = (u64)c.k * 65536u;
	c.cn = (c.n - c.cbase < 65536u) ? (uint32_t)(c.n - c.cbase) : 65536u;
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	c.lkc = links + (u64)lc * 65536u;
	c.lh_prev = lasthead + (u64)(lc - (c.k ? 1u : 0u)) * 32768u;
	c.mlen3c = mlen3 + (u64)lc * 65536u;
	c.moffc = moff + (u64)lc * 65536u;
	c.prev_last_hash = 0xFFFFFFFFu;
	const u64 wstart = c.cbase >= WIN ? c.cbase - WIN : 0;         // unit position of s_data[0]
	c.crel = (uint32_t)(c.cbase - wstart);
	if (c.k > 0) {
		const u64 q = c.cbase - 1u;
		uint32_t w = 0;
		for (uint32_t j = 0; j < 3u; ++j) { if (q + j < c.n) { w |= (uint32_t)d[q + j] << (8u * j); } }
		c.prev_last_hash = xc_hash3(w);
	}
	const uint32_t cn = c.cn;

	// ---- 0. stage [wstart, chunk end + pad) (zero behind the unit), clear the spec bits and the chunk's offsets ---------------
	{
		const u64 wend = (c.cbase + cn + XC_PAD < c.n) ? c.cbase + cn + XC_PAD : c.n;
		const uint32_t wlen = (uint32_t)(wend - wstart), total = c.crel + cn + XC_PAD;
		const uint8_t* __restrict__ src = d + wstart;
		const uint32_t nvec = (((uintptr_t)src & 15u) == 0) ? (wlen & ~15u) : 0u;
		for (uint32_t i = tid * 16u; i < nvec; i += XC_NT * 16u) { *reinterpret_cast<uint4*>(s_data + i) = *reinterpret_cast<const uint4*>(src + i); }
		for (uint32_t i = nvec + tid; i < wlen; i += XC_NT) { s_data[i] = src[i]; }
		for (uint32_t i = wlen + tid; i < total; i += XC_NT) { s_data[i] = 0; }
		s_spec[tid] = 0; s_spec[tid + XC_NT] = 0;
		if (!XH && tid < XC_TAB + 2u + XC_TAB / 4u) { s_tab[tid] = 0; }
		uint4* __restrict__ mo = reinterpret_cast<uint4*>(c.moffc);   // (chunk arrays start at multiples of 128 KiB)
		for (uint32_t i = tid; i * 8u < cn; i += XC_NT) { mo[i] = make_uint4(0u, 0u, 0u, 0u); }
	}
	__syncthreads();

#define XC_LONG(p_, len_) { if (!XH && (len_) > 0x2000u) { xc_note_long(s_tab, (p_) + (len_)); } }

	// ---- 1. speculative parse of my segment ------------------------------------------------------------------------------------
	uint32_t p = tid * XC_SEG;
	{
		const uint32_t send = (p + XC_SEG < cn) ? p + XC_SEG : cn;
		while (p < send) {
			const uint32_t len = xc_token<XH>(c, s_data, p);
			atomicOr(&s_spec[p >> 5], 1u << (p & 31u));
			XC_LONG(p, len)
			p += len;
		}
		if (p > cn) { p = cn; }
	}
	__syncthreads();
	// ---- 2. on, until I stand on a speculative token start of a later segment ---------------------------------------------------
	while (p < cn && !((s_spec[p >> 5] >> (p & 31u)) & 1u)) {
		const uint32_t len = xc_token<XH>(c, s_data, p);
		XC_LONG(p, len)
		p += len;
	}
	// ---- 3. Xpress: the possible resume points behind matches longer than 8 KiB (lagging Fill, xpress_compress.cpp:269) -------
	if (!XH) {
		uint8_t* const s_done = reinterpret_cast<uint8_t*>(s_tab + XC_TAB + 2u);
		for (;;) {
			__syncthreads();
			const uint32_t slot = tid >> 3;                              // 512 threads: 8 resume points for each of the 64 slots
			const uint32_t v = (tid < XC_TAB * 8u) ? s_tab[slot] : 0u;
			const bool mine = v != 0u && s_done[slot] == 0;
			if (!__syncthreads_or(mine ? 1 : 0)) { break; }
			if (mine) {
				uint32_t q = v + (tid & 7u);                                 // the table holds end + 1: end + 1 .. end + 8 (the end itself is the match's own continuation)
				while (q < cn && !((s_spec[q >> 5] >> (q & 31u)) & 1u)) {
					const uint32_t len = xc_token<XH>(c, s_data, q);
					XC_LONG(q, len)
					q += len;
				}
			}
			__syncthreads();
			if (mine && (tid & 7u) == 0) { s_done[slot] = 1; }
		}
		if (s_tab[XC_TAB + 1u]) {                                       // more than XC_TAB distinct ends: evaluate every position (never seen)
			for (uint32_t q = tid; q < cn; q += XC_NT) { (void)xc_token<XH>(c, s_data, q); }
		}
	}
#undef XC_LONG
}

void launch_xp_lazy(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* links, const uint16_t* lasthead,
                    uint16_t* mlen3, uint16_t* moff, int xh, uint32_t chunk_base, uint32_t chunk_count)
{
	if (chunk_count == 0) { return; }
	static bool attr_set = false;
	const uint32_t lds_xh = 65536u + 65536u + XC_PAD + 8192u;
	const uint32_t lds_xp = 65536u + XC_PAD + 8192u + (XC_TAB + 2u) * 4u + XC_TAB;
	if (!attr_set) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(xp_lazy_kernel<65536u, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_xh);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(xp_lazy_kernel<0u, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_xp);
		attr_set = true;
	}
	if (xh) { hipLaunchKernelGGL((xp_lazy_kernel<65536u, true>), dim3(chunk_count), dim3(XC_NT), lds_xh, st, d_in, bt, links, lasthead, mlen3, moff, chunk_base); }
	else    { hipLaunchKernelGGL((xp_lazy_kernel<0u, false>), dim3(chunk_count), dim3(XC_NT), lds_xp, st, d_in, bt, links, lasthead, mlen3, moff, chunk_base); }
}

} // namespace msc

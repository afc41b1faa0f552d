// xpress_match.hip -- the Xpress / Xpress+Huffman match finder for gfx950: hash-chain links + per-position Find.
//
// Replaces XpressDictionary<MaxOffset,ChunkSize,15,.,3>::Fill / Find / GetMatchLength
// (/root/reference/include/mscomp/XpressDictionary.h:104-118, :145-183, :72-94), used by xpress_compress
// (/root/reference/src/xpress_compress.cpp:269-271) and xh_compress_lz77 (/root/reference/src/xpress_huff_compress.cpp:60,90).
//
// The reference keeps one head table (32768 pointers) + a predecessor ring for the whole buffer and inserts
// positions serially. Here every 64 KiB slice ("link chunk") of a unit is independent:
//   xp_links_kernel : one 1024-thread block per link chunk, head table (32768 x u32 = 128 KiB) in LDS. 15 producer
//                     waves hash the positions of a tile; one consumer wave inserts them 64 at a time with ONE
//                     returning LDS exchange per batch, link[p] = exchange(head[hash(p)], p) (same-address atomics
//                     of one DS instruction are served in lane order, so the serial insertion order is kept); the
//                     links leave as coalesced u16 stores. pred(p) is restricted to the chunk; the chunk's final
//                     head table is exported so the NEXT chunk can continue a chain into it (a chain never needs to
//                     reach further back than one chunk: 65535 / 8192 byte windows).
//   xp_find_kernel  : 4096-position tiles, the tile's window staged in LDS; per position: walks <= 11 links
//                     (MaxChain, Level 3) while inside the window; every candidate is compared 16 bytes at a time
//                     against the position's own 48 bytes held in registers (branch-free first-difference),
//                     strictly-longer wins (nearest on ties), stop at >= 48 (NiceLength). Lengths are CAPPED at 48
//                     here: the candidate choice never depends on more (48 ends the walk); the parse kernels extend
//                     the chosen match. The reference's "never count the buffer's final byte" rule
//                     (XpressDictionary.h:88-93) is the limit n-p-1.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

namespace msc {

__device__ __forceinline__ uint32_t xp_hash3(uint32_t w)      // w = b0 | b1<<8 | b2<<16  (XpressDictionary.h:57-60 closed form)
{
	return (((w & 0x1Fu) << 10) ^ (((w >> 8) & 0xFFu) << 5) ^ ((w >> 16) & 0xFFu)) & 0x7FFFu;
}

// 4 bytes at unit offset pos (any alignment); bytes at or beyond n read as 0
__device__ __forceinline__ uint32_t ldg32_safe(const uint8_t* __restrict__ d, u64 pos, u64 n)
{
	if (pos + 4u <= n) { return ld32(d + pos); }
	uint32_t v = 0;
	for (uint32_t k = 0; k < 4u && pos + k < n; ++k) { v |= (uint32_t)d[pos + k] << (8u * k); }
	return v;
}

#ifdef XL_PROFILE
__device__ unsigned long long g_xl_prof[8];
extern "C" void mscomp_amd_debug_xl_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xl_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xl_prof), z, 64); }
#endif
#ifndef XL_FLIGHT
#define XL_FLIGHT 32u                              // exchanges of the consumer wave in flight (full tiles; 8: 7.32 ms per pass of BASELINE configs[4], 16: 7.00, 32: 6.77)
#endif
#define XL_NC 1u                                   // consumer waves (hash classes); the other 16-XL_NC waves produce hashes
#define XL_NP (16u - XL_NC)
#define XL_NQ ((64u + XL_NP - 1u) / XL_NP)         // batches of 64 positions per producer wave and tile
// LDS of xp_links_kernel: 32768 heads (u32) | hashes of two tiles (u16)
#define XL_HASH_OFF   131072u
#define XL_LDS_BYTES  (XL_HASH_OFF + 2u * 4096u * 2u)

// 16 bytes at ANY byte address of an LDS array whose base is 4-byte aligned. A byte-misaligned ds_read_b128 is replayed
// (SQ_LDS_UNALIGNED_STALL was 77 % of this kernel's LDS cycles), so read 5 ALIGNED dwords and funnel-shift (v_alignbyte).
__device__ __forceinline__ uint4 ld128(const uint8_t* base, uint32_t off)
{
	const uint32_t* a = reinterpret_cast<const uint32_t*>(base + (off & ~3u));
	const uint32_t sh = off & 3u;
	const uint32_t w0 = a[0], w1 = a[1], w2 = a[2], w3 = a[3], w4 = a[4];
	return make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
	                  __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh));
}
// index of the first differing byte of two 16-byte blocks given their XOR (16 if equal)
__device__ __forceinline__ uint32_t first_diff16(const uint4 x) { return first_nz_byte16(x.x, x.y, x.z, x.w); }
// the same as a BIT index (common.h: first_nz_byte16 without its final shift), limited to capbits (<= 128): the limit rides in the minimum
__device__ __forceinline__ uint32_t first_diff_bits16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t capbits)
{
	const uint32_t a = ffbl_raw(x1) | 32u, b = ffbl_raw(x2) | 64u, c = ffbl_raw(x3) | 96u;
	return min3u(min3u(ffbl_raw(x0), a, b), c, capbits);
}

// One 1024-thread block per 64 KiB link chunk (its 128 KiB head table fills the CU's LDS), software-pipelined over
// 4096-position tiles:
//   waves XL_NC..15 (producers): the hash of every position of tile t, straight from global memory, into LDS;
//   waves 0..XL_NC-1 (consumers, one per hash class h mod XL_NC so they never touch the same head): the serial pass
//       over tile t-1 -- per 64 positions ONE returning LDS exchange  link[p] = exchange(head[hash(p)], p).
// Batches are issued in ascending position order and, within one DS instruction, same-address atomics are served in
// lane order (gfx950 behaviour, tools/dev/lds_order_test.hip; every parity test depends on it), so each position
// receives exactly the most recent earlier position with its hash: the chain link of XpressDictionary.h:120-135.
// No conflict detection, no head gather/scatter pairs: 8 exchanges in flight, links leave as coalesced u16 stores.
template <bool serial>    // serial: the exchange one lane at a time (kernels.h set_serial_atomics); a template so that the default kernel is the code it was
__global__ __launch_bounds__(1024) void xp_links_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                      uint16_t* __restrict__ links, uint16_t* __restrict__ lasthead, uint32_t chunk_base)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint32_t* const s_head = reinterpret_cast<uint32_t*>(smem);
	uint16_t* const s_hash = reinterpret_cast<uint16_t*>(smem + XL_HASH_OFF);

	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	const uint32_t lc = chunk_base + blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	const uint32_t k = lc - bt.chunk_prefix[u];
	const u64 n = bt.in_len[u];
	const u64 cbase = (u64)k * 65536u;
	const uint32_t cn = (n - cbase < 65536u) ? (uint32_t)(n - cbase) : 65536u;            // positions in this chunk
	const uint32_t ins = (n >= cbase + 3u) ? ((n - 2u - cbase < cn) ? (uint32_t)(n - 2u - cbase) : cn) : 0u;   // p < n-2
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	uint16_t* __restrict__ lk = links + (u64)lc * 65536u;

	for (uint32_t i = tid * 4u; i < 32768u; i += 4096u) { *reinterpret_cast<uint4*>(s_head + i) = make_uint4(~0u, ~0u, ~0u, ~0u); }
	// positions without a hash (the unit's last two) have no link
	for (uint32_t o = ins + tid; o < cn; o += 1024u) { lk[o] = 0xFFFFu; }

#ifdef XL_PROFILE
	unsigned long long xl_c = 0, xl_p = 0; const unsigned long long xl_t0 = __builtin_readcyclecounter();
#endif
	if (wv < XL_NC) { __builtin_amdgcn_s_setprio(3); }          // the serial consumers are the critical path of the pipeline
	const uint32_t ntiles = (ins + 4095u) >> 12;
	// producers keep the NEXT tile's bytes in flight (registers) while they hash the current one: 15 waves x 5 batches
	// (4800 >= 4096 positions per tile)
	uint32_t v[XL_NQ];
#define XL_LOAD_TILE(tb_) { const uint32_t tn_ = (ins - (tb_) < 4096u) ? ins - (tb_) : 4096u; _Pragma("unroll") for (uint32_t q = 0; q < XL_NQ; ++q) { \
		const uint32_t r_ = q * (XL_NP * 64u) + (wv - XL_NC) * 64u + lane; v[q] = r_ < tn_ ? ldg32_safe(d, cbase + (tb_) + r_, n) : 0u; } }
	if (wv >= XL_NC && ntiles) { XL_LOAD_TILE(0u) }
	for (uint32_t t = 0; t <= ntiles; ++t) {
		const uint32_t tbase = t * 4096u;
#ifdef XL_PROFILE
		const unsigned long long xl_a = __builtin_readcyclecounter();
#endif
		if (wv >= XL_NC) {
			if (t < ntiles) {                                     // ---- producers: hashes of tile t
				const uint32_t tn = (ins - tbase < 4096u) ? ins - tbase : 4096u;
				uint16_t* const hs = s_hash + (t & 1u) * 4096u;
				#pragma unroll
				for (uint32_t q = 0; q < XL_NQ; ++q) {
					const uint32_t r = q * (XL_NP * 64u) + (wv - XL_NC) * 64u + lane;
					if (r < tn) { hs[r] = (uint16_t)xp_hash3(v[q]); }
				}
				if (t + 1u < ntiles) { XL_LOAD_TILE(tbase + 4096u) }
			}
		} else if (t > 0) {                                       // ---- consumers: serial exchange pass over tile t-1 (my hash class)
			const uint32_t pbase = tbase - 4096u;
			const uint32_t tn = (ins - pbase < 4096u) ? ins - pbase : 4096u;
			const uint16_t* const hs = s_hash + ((t - 1u) & 1u) * 4096u;
			if (XL_NC == 1u && tn == 4096u && !serial) {
				// full tile, single consumer: nothing to mask -- per batch one hash read, one exchange, one link store
				for (uint32_t b0 = 0; b0 < 64u; b0 += XL_FLIGHT) {
					uint32_t h[XL_FLIGHT], old[XL_FLIGHT];
					#pragma unroll
					for (uint32_t j = 0; j < XL_FLIGHT; ++j) { h[j] = hs[(b0 + j) * 64u + lane]; }
					#pragma unroll
					for (uint32_t j = 0; j < XL_FLIGHT; ++j) { old[j] = __hip_atomic_exchange(&s_head[h[j]], pbase + (b0 + j) * 64u + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
					#pragma unroll
					for (uint32_t j = 0; j < XL_FLIGHT; ++j) { lk[pbase + (b0 + j) * 64u + lane] = (uint16_t)old[j]; }
				}
			} else
			for (uint32_t b0 = 0; b0 * 64u < tn; b0 += 8u) {
				uint32_t h[8], old[8];
				#pragma unroll
				for (int j = 0; j < 8; ++j) { h[j] = hs[((b0 + j) * 64u + lane) & 4095u]; }              // unconditional: waits once
				#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const uint32_t r = (b0 + j) * 64u + lane;
					old[j] = 0xFFFFu;
					const bool act = r < tn && (h[j] & (XL_NC - 1u)) == wv;
					if (!serial) { if (act) { old[j] = __hip_atomic_exchange(&s_head[h[j]], pbase + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); } }
					else {    // one lane at a time, in lane order (kernels.h): DS operations of a wave execute in program order
						for (uint32_t l = 0; l < 64u; ++l) {
							if (lane == l && act) { old[j] = __hip_atomic_exchange(&s_head[h[j]], pbase + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
							__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
						}
					}
				}
				#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const uint32_t r = (b0 + j) * 64u + lane;
					if (r < tn && (h[j] & (XL_NC - 1u)) == wv) { lk[pbase + r] = (uint16_t)old[j]; }
				}
			}
		}
#ifdef XL_PROFILE
		if (wv == 0) { xl_c += __builtin_readcyclecounter() - xl_a; } else { xl_p += __builtin_readcyclecounter() - xl_a; }
#endif
		__syncthreads();
	}
#ifdef XL_PROFILE
	if (lane == 0 && wv == 0) { atomicAdd(&g_xl_prof[0], xl_c); atomicAdd(&g_xl_prof[2], __builtin_readcyclecounter() - xl_t0); atomicAdd(&g_xl_prof[3], 1ull); }
	if (lane == 0 && wv == XL_NC) { atomicAdd(&g_xl_prof[1], xl_p); }
#endif
	if (lc + 1u < bt.chunk_prefix[u + 1]) {                     // a later chunk of this unit continues chains into this one
		uint16_t* __restrict__ lh = lasthead + (u64)lc * 32768u;
		for (uint32_t i = tid * 2u; i < 32768u; i += 2048u) {
			const uint32_t a0 = s_head[i], a1 = s_head[i + 1u];
			*reinterpret_cast<uint32_t*>(lh + i) = (a0 & 0xFFFFu) | (a1 << 16);
		}
	}
}

// Find for a tile of XP_TILE positions per block. The tile's whole match window is staged in LDS:
//   data  [P0-WINDOW, P0+4096+64)  (WINDOW = 8192 for Xpress, 65536 for Xpress+Huffman; clipped at the unit start),
//   links of the same positions when they fit (Xpress: 24 KiB; Xpress+Huffman would need 136 KiB -> read from L2),
// so the chain walk (<= 11 dependent steps) and the byte compares are LDS gathers instead of L2 gathers.
// clip != 0: positions with fewer than 3 bytes left in their 64 KiB chunk get no match (xpress_huff_compress.cpp:90).
#ifdef XF_PROFILE
__device__ unsigned long long g_xf_prof[8];
extern "C" void mscomp_amd_debug_xf_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xf_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xf_prof), z, 64); }
#define XF_CNT(i, v) xf_acc[i] += (v);
#else
#define XF_CNT(i, v)
#endif
template <uint32_t WINDOW, uint32_t LINKW, uint32_t NT, uint32_t XP_TILE>   // LINKW: how many positions before the tile have their links in LDS
__global__ __launch_bounds__(NT) void xp_find_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                     const uint16_t* __restrict__ links, const uint16_t* __restrict__ lasthead,
                                                     S16 mlen3, S16 moff,
                                                     uint32_t max_off, int clip, uint32_t tile_base)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint8_t* const s_data = smem;                                             // WINDOW + XP_TILE + 64 bytes
	uint16_t* const s_links = reinterpret_cast<uint16_t*>(smem + WINDOW + XP_TILE + 64u);   // LINKW + XP_TILE entries

	const uint32_t tid = threadIdx.x;
	// XCD-aware tile order: consecutive workgroup ids go to different XCDs (private L2s). Inside every group of 128 ids
	// XCD x takes 16 consecutive tiles (one or two chunks), which re-stage the same window, so they share one L2;
	// across groups the chunks stay round-robin over the XCDs (cheap and expensive files are spread evenly).
	uint32_t bid = blockIdx.x;
	if ((bid | 127u) < gridDim.x) { const uint32_t wi = bid & 127u; bid = (bid & ~127u) + (wi & 7u) * 16u + (wi >> 3); }
	bid += tile_base;                                                         // (a launch over a RANGE of chunks: the pipelined one-shot call, api.hip)
	constexpr uint32_t TPC = 65536u / XP_TILE;                                // tiles per chunk
	const uint32_t lc = bid / TPC;
	const uint32_t tstart = (bid % TPC) * XP_TILE;                            // tile start inside the chunk
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	const uint32_t k = lc - bt.chunk_prefix[u];
	const u64 n = bt.in_len[u];
	const u64 cbase = (u64)k * 65536u;
	const uint32_t cn = (n - cbase < 65536u) ? (uint32_t)(n - cbase) : 65536u;
	if (tstart >= cn) { return; }
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	const u64 P0 = cbase + tstart;                                            // unit position of the tile start
	const u64 wstart = P0 >= WINDOW ? P0 - WINDOW : 0;                        // unit position of s_data[0]
	const u64 wend = (P0 + XP_TILE + 64u < n) ? P0 + XP_TILE + 64u : n;       // staged bytes: [wstart, wend), zero beyond
	const uint32_t wlen = (uint32_t)(wend - wstart);
	const u64 lwstart = P0 >= LINKW ? P0 - LINKW : 0;                         // unit position of s_links[0] (>= wstart)

	// ---- stage data (16 B / thread when aligned) and links ------------------------------------------------------
	{
		const uint8_t* __restrict__ src = d + wstart;
		const uint32_t nvec = (((uintptr_t)src & 15u) == 0) ? (wlen & ~15u) : 0u;
		for (uint32_t i = tid * 16u; i < nvec; i += NT * 16u) { *reinterpret_cast<uint4*>(s_data + i) = *reinterpret_cast<const uint4*>(src + i); }
		for (uint32_t i = nvec + tid; i < wlen; i += NT) { s_data[i] = src[i]; }
		for (uint32_t i = wlen + tid; i < WINDOW + XP_TILE + 64u; i += NT) { s_data[i] = 0; }
		if (LINKW == WINDOW) {
			const uint32_t npos = (uint32_t)((P0 + XP_TILE < cbase + cn ? P0 + XP_TILE : cbase + cn) - lwstart);
			// the link arrays of a unit's chunks are contiguous, lwstart is a multiple of 4096 and s_links is 16-byte
			// aligned: 8 links per load (entries beyond npos are never read)
			const uint16_t* __restrict__ lsrc = links + (u64)bt.chunk_prefix[u] * 65536u + lwstart;
			for (uint32_t r = tid * 8u; r < npos; r += NT * 8u) {
				*reinterpret_cast<uint4*>(s_links + r) = *reinterpret_cast<const uint4*>(lsrc + r);
			}
		}
	}
	__syncthreads();

	// links of this chunk; the previous chunk's array directly precedes it, so a window-relative position xr is entry
	// lkw[xr] of an array that starts where the window starts (unsigned 32-bit index: scalar base + lane offset)
	const uint16_t* __restrict__ lkg = links + (u64)lc * 65536u;
	const uint16_t* __restrict__ lkw = lkg - (ptrdiff_t)(cbase - wstart); (void)lkw;   // (dev probes only)
	const uint16_t* __restrict__ lh_prev = lasthead + (u64)(lc - (k ? 1u : 0u)) * 32768u;
	const uint32_t tn = (cn - tstart < XP_TILE) ? cn - tstart : XP_TILE;
	// 65535 as the previous chunk's head is indistinguishable from "none" (0xFFFF): decide by that position's hash
	const uint32_t prev_last_hash = (k > 0) ? xp_hash3(ldg32_safe(d, cbase - 1u, n)) : 0xFFFFFFFFu;
	// everything below is 32-bit and relative to the staged window (s_data[0] = unit position wstart)
	const int32_t crel = (int32_t)(uint32_t)(cbase - wstart);                 // start of this chunk   (>= 0)
	const uint32_t lrel = (uint32_t)(lwstart - wstart);                       // first position whose link is in s_links
	const uint32_t p0r = (uint32_t)(P0 - wstart);
	const u64 tail = n - P0;                                                  // bytes from the tile start to the unit end

#ifdef XF_PROFILE
	unsigned long long xf_acc[8] = {0};
#endif
	// The walk in numbers relative to THIS chunk's start: rc = candidate position (< 0: in the previous chunk), o = my own; a link value is an
	// offset inside its position's chunk, so the next candidate is x + adj with adj = 0 until the chain crosses into the previous chunk
	// (-65536 from then on). Best so far as ONE signed key = (len << 16) - distance: the maximum is the longest and, among equals, the
	// nearest = the first met (XpressDictionary.h:164-176 takes strictly longer only); len < 3 never beats the start value 2 << 16.
	// The chain counter is the same in every lane still walking, so it lives on the scalar unit. (Round 5: 49 -> 34 vector instructions
	// per candidate -- and no time gained, 65.1 -> 64.9 ms: a step of the 64 KiB-window instance is the divergent link gather, one L2
	// request per lane -- 1.1e10 per pass of configs[4], the L1 is stalled on pending misses 60 % of the time -- DESIGN 8.)
	const int32_t nmax = -(int32_t)max_off;
	const uint16_t* __restrict__ lkb = lkg - 65536;                            // links by (rc + 65536): the previous chunk's array directly precedes this one
	constexpr int32_t K48 = (47 << 16) + 1;                                    // key >= K48 <=> len == 48 (NiceLength: the walk ends)
	for (uint32_t t = tid; t < tn; t += NT) {
		const uint32_t o = tstart + t;                                          // offset in chunk
		const uint32_t pr = p0r + t;                                            // window-relative position of P
		int32_t key = 2 << 16;
		const bool can = ((u64)t + 2u < tail) && (!clip || cn - o >= 3u);
#if defined(XF_PROBE) && XF_PROBE == 4       /* dev probe (SUBTRACTIVE, not bit-exact): walk 6 chain candidates instead of 11 */
		uint32_t chain = 6;
#else
		uint32_t chain = 11;
#endif
		if (can) {
			const uint4 oa = ld128(s_data, pr), ob = ld128(s_data, pr + 16u), oc = ld128(s_data, pr + 32u);
			const uint32_t h = xp_hash3(oa.x);
			const u64 lim = tail - t - 1u;                                          // n - P - 1: never count the buffer's final byte
			const uint32_t cap = lim < 48u ? (uint32_t)lim : 48u;
			const uint32_t capb = (cap < 16u ? cap : 16u) << 3;                     // the cap on the first 16 bytes, in bits
			const bool prev_last = (prev_last_hash == h);
			// first candidate: my own link (an offset inside the chunk), else the previous chunk's last position with my hash
			int32_t rc, adj = 0, negd;
			bool alive;
			{
				uint32_t x = (LINKW == WINDOW) ? (uint32_t)s_links[pr - lrel] : (uint32_t)lkg[o];
				alive = true;
				if (x == 0xFFFFu) {
					if (k == 0) { alive = false; }
					else { x = lh_prev[h]; adj = -65536; alive = (x != 0xFFFFu) || prev_last; }
				}
				rc = (int32_t)x + adj;
				negd = rc - (int32_t)o;                                             // -(distance)
				alive = alive && negd >= nmax;                                      // (also catches a candidate in front of the window)
			}
			XF_CNT(0, 1)
			// One candidate per iteration, straight-line: every live lane does the 16-byte compare (in a wave some lane
			// always needs it, so a per-lane prefix filter only adds instructions); the reference compares everything too
			// (XpressDictionary.h:164-176).
			while (alive) {
				XF_CNT(1, 1)
				if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (uint32_t)__builtin_ctzll(__ballot(1))) { XF_CNT(2, 64) }
				const uint32_t xr = (uint32_t)(rc + crel);                         // window-relative
				// the link of the candidate is fetched first (L2 latency for Xpress+Huffman) and consumed after the compare
				// (byte offset in 32 bits, zero-extended: the gather is `global_load_ushort v, v_offset, s[base]`)
				uint32_t x;
#if defined(XF_PROBE) && XF_PROBE == 7        /* dev probe (SUBTRACTIVE, not bit-exact): the link of ANOTHER position, chosen so that the 64 lanes of a wave read consecutive links (a coalesced load instead of a divergent gather; the chain then walks other, equally real, positions) */
				x = (uint32_t)lkw[((uint32_t)pr - 1u - chain) & 0xFFFFu];
				if (false)
#endif
				// (cache-scope variants of this gather: non-temporal 126.7 ms, sc1 / sc0 sc1 78.7, sc0 65.1, plain 64.9: the L1 serves ~60 % of the lanes)
#if defined(XF_PROBE) && XF_PROBE == 8        /* dev probe (not bit-exact): NO link load and no dependent chain -- the next candidate lies 37 bytes further back; the compares are the product's */
				x = (uint32_t)(rc - adj - 37) & 0xFFFFu; if (rc - adj < 37) { x = 0xFFFFu; }
				if (false)
#elif defined(XF_PROBE) && XF_PROBE == 9      /* dev probe (not bit-exact): the link gather on every SECOND step only (what a two-links-per-gather layout would issue), arithmetic in between */
				x = (uint32_t)(rc - adj - 37) & 0xFFFFu; if (rc - adj < 37) { x = 0xFFFFu; }
				if (chain & 1u)
#endif
				x = (LINKW == WINDOW) ? (uint32_t)s_links[xr - lrel]
				                      : (uint32_t)*reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(lkb) + (u64)(uint32_t)((rc << 1) + 131072));
				// 16 bytes per step against my own 48 bytes held in registers (lengths are capped at 48 here)
				uint4 c = ld128(s_data, xr);
#if defined(XF_PROBE) && XF_PROBE == 1        /* dev probe: one more divergent global gather per step */
				{ const uint32_t y = lkw[((uint32_t)xr * 7u + 13u) & 0xFFFFu]; asm volatile("" :: "v"(y)); }
#elif defined(XF_PROBE) && XF_PROBE == 2      /* dev probe: ten more VALU instructions per step */
				{ uint32_t y = (uint32_t)negd; _Pragma("unroll") for (int q_ = 0; q_ < 10; ++q_) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"(c.x)); } asm volatile("" :: "v"(y)); }
#elif defined(XF_PROBE) && XF_PROBE == 3      /* dev probe: five more LDS dword reads per step */
				{ const uint4 y = ld128(s_data, ((uint32_t)xr * 5u + 77u) & 0xFFFFu); asm volatile("" :: "v"(y.x), "v"(y.y), "v"(y.z), "v"(y.w)); }
#endif
				// matched length in BITS, capped inside the minimum that finds the first difference (the low three bits are dropped below)
				uint32_t lb = first_diff_bits16(c.x ^ oa.x, c.y ^ oa.y, c.z ^ oa.z, c.w ^ oa.w, capb);
#if defined(XF_PROBE) && XF_PROBE == 5       /* dev probe (SUBTRACTIVE, not bit-exact): no compare beyond the first 16 bytes */
				if (false) {
#else
				if (lb >= 128u && cap > 16u) {
#endif
					c = ld128(s_data, xr + 16u);
					uint32_t l = 16u + first_diff16(make_uint4(c.x ^ ob.x, c.y ^ ob.y, c.z ^ ob.z, c.w ^ ob.w));
					if (l == 32u && cap > 32u) {
						c = ld128(s_data, xr + 32u);
						l = 32u + first_diff16(make_uint4(c.x ^ oc.x, c.y ^ oc.y, c.z ^ oc.z, c.w ^ oc.w));
					}
					lb = (l < cap ? l : cap) << 3;
				}
				const int32_t kc = (int32_t)((lb & ~7u) << 13) + negd;               // (len << 16) - distance
				key = kc > key ? kc : key;
				// next candidate: x is an offset inside the candidate's chunk (a link is always < its position, so 0xFFFF is unambiguous)
				if (--chain == 0u) { break; }                                          // MaxChain (uniform)
				bool more = key < K48;
				if (x == 0xFFFFu) {
					if (rc >= 0 && k != 0) { x = lh_prev[h]; adj = -65536; more = more && ((x != 0xFFFFu) || prev_last); }
					else { more = false; }
				}
				rc = (int32_t)x + adj;
				negd = rc - (int32_t)o;
				alive = more && negd >= nmax;
			}
		}
		const uint32_t best = (uint32_t)(key + 65535) >> 16, boff = (best << 16) - (uint32_t)key;
		const bool m = best >= 3u;
		const u64 gi = (u64)lc * 65536u + o;
		*reinterpret_cast<uint32_t*>(&mlen3[gi]) = m ? (best - 3u) | (boff << 16) : 0u;      // one word: length - 3 | offset << 16 (moff is its upper half)
		XF_CNT(4, m ? 1 : 0)
	}
#ifdef XF_PROFILE
	for (int i_ = 0; i_ < 5; ++i_) { if (xf_acc[i_]) { atomicAdd(&g_xf_prof[i_], xf_acc[i_]); } }
#endif
}


void launch_xp_links_range(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint16_t* links, uint16_t* lasthead, uint32_t chunk_base, uint32_t chunk_count)
{
	if (chunk_count == 0) { return; }
	const uint32_t lds = XL_LDS_BYTES;
	static PerDeviceOnce attr;
	if (attr.needed()) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(xp_links_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(xp_links_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		attr.done();
	}
	if (serial_atomics_on_current_device()) { hipLaunchKernelGGL(xp_links_kernel<true>, dim3(chunk_count), dim3(1024), lds, st, d_in, bt, links, lasthead, chunk_base); }
	else { hipLaunchKernelGGL(xp_links_kernel<false>, dim3(chunk_count), dim3(1024), lds, st, d_in, bt, links, lasthead, chunk_base); }
}
void launch_xp_links(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint16_t* links, uint16_t* lasthead)
{
	launch_xp_links_range(st, d_in, bt, links, lasthead, 0u, bt.n_chunks);
}
// tile = positions per block: 4096 for Xpress (4 blocks/CU), 8192 for Xpress+Huffman (the 64 KiB window is re-staged half as
// often; 72 KiB of LDS, still 2 blocks/CU)
#define XH_TILE_SEL 8192u
void launch_xp_find_range(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* links, const uint16_t* lasthead,
                          uint16_t* mlen3, uint16_t* moff, uint32_t max_off, int clip, uint32_t chunk_base, uint32_t chunk_count)
{
	if (chunk_count == 0) { return; }
	static PerDeviceOnce attr;
	constexpr uint32_t TXP = 4096u, TXH = XH_TILE_SEL;
	const uint32_t lds_xp = 0x2000u + TXP + 64u + (0x2000u + TXP) * 2u;           // data + all links of the window in LDS
	static const uint32_t xh_pad = [] { const char* e = getenv("MSCOMP_AMD_XF_LDS_PAD_KB"); return e ? (uint32_t)atoi(e) * 1024u : 0u; }();   // dev: more LDS per block = fewer blocks per CU (occupancy probe)
	const uint32_t lds_xh = 0x10000u + TXH + 64u + xh_pad;                        // data only (2 blocks/CU); links come from L2
	if (attr.needed()) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(xp_find_kernel<0x2000u, 0x2000u, 512u, TXP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_xp);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(xp_find_kernel<0x10000u, 0u, 1024u, TXH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_xh);
		attr.done();
	}
	if (max_off <= 0x2000u) {
		hipLaunchKernelGGL((xp_find_kernel<0x2000u, 0x2000u, 512u, TXP>), dim3(chunk_count * (65536u / TXP)), dim3(512), lds_xp, st, d_in, bt, links, lasthead, mlen3, moff, max_off, clip, chunk_base * (65536u / TXP));
	} else {
		hipLaunchKernelGGL((xp_find_kernel<0x10000u, 0u, 1024u, TXH>), dim3(chunk_count * (65536u / TXH)), dim3(1024), lds_xh, st, d_in, bt, links, lasthead, mlen3, moff, max_off, clip, chunk_base * (65536u / TXH));
	}
}
void launch_xp_find(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* links, const uint16_t* lasthead,
                    uint16_t* mlen3, uint16_t* moff, uint32_t max_off, int clip)
{
	launch_xp_find_range(st, d_in, bt, links, lasthead, mlen3, moff, max_off, clip, 0u, bt.n_chunks);
}

} // namespace msc

// xpress_sort.hip -- the Xpress+Huffman match finder WITHOUT the dependent chain walk (round 5): positions sorted by (hash, position),
// a position's chain candidates read as one span of that array.
//
// Replaces, like xpress_match.hip, XpressDictionary<0x10000,...>::Fill / Find / GetMatchLength
// (/root/reference/include/mscomp/XpressDictionary.h:104-118, :145-183, :72-94) as called by xh_compress_lz77
// (/root/reference/src/xpress_huff_compress.cpp:60,90). Same results as xp_links_kernel + xp_find_kernel.
//
// xp_find_kernel follows the hash chain link by link: 11 DEPENDENT L2 round trips per position, and 32 waves per CU cannot cover them
// (time = chain steps; the same kernel with arithmetic "links" runs 35 % faster). But the chain of a position is nothing
// else than the earlier positions with its hash, nearest first -- in the chunk, then in the previous chunk (a link never reaches
// further: xpress_match.hip). So:
//   xp_sort_kernel  : one 1024-thread block per 64 KiB chunk. (1) the ordered pass of xp_links_kernel, with a returning LDS ADD on a
//                     zeroed counter instead of the exchange: the value returned is the position's RANK r among the chunk's
//                     positions with its hash (same-address atomics of one DS instruction are served in lane order, batches
//                     ascend); (2) exclusive scan of the 32768 counters = bucket starts, exported (the NEXT chunk's finder needs
//                     where this chunk's buckets end); (3) sorted[start[h] + r] = p, and per position the word
//                     index-in-sorted | min(r, 15) << 16 -- written where the finder will put the position's match word.
//   xp_find2_kernel : xp_find_kernel's tiles (window in LDS, lane = position), but a lane's <= 11 candidates are
//                     sorted[i-1], sorted[i-2], ... (at most r of them: they share my bucket) and then the LAST entries of my
//                     bucket in the previous chunk's array: two 24-byte reads whose addresses depend only on the position's own
//                     word -- all candidates known at once, merged into one 11-entry list in registers by funnel shifts; the
//                     compares are xp_find_kernel's.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

namespace msc {

__device__ __forceinline__ uint32_t xs_hash3(uint32_t w)      // (XpressDictionary.h:57-60 closed form, as xpress_match.hip)
{
	return (((w & 0x1Fu) << 10) ^ (((w >> 8) & 0xFFu) << 5) ^ ((w >> 16) & 0xFFu)) & 0x7FFFu;
}
__device__ __forceinline__ uint32_t xs_ldg32_safe(const uint8_t* __restrict__ d, u64 pos, u64 n)
{
	if (pos + 4u <= n) { return ld32(d + pos); }
	uint32_t v = 0;
	for (uint32_t k = 0; k < 4u && pos + k < n; ++k) { v |= (uint32_t)d[pos + k] << (8u * k); }
	return v;
}
__device__ __forceinline__ uint4 xs_ld128(const uint8_t* base, uint32_t off)   // (common.h lds_ld128)
{
	const uint32_t* a = reinterpret_cast<const uint32_t*>(base + (off & ~3u));
	const uint32_t sh = off & 3u;
	const uint32_t w0 = a[0], w1 = a[1], w2 = a[2], w3 = a[3], w4 = a[4];
	return make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
	                  __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh));
}
__device__ __forceinline__ uint32_t xs_diff_bits16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t capbits)
{
	const uint32_t a = ffbl_raw(x1) | 32u, b = ffbl_raw(x2) | 64u, c = ffbl_raw(x3) | 96u;
	return min3u(min3u(ffbl_raw(x0), a, b), c, capbits);
}

#define XS_NP 15u                                  // producer waves (wave 0 consumes)
#define XS_NQ ((64u + XS_NP - 1u) / XS_NP)
#define XS_FLIGHT 32u
#define XS_HASH_OFF 131072u                        // LDS: 32768 counters (u32) | hashes of two tiles (u16)
#define XS_LDS_BYTES (XS_HASH_OFF + 2u * 4096u * 2u)

template <bool serial>
__global__ __launch_bounds__(1024) void xp_sort_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                     uint16_t* __restrict__ sorted, uint32_t* __restrict__ words, uint32_t* __restrict__ starts)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint32_t* const s_cnt = reinterpret_cast<uint32_t*>(smem);
	uint16_t* const s_hash = reinterpret_cast<uint16_t*>(smem + XS_HASH_OFF);
	uint32_t* const s_tot = reinterpret_cast<uint32_t*>(smem + XS_HASH_OFF);      // (after the ordered pass) 16 wave totals

	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	const uint32_t lc = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	const uint32_t k = lc - bt.chunk_prefix[u];
	const u64 n = bt.in_len[u];
	const u64 cbase = (u64)k * 65536u;
	const uint32_t cn = (n - cbase < 65536u) ? (uint32_t)(n - cbase) : 65536u;            // positions in this chunk
	const uint32_t ins = (n >= cbase + 3u) ? ((n - 2u - cbase < cn) ? (uint32_t)(n - 2u - cbase) : cn) : 0u;   // positions that have a hash: p < n-2
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	uint32_t* __restrict__ wd = words + (u64)lc * 65536u;
	uint16_t* __restrict__ so = sorted + (u64)lc * 65536u;

	for (uint32_t i = tid * 4u; i < 32768u; i += 4096u) { *reinterpret_cast<uint4*>(s_cnt + i) = make_uint4(0u, 0u, 0u, 0u); }
	for (uint32_t o = ins + tid; o < cn; o += 1024u) { wd[o] = 0u; }       // (positions without a hash are never looked up)

	// ---- (1) ranks: the pipeline of xp_links_kernel -- producers hash tile t while wave 0 runs the ordered pass over tile t-1
	if (wv == 0) { __builtin_amdgcn_s_setprio(3); }
	const uint32_t ntiles = (ins + 4095u) >> 12;
	uint32_t v[XS_NQ];
#define XS_LOAD_TILE(tb_) { const uint32_t tn_ = (ins - (tb_) < 4096u) ? ins - (tb_) : 4096u; _Pragma("unroll") for (uint32_t q = 0; q < XS_NQ; ++q) { \
		const uint32_t r_ = q * (XS_NP * 64u) + (wv - 1u) * 64u + lane; v[q] = r_ < tn_ ? xs_ldg32_safe(d, cbase + (tb_) + r_, n) : 0u; } }
	if (wv >= 1u && ntiles) { XS_LOAD_TILE(0u) }
	__syncthreads();
	for (uint32_t t = 0; t <= ntiles; ++t) {
		const uint32_t tbase = t * 4096u;
		if (wv >= 1u) {
			if (t < ntiles) {
				const uint32_t tn = (ins - tbase < 4096u) ? ins - tbase : 4096u;
				uint16_t* const hs = s_hash + (t & 1u) * 4096u;
				#pragma unroll
				for (uint32_t q = 0; q < XS_NQ; ++q) {
					const uint32_t r = q * (XS_NP * 64u) + (wv - 1u) * 64u + lane;
					if (r < tn) { hs[r] = (uint16_t)xs_hash3(v[q]); }
				}
				if (t + 1u < ntiles) { XS_LOAD_TILE(tbase + 4096u) }
			}
		} else if (t > 0) {
			const uint32_t pbase = tbase - 4096u;
			const uint32_t tn = (ins - pbase < 4096u) ? ins - pbase : 4096u;
			const uint16_t* const hs = s_hash + ((t - 1u) & 1u) * 4096u;
			if (tn == 4096u && !serial) {
				for (uint32_t b0 = 0; b0 < 64u; b0 += XS_FLIGHT) {
					uint32_t h[XS_FLIGHT], old[XS_FLIGHT];
					#pragma unroll
					for (uint32_t j = 0; j < XS_FLIGHT; ++j) { h[j] = hs[(b0 + j) * 64u + lane]; }
					#pragma unroll
					for (uint32_t j = 0; j < XS_FLIGHT; ++j) { old[j] = __hip_atomic_fetch_add(&s_cnt[h[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
					#pragma unroll
					for (uint32_t j = 0; j < XS_FLIGHT; ++j) { wd[pbase + (b0 + j) * 64u + lane] = h[j] | (old[j] << 16); }
				}
			} else
			for (uint32_t b0 = 0; b0 * 64u < tn; b0 += 8u) {
				uint32_t h[8], old[8];
				#pragma unroll
				for (int j = 0; j < 8; ++j) { h[j] = hs[((b0 + j) * 64u + lane) & 4095u]; }
				#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const uint32_t r = (b0 + j) * 64u + lane;
					old[j] = 0u;
					const bool act = r < tn;
					if (!serial) { if (act) { old[j] = __hip_atomic_fetch_add(&s_cnt[h[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); } }
					else {    // one lane at a time, in lane order (kernels.h): DS operations of a wave execute in program order
						for (uint32_t l = 0; l < 64u; ++l) {
							if (lane == l && act) { old[j] = __hip_atomic_fetch_add(&s_cnt[h[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
							__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
						}
					}
				}
				#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const uint32_t r = (b0 + j) * 64u + lane;
					if (r < tn) { wd[pbase + r] = h[j] | (old[j] << 16); }
				}
			}
		}
		__syncthreads();
	}
#undef XS_LOAD_TILE
	if (wv == 0) { __builtin_amdgcn_s_setprio(0); }

	// ---- (2) exclusive scan of the 32768 counts -> bucket starts (in place), exported with the total behind them.
	// Wave w owns the counters [2048 w, 2048 w + 2048): its total first, then its part of the scan from the sum of the totals before it.
	uint32_t* __restrict__ st = starts + (u64)lc * XS_STARTS_STRIDE;
	{
		uint32_t sum = 0;
		#pragma unroll
		for (uint32_t rnd = 0; rnd < 8u; ++rnd) {
			const uint4 a = *reinterpret_cast<const uint4*>(s_cnt + wv * 2048u + rnd * 256u + lane * 4u);
			sum += a.x + a.y + a.z + a.w;
		}
		const uint32_t incl = wave_incl_scan_add_u32(sum);
		if (lane == 63u) { s_tot[wv] = incl; }
	}
	__syncthreads();
	{
		uint32_t run = 0;
		for (uint32_t w = 0; w < wv; ++w) { run += s_tot[w]; }
		#pragma unroll 1
		for (uint32_t rnd = 0; rnd < 8u; ++rnd) {
			uint32_t* const q = s_cnt + wv * 2048u + rnd * 256u + lane * 4u;
			const uint4 a = *reinterpret_cast<const uint4*>(q);
			const uint32_t sum = a.x + a.y + a.z + a.w;
			const uint32_t incl = wave_incl_scan_add_u32(sum);
			const uint32_t e0 = run + incl - sum;
			const uint4 ex = make_uint4(e0, e0 + a.x, e0 + a.x + a.y, e0 + a.x + a.y + a.z);
			*reinterpret_cast<uint4*>(q) = ex;
			*reinterpret_cast<uint4*>(st + wv * 2048u + rnd * 256u + lane * 4u) = ex;
			run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		}
	}
	if (tid == 0) { st[32768] = ins; }
	__syncthreads();

	// ---- (3) scatter: sorted[start[h] + r] = p; the position's word becomes  index | min(r, 15) << 16
	for (uint32_t o0 = 0; o0 < ins; o0 += 8192u) {
		uint32_t w[8];
		#pragma unroll
		for (uint32_t j = 0; j < 8u; ++j) { const uint32_t o = o0 + j * 1024u + tid; w[j] = o < ins ? wd[o] : 0u; }
		#pragma unroll
		for (uint32_t j = 0; j < 8u; ++j) {
			const uint32_t o = o0 + j * 1024u + tid;
			if (o < ins) {
				const uint32_t r = w[j] >> 16;
				const uint32_t i = s_cnt[w[j] & 0x7FFFu] + r;
				so[i] = (uint16_t)o;
				wd[o] = i | ((r < 15u ? r : 15u) << 16);
			}
		}
	}
}

// Find for tiles of XP_TILE positions of the 64 KiB-window formats (xpress_match.hip xp_find_kernel<65536, 0, 1024, 8192>: same staging, same
// compares, same output word); the candidates come from the sorted arrays.
template <uint32_t NT, uint32_t XP_TILE>
__global__ __launch_bounds__(NT) void xp_find2_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                      const uint16_t* __restrict__ sorted, const uint32_t* __restrict__ starts,
                                                      uint32_t* __restrict__ words,       // in: index | rank << 16 (xp_sort_kernel); out: the match word (len - 3 | offset << 16)
                                                      uint32_t max_off, int clip)
{
	constexpr uint32_t WINDOW = 65536u;
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint8_t* const s_data = smem;                                             // WINDOW + XP_TILE + 64 bytes

	const uint32_t tid = threadIdx.x;
	uint32_t bid = blockIdx.x;                                                // (XCD-aware tile order: xp_find_kernel)
	if ((bid | 127u) < gridDim.x) { const uint32_t wi = bid & 127u; bid = (bid & ~127u) + (wi & 7u) * 16u + (wi >> 3); }
	constexpr uint32_t TPC = 65536u / XP_TILE;
	const uint32_t lc = bid / TPC;
	const uint32_t tstart = (bid % TPC) * XP_TILE;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	const uint32_t k = lc - bt.chunk_prefix[u];
	const u64 n = bt.in_len[u];
	const u64 cbase = (u64)k * 65536u;
	const uint32_t cn = (n - cbase < 65536u) ? (uint32_t)(n - cbase) : 65536u;
	if (tstart >= cn) { return; }
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	const u64 P0 = cbase + tstart;
	const u64 wstart = P0 >= WINDOW ? P0 - WINDOW : 0;
	const u64 wend = (P0 + XP_TILE + 64u < n) ? P0 + XP_TILE + 64u : n;
	const uint32_t wlen = (uint32_t)(wend - wstart);
	{
		const uint8_t* __restrict__ src = d + wstart;
		const uint32_t nvec = (((uintptr_t)src & 15u) == 0) ? (wlen & ~15u) : 0u;
		for (uint32_t i = tid * 16u; i < nvec; i += NT * 16u) { *reinterpret_cast<uint4*>(s_data + i) = *reinterpret_cast<const uint4*>(src + i); }
		for (uint32_t i = nvec + tid; i < wlen; i += NT) { s_data[i] = src[i]; }
		for (uint32_t i = wlen + tid; i < WINDOW + XP_TILE + 64u; i += NT) { s_data[i] = 0; }
	}
	__syncthreads();

	const uint32_t tn = (cn - tstart < XP_TILE) ? cn - tstart : XP_TILE;
	const int32_t crel = (int32_t)(uint32_t)(cbase - wstart);                 // start of this chunk in the staged window
	const uint32_t p0r = (uint32_t)(P0 - wstart);
	const u64 tail = n - P0;
	const int32_t nmax = -(int32_t)max_off;
	constexpr int32_t K48 = (47 << 16) + 1;                                    // key >= K48 <=> len == 48 (NiceLength: the walk ends)
	// byte bases of the two sorted arrays, 24 bytes (12 entries) in FRONT of entry 0 so that a span that starts before the array needs no clamp
	// (the arrays are laid out chunk after chunk behind XS_FRONT_PAD entries of padding; what lies in front of entry 0 is never selected)
	const uint8_t* __restrict__ sa = reinterpret_cast<const uint8_t*>(sorted + (u64)lc * 65536u) - 24;
	const uint8_t* __restrict__ sb = sa - 131072;
	const uint32_t* __restrict__ stp = starts + (u64)(lc - (k ? 1u : 0u)) * XS_STARTS_STRIDE;
	uint32_t* __restrict__ wd = words + (u64)lc * 65536u;

	for (uint32_t t = tid; t < tn; t += NT) {
		const uint32_t o = tstart + t;
		const uint32_t pr = p0r + t;
		int32_t key = 2 << 16;
		const bool can = ((u64)t + 2u < tail) && (!clip || cn - o >= 3u);
		if (can) {
			const uint32_t ri = wd[o];
			const uint4 oa = xs_ld128(s_data, pr), ob = xs_ld128(s_data, pr + 16u), oc = xs_ld128(s_data, pr + 32u);
			const uint32_t h = xs_hash3(oa.x);
			const u64 lim = tail - t - 1u;                                          // n - P - 1: never count the buffer's final byte
			const uint32_t cap = lim < 48u ? (uint32_t)lim : 48u;
			const uint32_t capb = (cap < 16u ? cap : 16u) << 3;
			const uint32_t i = ri & 0xFFFFu, r = ri >> 16;
			const uint32_t nin = r < 11u ? r : 11u;                                 // candidates in this chunk: sorted[i-1] ... sorted[i-nin]
			// ---- span A: the 12 entries that end with entry i-1 or i (the even entry index at or below i-11 starts a dword)
			// byte offset from sa: 2 * ((i + 1) & ~1)  [= 2 * (((i - 11) & ~1) + 12)]
			uint32_t A[7], B[7];
			{
				const uint32_t* __restrict__ pa = reinterpret_cast<const uint32_t*>(sa + (((i + 1u) & ~1u) << 1));
				uint4 a0; uint2 a1;                                                   // (only 4-byte aligned: loads that say so -- ADVICE r05; the compiler still issues dwordx4 + dwordx2)
				__builtin_memcpy(&a0, __builtin_assume_aligned(pa, 4), 16); __builtin_memcpy(&a1, __builtin_assume_aligned(pa + 4, 4), 8);
				A[0] = a0.x; A[1] = a0.y; A[2] = a0.z; A[3] = a0.w; A[4] = a1.x; A[5] = a1.y; A[6] = 0u;
			}
			// ---- span B: the last entries of my bucket in the previous chunk's array
			uint32_t npv = 0, eprev = 0;
			if (k != 0 && nin < 11u) {
				uint2 se; __builtin_memcpy(&se, __builtin_assume_aligned(stp + h, 4), 8);   // start of bucket h, start of bucket h + 1 (or the total)
				eprev = se.y;
				const uint32_t avail = se.y - se.x;
				npv = (11u - nin) < avail ? (11u - nin) : avail;
			}
			B[0] = B[1] = B[2] = B[3] = B[4] = B[5] = 0u; B[6] = 0u;
			if (npv) {
				const uint32_t* __restrict__ pb = reinterpret_cast<const uint32_t*>(sb + (((eprev + 1u) & ~1u) << 1));
				uint4 b0; uint2 b1;
				__builtin_memcpy(&b0, __builtin_assume_aligned(pb, 4), 16); __builtin_memcpy(&b1, __builtin_assume_aligned(pb + 4, 4), 8);
				B[0] = b0.x; B[1] = b0.y; B[2] = b0.z; B[3] = b0.w; B[4] = b1.x; B[5] = b1.y;
			}
			// ---- normalise: An halfword t = entry i - 11 + t (t = 0..10); the span starts at entry (i - 11) & ~1, one entry early when i is even
			// [(i - 11) odd]. The previous chunk's span is shifted down by its own parity PLUS nin entries: Bs halfword t = entry e' - 11 + t + nin.
			const uint32_t sha = ((i & 1u) ^ 1u) << 4;
			const uint32_t sB = ((eprev & 1u) ^ 1u) + nin;                          // halfwords (0..12)
			const uint32_t shb = (sB & 1u) << 4, qB = sB >> 1;
			uint32_t An[6], Bh[7], R[6];
			#pragma unroll
			for (int dd = 0; dd < 6; ++dd) { An[dd] = __builtin_amdgcn_alignbit(A[dd + 1], A[dd], sha); }
			#pragma unroll
			for (int dd = 0; dd < 6; ++dd) { Bh[dd] = __builtin_amdgcn_alignbit(B[dd + 1], B[dd], shb); }
			Bh[6] = 0u;
			{
				uint32_t X1[7], X2[7];
				const bool q1 = qB & 1u, q2 = qB & 2u, q4 = qB & 4u;
				#pragma unroll
				for (int dd = 0; dd < 6; ++dd) { X1[dd] = q1 ? Bh[dd + 1] : Bh[dd]; }
				X1[6] = 0u;
				#pragma unroll
				for (int dd = 0; dd < 6; ++dd) { X2[dd] = q2 ? (dd + 2 < 7 ? X1[dd + 2 < 7 ? dd + 2 : 6] : 0u) : X1[dd]; }
				X2[6] = 0u;
				// the merged list: halfword t >= 11 - nin from this chunk, below from the previous one
				#pragma unroll
				for (int dd = 0; dd < 6; ++dd) {
					const uint32_t bs = q4 ? (dd + 4 < 7 ? X2[dd + 4 < 7 ? dd + 4 : 6] : 0u) : X2[dd];
					int32_t s = (int32_t)((11u - nin) << 4) - 32 * dd;                  // bits of this dword that belong to the previous chunk's part
					s = s < 0 ? 0 : (s > 32 ? 32 : s);
					const uint32_t m = (uint32_t)(((u64)0xFFFFFFFFu << s));             // low 32 bits: ones where this chunk's entries lie
					R[dd] = (An[dd] & m) | (bs & ~m);
				}
			}
			const uint32_t nc = nin + npv;
			const int32_t o_in = (int32_t)o, o_pv = (int32_t)o + 65536;
			const int32_t c_in = crel, c_pv = crel - 65536;
			// ---- the candidates, nearest first: entry 10 - j of the merged list
			// (written out eleven times through a macro: with the early exit in it the optimizer leaves the loop rolled, and R[] would live in scratch)
			bool go = true;
#define XS_STEP(j) if (go) { \
				constexpr uint32_t hw = 10u - (j); \
				const uint32_t x = (hw & 1u) ? (R[hw >> 1] >> 16) : (R[hw >> 1] & 0xFFFFu); \
				const bool inch = (j) < nin; \
				const int32_t negd = (int32_t)x - (inch ? o_in : o_pv);                /* -(distance) */ \
				const bool alive = (j) < nc && negd >= nmax && key < K48;              /* (the distances grow: once one is too far all the later ones are) */ \
				if (!__builtin_amdgcn_ballot_w64(alive)) { go = false; } \
				else if (alive) { \
					const uint32_t xr = (uint32_t)((int32_t)x + (inch ? c_in : c_pv));      /* window-relative */ \
					uint4 c = xs_ld128(s_data, xr); \
					uint32_t lb = xs_diff_bits16(c.x ^ oa.x, c.y ^ oa.y, c.z ^ oa.z, c.w ^ oa.w, capb); \
					if (lb >= 128u && cap > 16u) { \
						c = xs_ld128(s_data, xr + 16u); \
						uint32_t l = 16u + first_nz_byte16(c.x ^ ob.x, c.y ^ ob.y, c.z ^ ob.z, c.w ^ ob.w); \
						if (l == 32u && cap > 32u) { \
							c = xs_ld128(s_data, xr + 32u); \
							l = 32u + first_nz_byte16(c.x ^ oc.x, c.y ^ oc.y, c.z ^ oc.z, c.w ^ oc.w); \
						} \
						lb = (l < cap ? l : cap) << 3; \
					} \
					const int32_t kc = (int32_t)((lb & ~7u) << 13) + negd;               /* (len << 16) - distance: longest, then nearest */ \
					key = kc > key ? kc : key; \
				} }
			XS_STEP(0u) XS_STEP(1u) XS_STEP(2u) XS_STEP(3u) XS_STEP(4u) XS_STEP(5u) XS_STEP(6u) XS_STEP(7u) XS_STEP(8u) XS_STEP(9u) XS_STEP(10u)
#undef XS_STEP
		}
		const uint32_t best = (uint32_t)(key + 65535) >> 16, boff = (best << 16) - (uint32_t)key;
		wd[o] = best >= 3u ? (best - 3u) | (boff << 16) : 0u;
	}
}

void launch_xp_sort(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint16_t* sorted, uint32_t* words, uint32_t* starts)
{
	if (bt.n_chunks == 0) { return; }
	static PerDeviceOnce attr;
	if (attr.needed()) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(xp_sort_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XS_LDS_BYTES);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(xp_sort_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XS_LDS_BYTES);
		attr.done();
	}
	if (serial_atomics_on_current_device()) { hipLaunchKernelGGL(xp_sort_kernel<true>, dim3(bt.n_chunks), dim3(1024), XS_LDS_BYTES, st, d_in, bt, sorted, words, starts); }
	else { hipLaunchKernelGGL(xp_sort_kernel<false>, dim3(bt.n_chunks), dim3(1024), XS_LDS_BYTES, st, d_in, bt, sorted, words, starts); }
}
void launch_xp_find2(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* sorted, const uint32_t* starts, uint32_t* words, uint32_t max_off, int clip)
{
	if (bt.n_chunks == 0) { return; }
	constexpr uint32_t TXH = 8192u;
	const uint32_t lds = 0x10000u + TXH + 64u;
	static PerDeviceOnce attr;
	if (attr.needed()) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(xp_find2_kernel<1024u, TXH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		attr.done();
	}
	hipLaunchKernelGGL((xp_find2_kernel<1024u, TXH>), dim3(bt.n_chunks * (65536u / TXH)), dim3(1024), lds, st, d_in, bt, sorted, starts, words, max_off, clip);
}

} // namespace msc

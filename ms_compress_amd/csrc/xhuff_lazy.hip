// xhuff_lazy.hip -- the Xpress+Huffman match finder run LAZILY (round 5; a dev / measurement mode: MSCOMP_AMD_XH_LAZY=1):
// Find only where a greedy parse can start a token, with the CHAIN LINKS of the chunk in LDS and the DATA gathered from L2.
//
// Replaces XpressDictionary<0xFFFF, 0x10000>::Find / GetMatchLength (/root/reference/include/mscomp/XpressDictionary.h:145-183, :72-94) as called
// by xh_compress_lz77 (/root/reference/src/xpress_huff_compress.cpp:90-93). xp_find_kernel evaluates EVERY position (9-10 chain steps each); the
// reference only the 0.04-0.38 token starts per byte of its greedy parse (tools/dev/xh_far_study.c: 4.4-5.3 x fewer position / candidate pairs).
//
// The 64 K-position window of this codec (64 KiB of data + 128 KiB of links) does not fit the CU's 160 KiB of LDS; rounds 3 and 4 kept the DATA
// there and took the links from L2 -- a chain of up to 11 dependent L2 round trips per token: 99 ms against 65. Here it is the other way round
// (VERDICT r04 item 2): one 1 024-lane block per 64 KiB chunk holds
//   * the chunk's links as 16-bit DISTANCES to the predecessor (0 = none), the chain into the previous chunk already resolved while staging
//     (the previous chunk's last position with the hash, from its exported head table) ........................................ 128 KiB
//   * the same for the last XHZ_PT positions of the previous chunk ............................................................. 20 KiB
//   * a claim bit per position ................................................................................................. 8 KiB
// so that a token walks its <= 11 links in LDS WITHOUT comparing anything and then asks for the first 16 bytes of all its candidates (and its
// own) from L2 together: one exposed round trip per token. A chain that runs further back in the previous chunk than XHZ_PT positions goes on
// through L2 (the previous chunk's link array), one candidate per wave step, as a state of its own (XHZ_FARL / XHZ_FARD) so that the other lanes
// do not wait for it. Claimed walks as in xpress_lazy.hip: every lane starts at its 64-byte segment, claims the position, runs Find, steps to
// the end of the token (clipped to the chunk, xpress_huff_compress.cpp:93) ... until it meets a claimed position. What is stored for a position
// is what xp_find_kernel stores (length capped at 48 | offset << 16); the words of unvisited positions are 0 and xh_parse_kernel, which only
// looks at positions on its own path, finds every true token start visited (position 0 is claimed; a claimed true token start is followed by its
// successor).
#include "common.h"
#include "kernels.h"
#include <cstdlib>

namespace msc {

#ifndef XHZ_PT
#define XHZ_PT 10240u                                         // positions of the previous chunk whose links are in LDS too
#endif
#ifndef XHZ_JOB
#define XHZ_JOB 32u                                           // bytes per job (a segment whose first position a lane tries to claim)
#endif
#ifndef XHZ_TRIES
#define XHZ_TRIES 4u                                          // jobs / claims a lane may try per wave step
#endif
#define XHZ_CACHE 4u
#define XHZ_LDS (2u * (65536u + XHZ_PT) + 8192u + XHZ_CACHE * 8u + 16u)

enum { XHZ_IDLE = 0, XHZ_NEW = 1, XHZ_FARL = 2, XHZ_FARD = 3, XHZ_FIN = 4, XHZ_WORK = 5 };

__device__ __forceinline__ uint32_t xhz_hash3(uint32_t w) { return (((w & 0x1Fu) << 10) ^ (((w >> 8) & 0xFFu) << 5) ^ ((w >> 16) & 0xFFu)) & 0x7FFFu; }
__device__ __forceinline__ uint32_t xhz_ldg32(const uint8_t* __restrict__ d, u64 pos, u64 n)
{
	if (pos + 4u <= n) { return ld32(d + pos); }
	uint32_t v = 0;
	for (uint32_t k = 0; k < 4u && pos + k < n; ++k) { v |= (uint32_t)d[pos + k] << (8u * k); }
	return v;
}
// 16 bytes at any byte address of the unit; bytes at or beyond n read as 0 (only the unit's last 15 bytes take the slow path: a real call, so
// that its byte loops are not inlined at every one of the ~30 load sites)
__device__ __attribute__((noinline)) uint4 xhz_ldg128_tail(const uint8_t* __restrict__ d, u64 pos, u64 n)
{
	uint4 v;
	v.x = xhz_ldg32(d, pos, n); v.y = xhz_ldg32(d, pos + 4u, n); v.z = xhz_ldg32(d, pos + 8u, n); v.w = xhz_ldg32(d, pos + 12u, n);
	return v;
}
__device__ __forceinline__ uint4 xhz_ldg128(const uint8_t* __restrict__ d, u64 pos, u64 n)
{
	uint4 v;
	if (pos + 16u <= n) { __builtin_memcpy(&v, d + pos, 16); return v; }
	return xhz_ldg128_tail(d, pos, n);
}
__device__ __forceinline__ uint32_t xhz_diff16(const uint4 a, const uint4 b) { return first_nz_byte16(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }

#ifdef XHZ_PROFILE
__device__ unsigned long long g_xhz_prof[8];   // [0] wave steps, [1] claims, [2] near candidates, [3] far candidates, [4] cycles staging, [5] cycles walking, [6] far-link lanes x steps, [7] extension passes
extern "C" void mscomp_amd_debug_xhz_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xhz_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xhz_prof), z, 64); }
#define XHZ_CNT(i, v) xhz_acc[i] += (v);
#else
#define XHZ_CNT(i, v)
#endif

__global__ __launch_bounds__(1024) void xh_lazy_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, const uint16_t* __restrict__ links,
                                                      const uint16_t* __restrict__ lasthead, S16 mlen3)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint16_t* const s_dist = reinterpret_cast<uint16_t*>(smem);                              // [XHZ_PT + 65536]: index = XHZ_PT + offset in the chunk
	uint32_t* const s_bits = reinterpret_cast<uint32_t*>(smem + 2u * (65536u + XHZ_PT));     // [2048]
	u64* const s_cache = reinterpret_cast<u64*>(s_bits + 2048);                              // [XHZ_CACHE]
	uint32_t* const s_cnt = reinterpret_cast<uint32_t*>(s_cache + XHZ_CACHE);                 // [0] cache cursor, [1] next job

	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t lc = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	const uint32_t k = lc - bt.chunk_prefix[u];
	const u64 n = bt.in_len[u];
	const u64 cbase = (u64)k * 65536u;
	const uint32_t cn = (n - cbase < 65536u) ? (uint32_t)(n - cbase) : 65536u;            // positions in this chunk
	const uint32_t ins = (n >= cbase + 3u) ? ((n - 2u - cbase < cn) ? (uint32_t)(n - 2u - cbase) : cn) : 0u;   // positions with a hash: P + 2 < n
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	const uint16_t* __restrict__ lk = links + (u64)lc * 65536u;
	const uint16_t* __restrict__ lkp = lk - 65536;                                        // the previous chunk's links (k > 0 only)
	const uint16_t* __restrict__ lh_prev = lasthead + (u64)(lc - (k ? 1u : 0u)) * 32768u;
	const uint32_t prev_last_hash = (k > 0) ? xhz_hash3(xhz_ldg32(d, cbase - 1u, n)) : 0xFFFFFFFFu;   // 65535 as the previous chunk's head looks like "none": decided by that position's hash (xpress_match.hip)
	uint32_t* __restrict__ words = reinterpret_cast<uint32_t*>(mlen3.p) + (u64)lc * 65536u;           // one word per position: length - 3 | offset << 16

#ifdef XHZ_PROFILE
	unsigned long long xhz_acc[8] = {0}, xhz_t0 = __builtin_readcyclecounter();
#endif
	// ---- stage: links -> distances in LDS (the chain into the previous chunk resolved here), claim bits and result words cleared ----
	if (tid < XHZ_CACHE) { s_cache[tid] = 0ull; }
	if (tid == 0) { s_cnt[0] = 0u; s_cnt[1] = 0u; }
	for (uint32_t i = tid; i < 2048u; i += 1024u) { s_bits[i] = 0u; }
	for (uint32_t i = tid; i * 4u < cn; i += 1024u) { reinterpret_cast<uint4*>(words)[i] = make_uint4(0u, 0u, 0u, 0u); }      // (cn is a multiple of 4 except in a unit's last chunk: the 1-3 words behind it belong to no position)
	for (uint32_t r = tid * 8u; r < cn; r += 8192u) {                                        // the chunk arrays start at multiples of 128 KiB: 8 links per load
		const uint4 v = *reinterpret_cast<const uint4*>(lk + r);
		const uint32_t w8[4] = { v.x, v.y, v.z, v.w };
		// positions without a predecessor inside the chunk continue at the previous chunk's last position with their hash: the (up to 8) data words
		// are asked for together, then the (up to 8) head-table entries (iteration 1 did them one after the other: 2 dependent round trips each)
		uint32_t x[8], hw[8], xp[8];
		bool need[8];
		#pragma unroll
		for (uint32_t e = 0; e < 8u; ++e) { x[e] = (w8[e >> 1] >> ((e & 1u) * 16u)) & 0xFFFFu; need[e] = k != 0u && x[e] == 0xFFFFu && r + e < ins; }
		#pragma unroll
		for (uint32_t e = 0; e < 8u; ++e) { hw[e] = need[e] ? xhz_ldg32(d, cbase + r + e, n) : 0u; }
		#pragma unroll
		for (uint32_t e = 0; e < 8u; ++e) { hw[e] = xhz_hash3(hw[e]); xp[e] = need[e] ? (uint32_t)lh_prev[hw[e]] : 0xFFFFu; }
		uint32_t out[4];
		#pragma unroll
		for (uint32_t e = 0; e < 8u; ++e) {
			const uint32_t o = r + e;
			uint32_t dist = 0;
			if (o < cn) {
				if (x[e] != 0xFFFFu) { dist = o - x[e]; }
				else if (need[e] && (xp[e] != 0xFFFFu || prev_last_hash == hw[e])) { const uint32_t dd = o + 65536u - xp[e]; dist = dd <= 0xFFFFu ? dd : 0u; }
			}
			out[e >> 1] = (e & 1u) ? (out[e >> 1] | (dist << 16)) : dist;
		}
		*reinterpret_cast<uint4*>(s_dist + XHZ_PT + r) = make_uint4(out[0], out[1], out[2], out[3]);
	}
	for (uint32_t t = tid; t < XHZ_PT; t += 1024u) {                                         // the previous chunk's tail (a link leaving THAT chunk leaves my window too)
		uint32_t dist = 0;
		if (k != 0u) { const uint32_t op = 65536u - XHZ_PT + t; const uint32_t x = lkp[op]; if (x != 0xFFFFu) { dist = op - x; } }
		s_dist[t] = (uint16_t)dist;
	}
	__syncthreads();
#ifdef XHZ_PROFILE
	{ const unsigned long long t_ = __builtin_readcyclecounter(); xhz_acc[4] += t_ - xhz_t0; xhz_t0 = t_; }
#endif

	// ---- the walks ------------------------------------------------------------------------------------------------------------------
	// Jobs: the chunk's XHZ_JOB-byte segments, handed out in order by a counter in LDS to the lanes that have nothing to do (iteration 1 gave every
	// lane ONE 64-byte segment: the wave ran as long as its longest walk at 0.30 lane occupancy). A job whose first position is claimed already costs
	// one LDS atomic; up to XHZ_TRIES of them per step.
	uint32_t st = XHZ_IDLE, q = 0;
	uint32_t p = 0, best = 0, lim = 0, cap = 0, rem = 0;       // the token in work: offset in the chunk, best key (len << 16 | 0xFFFF - dist), n - P - 1, min(lim, 48), cn - p
	uint4 own = make_uint4(0, 0, 0, 0);
	uint32_t fo = 0, ftot = 0, fbud = 0, flink = 0;            // far walk: the previous-chunk offset whose link is awaited / was followed, its distance, candidates left, the link in flight
	uint4 fdata = make_uint4(0, 0, 0, 0);                      // far candidate's bytes in flight
	const uint32_t njobs = (cn + XHZ_JOB - 1u) / XHZ_JOB;
	bool jobs_left = true;                                     // (wave-uniform)
	for (;;) {
		// ---- 0. idle lanes take jobs; lanes with a position claim it (XHZ_NEW -> XHZ_WORK, or back to idle when somebody was there first) ---------
		for (uint32_t tries = 0; tries < XHZ_TRIES; ++tries) {
			const u64 im = __ballot(st == XHZ_IDLE);
			if (im && jobs_left) {
				const uint32_t first = ctz64(im);
				uint32_t base = 0;
				if (lane == first) { base = atomicAdd(&s_cnt[1], (uint32_t)__popcll(im)); }
				base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)first);
				if (st == XHZ_IDLE) { const uint32_t job = base + popc_below(im); if (job < njobs) { q = job * XHZ_JOB; st = XHZ_NEW; } }
				jobs_left = base + (uint32_t)__popcll(im) < njobs;
			}
			if (!__ballot(st == XHZ_NEW)) { break; }
			if (st == XHZ_NEW) {
				st = XHZ_IDLE;
				if (q < cn) {
					const uint32_t bit = 1u << (q & 31u);
					const uint32_t old = atomicOr(&s_bits[q >> 5], bit);
					if (!(old & bit)) { p = q; st = XHZ_WORK; }
				}
			}
			if (!(__ballot(st == XHZ_IDLE) && jobs_left)) { break; }
		}
		if (!__ballot(st != XHZ_IDLE)) { if (!jobs_left) { break; } else { continue; } }
		XHZ_CNT(0, lane == 0 ? 1 : 0)
		const u64 P = cbase + p;

		// ---- A. far candidates: what was asked for at the end of the last step has arrived ------------------------------------------
		if (__ballot(st == XHZ_FARL || st == XHZ_FARD)) {
			XHZ_CNT(6, (st == XHZ_FARL || st == XHZ_FARD) ? 1 : 0)
			bool hop = false;                                  // follow `flink` to the next far candidate
			if (st == XHZ_FARD) {                              // a candidate's first 16 bytes: compare
				XHZ_CNT(3, 1)
				uint32_t l = xhz_diff16(fdata, own);
				if (l == 16u && cap > 16u) {                    // (rare for the oldest candidates of a chain: finished here, two more round trips at most)
					l = 16u + xhz_diff16(xhz_ldg128(d, P - ftot + 16u, n), xhz_ldg128(d, P + 16u, n));
					if (l == 32u && cap > 32u) { l = 32u + xhz_diff16(xhz_ldg128(d, P - ftot + 32u, n), xhz_ldg128(d, P + 32u, n)); }
				}
				l = l < cap ? l : cap;
				const uint32_t key = (l << 16) | (0xFFFFu - ftot);
				best = key > best ? key : best;
				--fbud;
				if ((best >> 16) >= 48u || fbud == 0u) { st = XHZ_FIN; } else { hop = true; }
			} else if (st == XHZ_FARL) { hop = true; }
			if (hop) {
				if (flink == 0xFFFFu) { st = XHZ_FIN; }
				else {
					ftot += fo - flink;
					if (ftot > 0xFFFFu) { st = XHZ_FIN; }
					else { fo = flink; fdata = xhz_ldg128(d, P - ftot, n); flink = lkp[fo]; st = XHZ_FARD; }      // the candidate's bytes and ITS link, consumed in the next step
				}
			}
		}

		// ---- B. NEW: claim, own bytes, the chain walked in LDS, all candidates asked for together, compared ---------------------------
		if (__ballot(st == XHZ_WORK)) {
			bool work = false;
			if (st == XHZ_WORK) {
				XHZ_CNT(1, 1)
				rem = cn - p;
				const u64 Pn = cbase + p;
				best = 0;
				if (rem < 3u || Pn + 2u >= n) { st = XHZ_FIN; }                              // :90 / no hash for the unit's last two bytes: a literal
				else { lim = (uint32_t)((n - Pn - 1u) < 0xFFFFFFFFull ? (n - Pn - 1u) : 0xFFFFFFFFull); cap = lim < 48u ? lim : 48u; work = true; }
			}
			if (__ballot(work)) {
				const u64 Pn = cbase + p;
				// the chain, nearest first: cumulative distances cd[0..nc); a candidate whose OWN link is not in LDS ends the walk here (far)
				uint32_t cd[11];
				uint32_t nc = 0, tot = 0;
				int32_t i = (int32_t)(XHZ_PT + p);
				bool alive = work, far = false;
				if (work) { own = xhz_ldg128(d, Pn, n); }                                    // (on its way while the chain is walked)
				uint4 c[11];
				#pragma unroll
				for (uint32_t j = 0; j < 11u; ++j) {
					cd[j] = tot; c[j] = make_uint4(0, 0, 0, 0);
					if (!__ballot(alive)) { continue; }                                     // (uniform: no lane of the wave has a j-th candidate)
					const uint32_t dd = alive ? (uint32_t)s_dist[i] : 0u;
					alive = alive && dd != 0u && tot + dd <= 0xFFFFu;
					tot += alive ? dd : 0u;
					cd[j] = tot;
					nc += alive ? 1u : 0u;
					i -= alive ? (int32_t)dd : 0;
					if (alive) { c[j] = xhz_ldg128(d, Pn - tot, n); }                       // the candidate's first 16 bytes: asked for as soon as its place is known
					if (alive && i < 0) { far = true; alive = false; }                      // this candidate counts; its own link comes from L2
				}
				XHZ_CNT(2, work ? nc : 0)
				// first 16 bytes of every candidate; those that match all 16 are looked at again below
				uint32_t m16 = 0;
				const uint32_t ncmax = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_max(work ? nc : 0u), 63);
				#pragma unroll
				for (uint32_t j = 0; j < 11u; ++j) {
					if (j >= ncmax) { continue; }                                            // (uniform)
					if (work && j < nc) {
						uint32_t l = xhz_diff16(c[j], own);
						if (l == 16u && cap > 16u) { m16 |= 1u << j; }
						else { l = l < cap ? l : cap; const uint32_t key = (l << 16) | (0xFFFFu - cd[j]); best = key > best ? key : best; }
					}
				}
				// bytes 16..47 of the candidates that need them (two more passes at most; the registers of the first 16 bytes are reused)
				for (uint32_t pass = 1; pass <= 2u; ++pass) {
					if (!__ballot(m16 != 0u)) { break; }
					XHZ_CNT(7, lane == 0 ? 1 : 0)
					const uint32_t at = 16u * pass;
					uint4 own2 = make_uint4(0, 0, 0, 0);
					if (m16) { own2 = xhz_ldg128(d, Pn + at, n); }
					#pragma unroll
					for (uint32_t j = 0; j < 11u; ++j) { if ((m16 >> j) & 1u) { c[j] = xhz_ldg128(d, Pn - cd[j] + at, n); } }
					#pragma unroll
					for (uint32_t j = 0; j < 11u; ++j) {
						if ((m16 >> j) & 1u) {
							uint32_t l = at + xhz_diff16(c[j], own2);
							if (l == at + 16u && cap > at + 16u && pass < 2u) { continue; }        // still equal: one more pass
							l = l < cap ? l : cap;
							const uint32_t key = (l << 16) | (0xFFFFu - cd[j]);
							best = key > best ? key : best;
							m16 &= ~(1u << j);
						}
					}
				}
				if (work) {
					if (far && nc < 11u && (best >> 16) < 48u) { fo = (uint32_t)(i + (int32_t)(65536u - XHZ_PT)); ftot = tot; fbud = 11u - nc; st = XHZ_FARL; flink = lkp[fo]; }   // the walk goes on in L2: the far candidate's own link first
					else { st = XHZ_FIN; }
				}
			}
		}

		// ---- C. FIN: store, the token's real end, next position ------------------------------------------------------------------------
		if (__ballot(st == XHZ_FIN)) {
			uint32_t elen = 1u;
			bool coop = false;
			uint32_t xdist = 0;
			if (st == XHZ_FIN) {
				const uint32_t bl = best >> 16;
				if (bl >= 3u) {
					xdist = 0xFFFFu - (best & 0xFFFFu);
					words[p] = (bl - 3u) | (xdist << 16);
					elen = bl < rem ? bl : rem;                                             // :93
					coop = bl >= 48u && rem > 48u && lim > 48u;                            // capped: how long is it really (the walk needs the token's end)
				}
			}
			u64 cm = __ballot(coop);
			while (cm) {
				const uint32_t l0 = ctz64(cm);
				cm &= cm - 1u;
				const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)l0), dist0 = (uint32_t)__builtin_amdgcn_readlane((int)xdist, (int)l0);
				const uint32_t mx0 = (uint32_t)__builtin_amdgcn_readlane((int)lim, (int)l0), rem0 = (uint32_t)__builtin_amdgcn_readlane((int)rem, (int)l0);
				const uint32_t mx = mx0 < rem0 ? mx0 : rem0;                                 // nothing beyond the chunk matters to the walk
				// inside a long run every position finds the same source at the same distance: the end of a long match is kept (start | distance << 16 |
				// length << 32) and reused by every position inside it (xpress_lazy.hip)
				uint32_t res = 0xFFFFFFFFu;
				for (uint32_t ci = 0; ci < XHZ_CACHE; ++ci) {
					const u64 ce_l = s_cache[ci];                                           // (ONE lane's view, broadcast: see xpress_lazy.hip -- another wave's insertion between the two halves of this read made the hit / miss decision non-uniform)
					const u64 ce = (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ce_l) | ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ce_l >> 32)) << 32);
					const uint32_t cs = (uint32_t)ce & 0xFFFFu, cdd = (uint32_t)(ce >> 16) & 0xFFFFu, cl = (uint32_t)(ce >> 32);
					if (cdd == dist0 && pb >= cs && pb - cs < cl && cl - (pb - cs) >= 48u) { res = cl - (pb - cs); }
				}
				if (res == 0xFFFFFFFFu) {
					const u64 pbu = cbase + pb, pau = pbu - dist0;
					uint32_t done = 48u;
					res = mx;
					while (done < mx) {
						const uint32_t off = done + 4u * lane;
						uint32_t df = 0xFFFFFFFFu;
						if (off < mx) { df = xhz_ldg32(d, pau + off, n) ^ xhz_ldg32(d, pbu + off, n); }
						const u64 mis = __ballot(df != 0u);
						if (mis) {
							const uint32_t l1 = ctz64(mis);
							const uint32_t xl = (uint32_t)__builtin_amdgcn_readlane((int)df, (int)l1);
							const uint32_t r = done + 4u * l1 + ((uint32_t)__builtin_ctz(xl) >> 3);
							res = r < mx ? r : mx;
							break;
						}
						done += 256u;
					}
					if (lane == 0) {
						const uint32_t slot = atomicAdd(&s_cnt[0], 1u) % XHZ_CACHE;
						s_cache[slot] = (u64)pb | ((u64)dist0 << 16) | ((u64)res << 32);
					}
				}
				if (lane == l0) { elen = res < rem ? res : rem; }
			}
			if (st == XHZ_FIN) { q = p + elen; st = q < cn ? (uint32_t)XHZ_NEW : (uint32_t)XHZ_IDLE; }      // (claimed at the top of the next step)
		}
	}
#ifdef XHZ_PROFILE
	{ const unsigned long long t_ = __builtin_readcyclecounter(); xhz_acc[5] += t_ - xhz_t0; }
	for (int i_ = 0; i_ < 4; ++i_) { if (xhz_acc[i_]) { atomicAdd(&g_xhz_prof[i_], xhz_acc[i_]); } }
	if (xhz_acc[6]) { atomicAdd(&g_xhz_prof[6], xhz_acc[6]); } if (xhz_acc[7]) { atomicAdd(&g_xhz_prof[7], xhz_acc[7]); }
	if (tid == 0) { atomicAdd(&g_xhz_prof[4], xhz_acc[4]); atomicAdd(&g_xhz_prof[5], xhz_acc[5]); }
#endif
}

void launch_xh_lazy(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* links, const uint16_t* lasthead, uint16_t* mlen3)
{
	if (bt.n_chunks == 0) { return; }
	static PerDeviceOnce attr;
	if (attr.needed()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xh_lazy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XHZ_LDS); attr.done(); }
	hipLaunchKernelGGL(xh_lazy_kernel, dim3(bt.n_chunks), dim3(1024), XHZ_LDS, st, d_in, bt, links, lasthead, S16(mlen3));
}

} // namespace msc

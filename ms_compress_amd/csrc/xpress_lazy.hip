// xpress_lazy.hip -- the Xpress match finder run LAZILY, as a per-lane state machine with the window's data AND chain links in LDS.
//
// Replaces XpressDictionary<0x2000>::Find / GetMatchLength (/root/reference/include/mscomp/XpressDictionary.h:145-183, :72-94) as
// called by xpress_compress (/root/reference/src/xpress_compress.cpp:269-271) for units of at most 64 KiB (one link chunk; longer
// streams keep xp_find_kernel). The reference runs Find only where its greedy parse starts a token; xp_find_kernel evaluates EVERY
// position: 4.5-39 x the candidate compares on the bench corpus (tools/dev/xp_chain_study.c: 0.06-0.46 tokens per byte, and token
// starts have fewer candidates than the positions inside matches).
//
// The greedy walk is serial but it forgets: started at any position as if a token began there it meets the true parse within a few
// tokens (Find(p) depends on p alone). The parse kernel behind this one (xpress_emit*) walks the per-position (len-3, offset) arrays
// itself and only looks at positions on its path, so it is enough to fill them for a SUPERSET of the true token starts:
//   * one block per unit, tiles of XZ_TILE positions one after the other; the tile's bytes and links and those of the 8 KiB window in
//     front of it sit in LDS (the window part is moved down from the tile before, not fetched again);
//   * every lane starts a walk at its segment of XZ_SEG bytes: CLAIM the position (test-and-set of a bit in LDS), Find, step to the
//     end of the token, claim that ... until it reaches a position somebody else has claimed -- whoever claimed it goes on from there.
//     No position is evaluated twice, and the union of the walks contains the true parse: position 0 is claimed, and a claimed true
//     token start is followed by its successor (by the same lane, or by the lane that claimed the successor first);
//   * a walk that leaves the tile is parked in a small list in LDS and goes on when the next tile is staged;
//   * the ONE-lazy-Fill-per-token rule (xpress_compress.cpp:269) makes up to 7 positions behind a match longer than 8 KiB forced
//     literals and resumes the parse behind them: the 8 possible resume points behind every such match are parked as walks too.
// Every lane is its own state machine -- NEW (claim, own 16 bytes, first link) -> FIND (16-byte compares of chain candidates, up to
// XZ_FREP per step; the next candidate's bytes and link are asked for before the current one is compared) -> DONE (store, extend a capped
// match, next position) -- and the wave runs the blocks once per step, so lanes on short chains do not wait for lanes on long ones
// (the all-positions walk ran at 0.45 lane occupancy).
// What is stored for a position is what xp_find_kernel stores (length capped at 48, offset) -- or, for a capped match this kernel extended for its own
// walk, the match's full length (round 6); offsets of unvisited positions are 0.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

namespace msc {

#define XZ_WIN   0x2000u                  // Xpress window (MaxOffset)
#define XZ_PAD   144u                     // bytes staged behind the tile: a lane extends a capped match up to 112 + 16 bytes itself
#ifndef XZ_LIST
#define XZ_LIST  512u                     // parked walks (positions) of a whole unit: slots are handed out by a counter that only grows, slot i belongs to lane i
#endif
#define XZ_CACHE 4u                       // ends of long matches kept for the other positions inside them
#ifndef XZ_FREP
#define XZ_FREP 4u                        // chain candidates a lane may finish per step (the DONE / NEW blocks run once per step). Even: the two candidate register
                                          // sets of the compare loop are back in place after a step. configs[4], round 5: 2: 30.9 ms, 3: 31.0, 4: 28.0, 6: 29.3, 8: 32.3
#endif
#define XZ_OWN_EXT 112u                   // a lane compares up to here itself, 16 bytes a step; beyond, the wave compares 256 bytes a step from global memory

enum { XZ_IDLE = 0, XZ_NEW = 1, XZ_FIND = 2, XZ_FIND2 = 3, XZ_DONE = 4, XZ_EXT = 5, XZ_DONE2 = 6, XZ_COOP = 7 };

__device__ __forceinline__ uint32_t xz_hash3(uint32_t w) { return (((w & 0x1Fu) << 10) ^ (((w >> 8) & 0xFFu) << 5) ^ ((w >> 16) & 0xFFu)) & 0x7FFFu; }
__device__ __forceinline__ uint32_t xz_diff16(const uint4 a, const uint4 b) { return first_nz_byte16(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
__device__ __forceinline__ uint32_t xz_ldg32(const uint8_t* __restrict__ d, u64 pos, u64 n)
{
	if (pos + 4u <= n) { return ld32(d + pos); }
	uint32_t v = 0;
	for (uint32_t k = 0; k < 4u && pos + k < n; ++k) { v |= (uint32_t)d[pos + k] << (8u * k); }
	return v;
}

#ifdef XZ_PROFILE
__device__ unsigned long long g_xz_prof[8];
extern "C" void mscomp_amd_debug_xz_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xz_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xz_prof), z, 64); }
#define XZ_CNT(i, v) xz_acc[i] += (v);
#define XZ_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); xz_acc[4 + (i)] += t_ - xz_prev; xz_prev = t_; }   /* wave cycles: [4] staging, [5] walks, [6] waiting for the block */
#else
#define XZ_CNT(i, v)
#define XZ_T(i)
#endif

// LDS: data [XZ_WIN + TILE + XZ_PAD] | links u16 [XZ_WIN + TILE] | claim bits [TILE / 32] | list [XZ_LIST] | counters [4] | long-match cache [XZ_CACHE] u64
template <uint32_t TILE, uint32_t SEG, uint32_t WPE>   // WPE: waves per SIMD the register budget allows (HIP's second launch bound)
__global__ __launch_bounds__(TILE / SEG, WPE) void xp_lazy2_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, const uint16_t* __restrict__ links,
                                                             S16 mlen3, S16 moff)
{
	constexpr uint32_t NT = TILE / SEG;
	static_assert(NT >= XZ_WIN / 16u, "the window is moved down by 512 lanes");
	constexpr uint32_t DATA_BYTES = XZ_WIN + TILE + XZ_PAD;
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint8_t* const s_data = smem;
	uint16_t* const s_links = reinterpret_cast<uint16_t*>(smem + DATA_BYTES);
	uint32_t* const s_bits = reinterpret_cast<uint32_t*>(smem + DATA_BYTES + (XZ_WIN + TILE) * 2u);
	uint32_t* const s_list = s_bits + TILE / 32u;
	uint32_t* const s_cnt = s_list + XZ_LIST;              // [0] entries in the list, [1] list overflow -> every position of the tile is evaluated, [2] cache cursor
	u64* const s_cache = reinterpret_cast<u64*>(s_cnt + 4);

	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t lc = blockIdx.x;                          // units of at most 64 KiB: unit == link chunk
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	const u64 n64 = bt.in_len[u];
	const uint32_t cn = (uint32_t)n64;                       // <= 65536
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	const uint16_t* __restrict__ lk = links + (u64)lc * 65536u;
	S16 ml = mlen3 + (u64)lc * 65536u;
	(void)moff;                                              // (the offset is the upper half of the word `ml` points at)

	if (tid < 4u) { s_cnt[tid] = 0u; }
	if (tid < XZ_CACHE) { s_cache[tid] = 0ull; }
	for (uint32_t i = tid; i < XZ_LIST; i += NT) { s_list[i] = 0xFFFFFFFFu; }
#ifdef XZ_PROFILE
	unsigned long long xz_acc[8] = {0}, xz_prev = __builtin_readcyclecounter();
#endif

	for (uint32_t tb = 0; tb < cn; tb += TILE) {             // tile = chunk positions [tb, te)
		const uint32_t te = (tb + TILE < cn) ? tb + TILE : cn;
		__syncthreads();
		// ---- stage: the window part comes from the tile before (LDS -> LDS), the tile from global memory; LDS index = position - tb + XZ_WIN
		if (tb) {
			uint4 keep_d = make_uint4(0, 0, 0, 0), keep_a = keep_d, keep_b = keep_d;
			const bool mv = tid * 16u < XZ_WIN;                  // 512 lanes move 16 bytes of data and 16 links each
			uint8_t* const lb = reinterpret_cast<uint8_t*>(s_links);
			if (mv) {
				keep_d = *reinterpret_cast<const uint4*>(s_data + TILE + tid * 16u);
				keep_a = *reinterpret_cast<const uint4*>(lb + (TILE + tid * 16u) * 2u);
				keep_b = *reinterpret_cast<const uint4*>(lb + (TILE + tid * 16u + 8u) * 2u);
			}
			__syncthreads();
			if (mv) {
				*reinterpret_cast<uint4*>(s_data + tid * 16u) = keep_d;
				*reinterpret_cast<uint4*>(lb + (tid * 16u) * 2u) = keep_a;
				*reinterpret_cast<uint4*>(lb + (tid * 16u + 8u) * 2u) = keep_b;
			}
		}
		{
			const uint32_t dend = (tb + TILE + XZ_PAD < cn) ? tb + TILE + XZ_PAD : cn;      // bytes [tb, dend) of the unit, zero behind
			const uint32_t dlen = dend - tb;
			const uint8_t* __restrict__ src = d + tb;
			const uint32_t nvec = (((uintptr_t)src & 15u) == 0) ? (dlen & ~15u) : 0u;
			for (uint32_t i = tid * 16u; i < nvec; i += NT * 16u) { *reinterpret_cast<uint4*>(s_data + XZ_WIN + i) = *reinterpret_cast<const uint4*>(src + i); }
			for (uint32_t i = nvec + tid; i < dlen; i += NT) { s_data[XZ_WIN + i] = src[i]; }
			for (uint32_t i = dlen + tid; i < TILE + XZ_PAD; i += NT) { s_data[XZ_WIN + i] = 0; }
			const uint32_t ln = te - tb;                                                    // links of the tile (chunk arrays start at multiples of 128 KiB)
			for (uint32_t r = tid * 8u; r < ln; r += NT * 8u) { *reinterpret_cast<uint4*>(s_links + XZ_WIN + r) = *reinterpret_cast<const uint4*>(lk + tb + r); }
			for (uint32_t i = tid; i < TILE / 32u; i += NT) { s_bits[i] = 0u; }
			uint4* __restrict__ moz = reinterpret_cast<uint4*>(ml.p + 2u * tb);                // words (length | offset << 16) of unvisited positions read as "no match"
			for (uint32_t i = tid; i * 4u < ln; i += NT) { moz[i] = make_uint4(0u, 0u, 0u, 0u); }
		}
		__syncthreads();

		// ---- rounds: round 0 = my segment + my parked walk; later rounds = walks parked while the tile was worked on (resume points)
		XZ_T(0)
		bool swept = false;
		for (uint32_t round = 0;; ++round) {
			const bool do_sweep = s_cnt[1] != 0u && !swept;      // the list overflowed (never seen): every position of this and all later tiles is evaluated
			swept = swept || do_sweep;
			uint32_t job0 = 0xFFFFFFFFu, job1 = 0xFFFFFFFFu;
			if (round == 0) { const uint32_t q0 = tb + tid * SEG; if (q0 < te) { job0 = q0; } }
			for (uint32_t slot = tid; slot < XZ_LIST; slot += NT) {          // slot i of the list belongs to lane i mod NT; one parked walk per lane and round
				const uint32_t e = s_list[slot];
				if (job1 == 0xFFFFFFFFu && e != 0xFFFFFFFFu && e < te) { job1 = e; s_list[slot] = 0xFFFFFFFFu; }
			}
			uint32_t sweep = do_sweep ? tb + tid * SEG : 0xFFFFFFFFu;        // next position of my segment in all-positions mode
			const uint32_t sweep_end = (tb + tid * SEG + SEG < te) ? tb + tid * SEG + SEG : te;

			// ---- the state machine. One step = FIND (a compare), then DONE (store / extend / next position), then NEW (claim + start).
			// A walk's FIRST candidate (16 bytes and link) is fetched at the END of the step that claimed it (`cand`, `nlk`), so that the LDS round trip
			// overlaps the other blocks of the next step.
			uint32_t st = XZ_IDLE, p = 0, q = 0, best = 0, x = 0, dl = 0, cap = 0, lim = 0, chain = 0, elen = 0, nlk = 0;
			uint4 own = make_uint4(0, 0, 0, 0), cand = make_uint4(0, 0, 0, 0), candB = make_uint4(0, 0, 0, 0);
			uint32_t nlkB = 0, nkeep = 0;
			bool single = false, fetch = false;
			for (;;) {
				if (st == XZ_IDLE) {                               // idle lanes take their next job
					if (job0 != 0xFFFFFFFFu) { q = job0; job0 = 0xFFFFFFFFu; st = XZ_NEW; single = false; }
					else if (job1 != 0xFFFFFFFFu) { q = job1; job1 = 0xFFFFFFFFu; st = XZ_NEW; single = false; }
					else if (sweep < sweep_end) { q = sweep++; st = XZ_NEW; single = true; }
				}
				if (!__ballot(st != XZ_IDLE)) { break; }
				XZ_CNT(0, 1)
				// ---- FIND: one 16-byte compare of one chain candidate (its first 16 bytes against my own in registers)
				// (written with selects, not nested branches: the compiler's exec-mask bookkeeping for the nested form was as long as the
				// arithmetic; the only branch left guards the loads of the next candidate)
				// Two register sets for "the candidate being compared" and "the one after it" that swap roles from one compare to the next (round 5:
				// the select form `if (more) { cand = cand2; nlk = nlk2; }` cost ten v_mov per compare, and this kernel runs at 0.88 of the CU's
				// vector issue rate). Every lane in FIND takes part in every compare of a step, so all of them hold their live candidate in the same
				// set; a lane that leaves for FIND2 takes its candidate's link along in `nkeep`. (No test for "no link": 0xFFFF fails the distance
				// test, p <= 65533 in FIND.)
#define XZ_COMPARE(CA, NA, CB, NB) { \
					const bool f = st == XZ_FIND; \
					if (!__ballot(f)) { break; } \
					XZ_CNT(2, f ? 1 : 0) \
					/* the candidate AFTER this one (its place is known: my candidate's link) is asked for before this one is compared, */ \
					/* so its LDS round trip runs under the compare */ \
					const bool pf = f && chain > 1u && (p - NA) <= XZ_WIN; \
					if (pf) { \
						const uint32_t xw = NA - tb + XZ_WIN; \
						CB = lds_ld128(s_data, xw); \
						NB = s_links[xw]; \
					} \
					const uint32_t l = xz_diff16(CA, own); \
					const bool to2 = f && l == 16u && cap > 16u; \
					const bool upd = f && !to2; \
					const uint32_t lc2 = l < cap ? l : cap; \
					const uint32_t key = (lc2 << 16) | (0xFFFFu - (p - x)); \
					best = (upd && key > best) ? key : best; \
					chain -= upd ? 1u : 0u; \
					const bool more = upd && pf && (best >> 16) < 48u; \
					x = upd ? NA : x; \
					nkeep = to2 ? NA : nkeep; \
					dl = to2 ? 16u : dl; \
					st = to2 ? (uint32_t)XZ_FIND2 : (upd ? (more ? (uint32_t)XZ_FIND : (uint32_t)XZ_DONE) : st); \
					if (upd) { fetch = false; } }
				#pragma unroll 1
				for (uint32_t rep = 0; rep < XZ_FREP; rep += 2u) {
					XZ_COMPARE(cand, nlk, candB, nlkB)
					if (rep + 1u < XZ_FREP) { XZ_COMPARE(candB, nlkB, cand, nlk) }
					else if (st == XZ_FIND) { cand = candB; nlk = nlkB; }       // (an odd number of compares per step: back to the first set)
				}
#undef XZ_COMPARE
				// ---- FIND2 / EXT: 16 more bytes of the same candidate (both sides from LDS)
				if (__ballot(st == XZ_FIND2 || st == XZ_EXT)) {
					if (st == XZ_FIND2 || st == XZ_EXT) {
						const uint32_t xw = x - tb + XZ_WIN, pw = p - tb + XZ_WIN;
						const uint32_t la = xz_diff16(lds_ld128(s_data, xw + dl), lds_ld128(s_data, pw + dl));
						uint32_t l = dl + la;
						const uint32_t top = st == XZ_FIND2 ? cap : (lim < XZ_OWN_EXT ? lim : XZ_OWN_EXT);
						if (la == 16u && l < top) { dl = l; }
						else if (st == XZ_FIND2) {
							l = l < cap ? l : cap;
							const uint32_t key = (l << 16) | (0xFFFFu - (p - x));
							best = best > key ? best : key;
							--chain;
							fetch = chain != 0u && (p - nkeep) <= XZ_WIN && (best >> 16) < 48u;
							x = nkeep;
							st = fetch ? XZ_FIND : XZ_DONE;
						} else {
							if (la == 16u && l >= XZ_OWN_EXT && lim > XZ_OWN_EXT) { dl = XZ_OWN_EXT; st = XZ_COOP; }   // still equal at 112: the wave takes over
							else { elen = l < lim ? l : lim; st = XZ_DONE2; *reinterpret_cast<uint32_t*>(&ml[p]) = (elen - 3u) | ((p - x) << 16); }   // the extended match: its real length
						}
					}
				}
				// ---- DONE: store, extend a capped match, step to the next position
				if (__ballot(st == XZ_DONE || st == XZ_DONE2 || st == XZ_COOP)) {
					if (st == XZ_DONE) {
						const uint32_t bl = best >> 16;
						if (bl >= 3u) {
							const uint32_t dist = 0xFFFFu - (best & 0xFFFFu);
							XZ_CNT(3, 1)
							// A capped match is extended here -- the walk needs its end -- and its word leaves with the FULL length (round 6): the parse
							// kernels take any length but 48 as final, so xpress_emit*'s walk no longer stops at every long match for a wave-wide compare
							// from global memory (a microsecond on the serial chain of a unit, each). An all-positions sweep stores the capped length.
							*reinterpret_cast<uint32_t*>(&ml[p]) = (bl - 3u) | (dist << 16);      // one word: length - 3 | offset << 16
							if (bl >= 48u && lim > 48u && !single) { st = XZ_EXT; x = p - dist; dl = 48u; }    // how long is it really (the walk needs its end; the word is stored again with it)
							else { elen = bl; st = XZ_DONE2; }
						} else { elen = 1u; st = XZ_DONE2; }
					}
					u64 cm = __ballot(st == XZ_COOP);                // a match longer than a lane extends itself
					while (cm) {
						const uint32_t l0 = ctz64(cm);
						cm &= cm - 1u;
						const uint32_t pa = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)l0), pb = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)l0);
						const uint32_t from = (uint32_t)__builtin_amdgcn_readlane((int)dl, (int)l0), mx = (uint32_t)__builtin_amdgcn_readlane((int)lim, (int)l0);
						// Inside a long run every segment start finds the same source at the same distance: the end of a long match is kept
						// (start | distance << 16 | length << 32, one 64-bit LDS word per entry) and reused by every position inside it.
						uint32_t res = 0xFFFFFFFFu;
						// ONE lane's view of an entry, broadcast (round 6): a wave's LDS read is served in two halves of 32 lanes, and another wave's
						// insertion can land BETWEEN them -- lanes 0-31 then held the old entry, lanes 32-63 the new one, `res` stopped being
						// wave-uniform and the compare below ran (ballots, readlanes) for half a wave: a match end too far out, its successor never
						// claimed, literals where the reference has a match. Found by real files (64 KiB pieces of periodic GPU code tables: 3-10 %
						// of the units when many long matches of one distance are extended at once); tests/test_gpu_parity.py
						// test_lazy_finder_long_match_cache_under_concurrent_insertions keeps it.
						for (uint32_t c = 0; c < XZ_CACHE; ++c) {
							const u64 ce_l = s_cache[c];
							const u64 ce = (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ce_l) |
							               ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ce_l >> 32)) << 32);
							const uint32_t cs = (uint32_t)ce & 0xFFFFu, cd = (uint32_t)(ce >> 16) & 0xFFFFu, cl = (uint32_t)(ce >> 32);
							if (cd == pb - pa && pb >= cs && pb - cs < cl && cl - (pb - cs) >= from) { res = cl - (pb - cs); }
						}
						if (res == 0xFFFFFFFFu) {                        // the wave compares 256 bytes a step, from global memory (the match may leave the tile)
							uint32_t done = from;
							res = mx;
							while (done < mx) {
								const uint32_t off = done + 4u * lane;
								uint32_t df = 0xFFFFFFFFu;
								if (off < mx) { df = xz_ldg32(d, (u64)pa + off, n64) ^ xz_ldg32(d, (u64)pb + off, n64); }
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
								const uint32_t slot = atomicAdd(&s_cnt[2], 1u) % XZ_CACHE;
								s_cache[slot] = (u64)pb | ((u64)(pb - pa) << 16) | ((u64)res << 32);
							}
						}
						if (lane == l0) { elen = res; st = XZ_DONE2; *reinterpret_cast<uint32_t*>(&ml[p]) = (elen - 3u) | ((p - x) << 16); }
					}
					if (st == XZ_DONE2) {
						const uint32_t e = p + elen;
						if (elen > XZ_WIN && !single) {                 // the 8 possible resume points behind a match longer than 8 KiB
							for (uint32_t j = 1; j <= 8u; ++j) {
								if (e + j < cn) {
									const uint32_t slot = atomicAdd(&s_cnt[0], 1u);
									if (slot < XZ_LIST) { s_list[slot] = e + j; } else { s_cnt[1] = 1u; }
								}
							}
						}
						q = e;
						st = single ? XZ_IDLE : XZ_NEW;
					}
				}
				// ---- NEW: claim q, start its Find. The claim (a returning LDS atomic), the position's own 16 bytes and its first link
				// are asked for TOGETHER and waited for once; a lane whose claim fails has read 24 bytes in vain
				if (__ballot(st == XZ_NEW)) {
					const bool isnew = st == XZ_NEW, intile = isnew && q < te;
					const uint32_t qq = intile ? q : tb;
					const uint32_t qw = qq - tb + XZ_WIN;
					const uint32_t bit = 1u << (qq & 31u);
					uint32_t old = 0xFFFFFFFFu;
					if (intile) { old = atomicOr(&s_bits[(qq - tb) >> 5], bit); }
					const uint4 own_n = lds_ld128(s_data, qw);
					const uint32_t x_n = s_links[qw];
					if (isnew) {
						st = XZ_IDLE;
						if (!intile) {
							if (q < cn) {                                  // leaves the tile: parked for the next one
								const uint32_t slot = atomicAdd(&s_cnt[0], 1u);
								if (slot < XZ_LIST) { s_list[slot] = q; } else { s_cnt[1] = 1u; }
							}
						} else if (!(old & bit)) {
							XZ_CNT(1, 1)
							p = q;
							lim = cn - p - 1u;                             // never count the buffer's final byte (XpressDictionary.h:88-93)
							if (p + 2u >= cn) { elen = 1u; st = XZ_DONE2; }          // the unit's last two bytes are literals (xpress_compress.cpp:266)
							else {
								cap = lim < 48u ? lim : 48u;
								own = own_n;
								x = x_n;
								best = 0u; chain = 11u; dl = 0u;
								fetch = p - x <= XZ_WIN;                          // (x == 0xFFFF, no link: the difference wraps, p <= 65533 here)
								st = fetch ? XZ_FIND : XZ_DONE;
							}
						}
					}
				}
				// ---- the next candidate's bytes and link, on their way while the next step begins
				if (fetch) {
					const uint32_t xw = x - tb + XZ_WIN;
					cand = lds_ld128(s_data, xw);
					nlk = s_links[xw];
					fetch = false;
				}
			}
			XZ_T(1)
			__syncthreads();
			XZ_T(2)
			// another round when walks for THIS tile were parked meanwhile
			bool again = false;
			for (uint32_t slot = tid; slot < XZ_LIST; slot += NT) { const uint32_t e = s_list[slot]; again = again || (e != 0xFFFFFFFFu && e < te); }
			const bool need_sweep = s_cnt[1] != 0u && !swept;                       // overflow noticed during this round: sweep the tile now
			if (!__syncthreads_or((again || need_sweep) ? 1 : 0)) { break; }
		}
	}
#ifdef XZ_PROFILE
	for (int i_ = 0; i_ < 4; ++i_) { if (xz_acc[i_]) { atomicAdd(&g_xz_prof[i_], xz_acc[i_]); } }
	if (lane == 0) { for (int i_ = 4; i_ < 7; ++i_) { atomicAdd(&g_xz_prof[i_], xz_acc[i_]); } }
#endif
}

// units of at most 64 KiB only (the caller checks): one block per unit
template <uint32_t TILE, uint32_t SEG, uint32_t WPE>
static void launch_xz(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* links, uint16_t* mlen3, uint16_t* moff)
{
	const uint32_t lds = XZ_WIN + TILE + XZ_PAD + (XZ_WIN + TILE) * 2u + TILE / 8u + XZ_LIST * 4u + 16u + XZ_CACHE * 8u;
	static PerDeviceOnce attr;
	if (attr.needed()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xp_lazy2_kernel<TILE, SEG, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr.done(); }
	hipLaunchKernelGGL((xp_lazy2_kernel<TILE, SEG, WPE>), dim3(bt.n_chunks), dim3(TILE / SEG), lds, st, d_in, bt, links, mlen3, moff);
}
void launch_xp_lazy2(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* links, uint16_t* mlen3, uint16_t* moff)
{
	if (bt.n_chunks == 0) { return; }
	// shapes measured on BASELINE configs[4] (ms per pass): 16 KiB tiles / 32-byte segments (512 lanes, 2 blocks = 16 waves per CU) 35.3;
	// 16-byte segments with 1024 lanes at 64 registers 37.9; 8 KiB tiles / 16-byte segments (3 blocks = 24 waves) 39.5; / 8-byte segments 55.5
	static const int variant = [] { const char* e = getenv("MSCOMP_AMD_XZ"); return e ? atoi(e) : 0; }();   // dev switch
	switch (variant) {
	case 1:  launch_xz<16384u, 16u, 8u>(st, d_in, bt, links, mlen3, moff); break;
	case 2:  launch_xz<8192u, 16u, 6u>(st, d_in, bt, links, mlen3, moff); break;
	default: launch_xz<16384u, 32u, 4u>(st, d_in, bt, links, mlen3, moff); break;
	}
}

} // namespace msc

// xpress_emit.hip -- Xpress (plain LZ77) greedy parse + stream emission for gfx950, bit-exact with the reference.
//
// Replaces the token loop of xpress_compress (/root/reference/src/xpress_compress.cpp:262-345): greedy selection,
// the ONE-lazy-Fill-per-token rule (:269, "lagging fill": a position at or beyond `filled` is a forced literal),
// the 16-bit match symbol + nibble / byte / u16 / u32 length extensions (:274-315) and the 32-token flag words
// (:317-324, :343-344).
//
// One wavefront per unit (an Xpress stream is sequential: flag words and the shared length nibble couple all tokens).
// Per 64-position window the lanes hold the per-position matches found by xp_find_kernel; the greedy walk runs on
// the scalar unit over the ballot mask of candidates (one step per MATCH; literal runs are skipped with s_ff1); a
// chosen match whose length hit the finder's cap (48) is extended by the whole wave, 256 bytes per step. Every
// output byte position is a prefix sum (SURVEY.md 8a "scan layouts"):
//   pos(t) = 4*(t div 32 + 1) + sum size(u<t),  size = 1 | 2 + [L>=7 and long-rank even] + [L>=22] + [L>=277]*(2|6)
// evaluated with mbcnt popcounts and a DPP add-scan; the flag word in progress is kept in scalar registers.
#include "common.h"
#include "kernels.h"

namespace msc {

__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v) { return wave_incl_scan_add_u32(v); }
__device__ __forceinline__ uint32_t wave_incl_scan_max_u32(uint32_t v) { return wave_incl_scan_max(v); }

__device__ __forceinline__ uint32_t ldg32_lim(const uint8_t* __restrict__ d, u64 pos, u64 n)
{
	if (pos + 4u <= n) { return ld32(d + pos); }
	uint32_t v = 0;
	for (uint32_t k = 0; k < 4u && pos + k < n; ++k) { v |= (uint32_t)d[pos + k] << (8u * k); }
	return v;
}

// Number of equal bytes of d[a..] and d[b..] (a < b), at most maxadd; the whole wave compares 256 bytes per step.
__device__ __forceinline__ u64 wave_extend(const uint8_t* __restrict__ d, u64 a, u64 b, u64 maxadd, u64 n, uint32_t lane)
{
	u64 done = 0;
	while (done < maxadd) {
		const u64 off = done + 4u * lane;
		uint32_t x = 0xFFFFFFFFu;                               // lanes beyond the limit report a mismatch
		if (off < maxadd) { x = ldg32_lim(d, a + off, n) ^ ldg32_lim(d, b + off, n); }
		const u64 mis = __ballot(x != 0);
		if (mis) {
			const uint32_t l = ctz64(mis);
			const uint32_t xl = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)l);
			const u64 r = done + 4u * l + ((uint32_t)__builtin_ctz(xl) >> 3);
			return r < maxadd ? r : maxadd;
		}
		done += 256u;
	}
	return maxadd;
}

#ifdef XE_PROFILE
__device__ unsigned long long g_xe_prof[8];
extern "C" void mscomp_amd_debug_xe_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xe_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xe_prof), z, 64); }
#define XE_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); xe_acc[i] += t_ - xe_prev; xe_prev = t_; }
#else
#define XE_T(i)
#endif
__device__ __forceinline__ void put8(uint8_t* __restrict__ out, u64 cap, u64 pos, uint32_t v) { if (pos < cap) { out[pos] = (uint8_t)v; } }

// one unaligned 32-bit store (or four capacity-checked bytes when the window may cross the capacity)
__device__ __forceinline__ void xe_store32(uint8_t* __restrict__ out, u64 cap, u64 pos, uint32_t v, bool chk)
{
	if (!chk) { st32(out + pos, v); }
	else { put8(out, cap, pos, v); put8(out, cap, pos + 1u, v >> 8); put8(out, cap, pos + 2u, v >> 16); put8(out, cap, pos + 3u, v >> 24); }
}
// The bytes of this lane's token (:274-315). CHK = false: the whole window fits below the capacity (decided once, wave
// uniform), plain stores and the 16-bit symbol in one store; CHK = true: byte stores, each checked against the capacity.
template <bool CHK>
__device__ __forceinline__ void xe_emit_tokens(uint8_t* __restrict__ out, u64 cap, u64 base, uint32_t posrel, bool is_tok, bool is_m,
                                               uint32_t byte, uint32_t off, uint32_t L, bool lng, bool even, bool has_above,
                                               uint32_t nib, uint32_t pnib, bool pend, u64 pend_pos, uint32_t pend_low, u64 longmask, uint32_t lane)
{
#define XE_PUT(q_, v_) { if (CHK) { put8(out, cap, base + (q_), (v_)); } else { ob[(q_)] = (uint8_t)(v_); } }
	uint8_t* __restrict__ ob = out + base;
	if (!is_tok) { return; }
	if (!is_m) { XE_PUT(posrel, byte) return; }
	const uint32_t sym = ((off - 1u) << 3) | (L < 7u ? L : 7u);
	if (CHK) { put8(out, cap, base + posrel, sym); put8(out, cap, base + posrel + 1u, sym >> 8); } else { st16(ob + posrel, sym); }
	uint32_t q = posrel + 2u;
	if (lng) {
		if (even) { XE_PUT(q, nib | (has_above ? pnib << 4 : 0u)) ++q; }
		else if (pend && (longmask & ((((u64)1) << lane) - 1u)) == 0) { if (CHK) { put8(out, cap, pend_pos, pend_low | (nib << 4)); } else { out[pend_pos] = (uint8_t)(pend_low | (nib << 4)); } }
		if (L >= 22u) {
			XE_PUT(q, L - 22u < 255u ? L - 22u : 255u) ++q;
			if (L >= 277u) {
				if (L <= 0xFFFFu) { XE_PUT(q, L) XE_PUT(q + 1u, L >> 8) }
				else { XE_PUT(q, 0) XE_PUT(q + 1u, 0) XE_PUT(q + 2u, L) XE_PUT(q + 3u, L >> 8) XE_PUT(q + 4u, L >> 16) XE_PUT(q + 5u, L >> 24) }
			}
		}
	}
#undef XE_PUT
}

// P = the type of a position in the unit / in its output: uint32_t for units below 2 GiB (every position, the output size and the capacity that
// matters fit 32 bits: the scalar unit, which runs the walk, does one add / compare where the 64-bit form needs two), u64 otherwise. Same code.
#define XE_RING 128u                                           // tokens queued between the walk and the emission (a power of two >= 2 x 64)
template <typename P>
__device__ __forceinline__ void xpress_emit_body(const uint8_t* __restrict__ d_in, const BatchTables& bt,
                                                        S16 mlen3, S16 moff,
                                                        uint8_t* __restrict__ d_out, u64* __restrict__ d_out_len, int32_t* __restrict__ d_status, uint32_t u,
                                                        uint16_t (*s_in_off)[512], uint16_t (*s_in_len)[512], uint8_t (*s_in_byte)[512], uint16_t* s_qa, uint32_t* s_ql)
{
	const uint32_t lane = threadIdx.x;
	const P n = (P)bt.in_len[u];
	const P cap = bt.out_cap[u] > (u64)(P)~(P)0 ? (P)~(P)0 : (P)bt.out_cap[u];     // (a capacity beyond P's range never binds: the stream is shorter)
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	uint8_t* __restrict__ out = d_out + bt.out_off[u];
	const u64 mbase = (u64)bt.chunk_prefix[u] * 65536u;           // this unit's slice of the per-position match arrays
	const P end2 = n >= 2u ? n - 2u : 0u;

	P cur = 0, F = 0, S = 0, N = 0, R = 0;                      // next token start, filled, sum sizes, tokens, long matches
	bool pend = false; P pend_pos = 0; uint32_t pend_low = 0;   // length nibble byte waiting for its high half
	uint32_t facc = 0; P fposc = 0;                             // flag word in progress (first token = bit 0) and its slot

	// Inputs are burst-loaded 8 windows (512 positions) at a time into a double-buffered LDS stage: one wait on global
	// memory per 512 positions instead of one per window (loads are unconditional with a clamped index).
	// (the stage itself is declared by the kernel and handed in: two instances of this body must not each get their own)
	uint32_t g_off[8], g_len[8], g_byte[8];
#define XE_BURST_LOAD(gbase) { _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) { \
		const P q_ = (gbase) + (P)k_ * 64u + lane; const P c_ = q_ < n ? q_ : n - 1u; \
		{ const uint32_t w_ = mlen3.word(mbase + c_); g_off[k_] = w_ >> 16; g_len[k_] = w_ & 0xFFFFu; } g_byte[k_] = d[c_]; } }
#define XE_BURST_STORE(buf) { _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) { \
		s_in_off[buf][k_ * 64 + lane] = (uint16_t)g_off[k_]; s_in_len[buf][k_ * 64 + lane] = (uint16_t)g_len[k_]; s_in_byte[buf][k_ * 64 + lane] = (uint8_t)g_byte[k_]; } }
	if (n) { XE_BURST_LOAD((P)0) }
#ifdef XE_PROFILE
	unsigned long long xe_acc[6] = {0, 0, 0, 0, 0, 0}, xe_prev = __builtin_readcyclecounter();
#endif
	// Round 5: the emission works on 64 TOKENS at a time, not on the 64 positions of a window. A window holds 4-30 tokens (0.06-0.46 per
	// byte), so the size scan, the token stores, the flag words and the carries -- more than half of this kernel's ~150 scalar instructions per
	// window, and the scalar unit (one per CU) is what the kernel is bound by -- ran for a quarter of their lanes. The walk queues every
	// window's tokens in a 128-entry LDS ring in rank order (match: 0x8000 | offset and len - 3; literal: the byte); a step takes 64 of them,
	// lane = token: what follows is the per-window code it was, with tokmask = all lanes below cnt, tb = lane, and the match bits in token
	// order a plain ballot (no ds_permute).
	uint32_t qh = 0, qt = 0;                                     // ring: tokens taken / queued so far
	auto emit_step = [&](const uint32_t cnt) __attribute__((always_inline)) {
		const bool is_tok = lane < cnt;
		const uint32_t qa = __hip_atomic_load(&s_qa[(qh + lane) & (XE_RING - 1u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		const uint32_t L = is_tok ? __hip_atomic_load(&s_ql[(qh + lane) & (XE_RING - 1u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0u;
		const bool is_m = is_tok && (qa & 0x8000u);
		const uint32_t off = qa & 0x7FFFu, byte = qa & 0xFFu;
		const uint32_t nt = cnt, tb = lane;
		const bool lng = is_m && L >= 7u;
		const u64 longmask = __ballot(lng);
		const bool even = !(((uint32_t)R + popc_below(longmask)) & 1u);
		uint32_t sz = 0;
		if (is_tok) { sz = !is_m ? 1u : 2u + (uint32_t)(lng && even) + (uint32_t)(L >= 22u) + (L >= 277u ? (L <= 0xFFFFu ? 2u : 6u) : 0u); }
		const uint32_t incl = wave_incl_scan_add(sz);
		const uint32_t wsum = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		const uint32_t sh = (uint32_t)(N & 31u);                   // tokens already in the flag word in progress
		const uint32_t tq = sh + tb;                               // my token's index counted from that word's first token
		// every position of this step is base + a 32-bit offset: pos(t) = 4*(t div 32 + 1) + sum size(u<t)
		const P base = 4u * (N / 32u + 1u) + S;
		const uint32_t posrel = 4u * (tq >> 5) + (incl - sz);
		const uint32_t kdone = (sh + nt) >> 5;                     // flag words completed by this step (0..2)
		const bool fits = base + 4u * kdone + wsum <= cap;         // uniform: no store of this step can pass the capacity
		// the nibble of the NEXT long match in this step (it shares my byte when my rank is even)
		const u64 above = (longmask >> lane) >> 1;
		const uint32_t nib = L >= 7u ? (L - 7u < 15u ? L - 7u : 15u) : 0u;
		const uint32_t partner = above ? lane + 1u + ctz64(above) : lane;
		const uint32_t pnib = (uint32_t)__shfl((int)nib, (int)partner, 64);
		if (fits) { xe_emit_tokens<false>(out, cap, base, posrel, is_tok, is_m, byte, off, L, lng, even, above != 0, nib, pnib, pend, pend_pos, pend_low, longmask, lane); }
		else      { xe_emit_tokens<true >(out, cap, base, posrel, is_tok, is_m, byte, off, L, lng, even, above != 0, nib, pnib, pend, pend_pos, pend_low, longmask, lane); }
		// ---- flag words: the word in progress lives in scalars, first token in bit 0; a completed word is bit-reversed into its slot
		// (the first token is the MSB, :317-324) ----------
		{
			const u64 M = __ballot(is_m);
			u64 sm = __ballot(is_tok && (tq & 31u) == 0);             // tokens that open a flag word: its slot is the 4 bytes before them
			const u64 lo = (u64)facc | (M << sh);
			const uint32_t hi = sh ? (uint32_t)(M >> (64u - sh)) : 0u;
			P fp = fposc;
			uint32_t w = (uint32_t)lo;
			if (sh == 0) { const uint32_t l = ctz64(sm); fp = base + (uint32_t)__builtin_amdgcn_readlane((int)posrel, (int)l) - 4u; sm &= sm - 1u; }
			if (kdone >= 1u) {
				if (lane == 0) { xe_store32(out, cap, fp, __builtin_bitreverse32(w), !fits); }
				w = (uint32_t)(lo >> 32);
				if (sm) { const uint32_t l = ctz64(sm); fp = base + (uint32_t)__builtin_amdgcn_readlane((int)posrel, (int)l) - 4u; sm &= sm - 1u; }
				if (kdone >= 2u) {
					if (lane == 0) { xe_store32(out, cap, fp, __builtin_bitreverse32(w), !fits); }
					w = hi;
					if (sm) { const uint32_t l = ctz64(sm); fp = base + (uint32_t)__builtin_amdgcn_readlane((int)posrel, (int)l) - 4u; }
				}
			}
			facc = w; fposc = fp;
		}
		// carry
		if (longmask) {
			const uint32_t ll = 63u - (uint32_t)__builtin_clzll(longmask);        // last long match of the step
			const P rl = R + (uint32_t)__popcll(longmask) - 1u;
			pend = !(rl & 1u);
			if (pend) {
				pend_pos = base + (uint32_t)__builtin_amdgcn_readlane((int)posrel, (int)ll) + 2u;
				pend_low = (uint32_t)__builtin_amdgcn_readlane((int)nib, (int)ll);
			}
			R += (uint32_t)__popcll(longmask);
		}
		N += nt;
		S += wsum;
		qh += cnt;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");     // (the ring entries just read may be written again by the next window)
	};
	for (P wbase = 0; wbase < n; wbase += 64u) {
		XE_T(5)
		const uint32_t wi = (uint32_t)((wbase >> 6) & 7u), buf = (uint32_t)((wbase >> 9) & 1u);
		if (wi == 0) {                                            // group start: publish this group's inputs, start loading the next
			XE_BURST_STORE(buf)
			if (wbase + 512u < n) { XE_BURST_LOAD(wbase + 512u) }
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
		}
		const P wend = (wbase + 64u < n) ? wbase + 64u : n;
		const P p = wbase + lane;
		const bool inr = p < n;
		if (cur >= wend) { continue; }                            // window wholly covered by a match
		uint32_t off = inr ? (uint32_t)s_in_off[buf][wi * 64u + lane] : 0u, L = s_in_len[buf][wi * 64u + lane];
		const uint32_t byte = s_in_byte[buf][wi * 64u + lane];
		const u64 mm = __ballot(inr && off != 0 && p >= cur);
		XE_T(0)
		// The serial loop only decides which candidates are TAKEN (incl. the lagging-fill rule); the token mask is derived
		// in parallel afterwards.
		u64 matchmask = 0;
		const P cur_entry = cur;
		const uint32_t wn = (uint32_t)(wend - wbase);
		// Fast path (all but ~1 window in 128): `filled` (F) lies beyond this window and no token of it can reach F, so the
		// lazy-Fill rule cannot fire (F > position for every token start in the window) -- plain 32-bit walk.
		// F only moves when a token start reaches it; a match that ends beyond F is handled by the exact loop below.
		if (F > wend || wbase >= end2) {
			// Every match lane precomputes where the walk goes after taking it: the first candidate at or after its end
			// (relative to the window; >= wn leaves the window), so the scalar loop is one v_readlane per taken match.
			const uint32_t nx = lane + L + 3u;
			const u64 restl = nx < 64u ? mm >> nx : (u64)0;
			const uint32_t J = nx >= wn ? nx : (restl ? nx + ctz64(restl) : wn);
			u64 capm = sgpr64(__ballot(L == 45u) & mm);             // candidates the finder capped at 48: extended on demand
			uint32_t mp;
			{
				const uint32_t rel = (uint32_t)(cur - wbase);
				const u64 rest = mm >> rel;
				mp = rest ? rel + ctz64(rest) : wn;
			}
			mp = (uint32_t)__builtin_amdgcn_readfirstlane((int)mp);
			const uint32_t wn_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wn);
			bool far = false;
			while (mp < wn_s) {
				uint32_t st;
				matchmask = sgpr64(matchmask);
				// v_readlane needs 4 wait states after the write of its lane select (mp): on the loop edge the five scalar
				// instructions in between provide them, on entry the s_nop does.
				asm volatile(
					"s_nop 3\n\t"
					"1:\n\t"
					"s_bitcmp1_b64 %[cap], %[mp]\n\t"
					"s_cbranch_scc1 3f\n\t"
					"s_bitset1_b64 %[mk], %[mp]\n\t"
					"v_readlane_b32 %[mp], %[J], %[mp]\n\t"
					"s_cmp_lt_u32 %[mp], %[wn]\n\t"
					"s_cbranch_scc1 1b\n\t"
					"s_mov_b32 %[st], 0\n\t"
					"s_branch 4f\n\t"
					"3:\n\t"
					"s_mov_b32 %[st], 1\n\t"
					"4:\n\t"
					: [mp] "+s"(mp), [mk] "+s"(matchmask), [st] "=&s"(st)
					: [cap] "s"(capm), [wn] "s"(wn_s), [J] "v"(J)
					: "scc");
				if (st == 0) { break; }
				// the finder capped this match at 48: extend it
				matchmask |= ((u64)1) << mp;
				capm &= ~(((u64)1) << mp);
				const P pm = wbase + mp;
				const P x = pm - (uint32_t)__builtin_amdgcn_readlane((int)off, (int)mp);
				const P ext = (P)(45u + wave_extend(d, x + 48u, pm + 48u, n - pm - 1u - 48u, n, lane));
				if (lane == mp) { L = (uint32_t)ext; }
				if (ext > 0x10000u) { cur = pm + ext + 3u; far = true; break; }
				const uint32_t nxs = mp + (uint32_t)ext + 3u;
				if (nxs >= wn_s) { mp = nxs; }
				else { const u64 rest = mm >> nxs; mp = rest ? nxs + ctz64(rest) : wn_s; }
			}
			if (!far) { cur = wbase + mp; }
		} else
		while (cur < wend) {
			if (cur < end2) {
				if (F <= cur) { F = (F + 0x2000u < end2) ? F + 0x2000u : end2; }          // this token's lazy Fill (:269)
				if (cur >= F) { ++cur; continue; }                                         // lagging fill => literal
			}
			const uint32_t rel = (uint32_t)(cur - wbase);
			const u64 rest = (mm >> rel);
			if (rest == 0) {
				const P lastp = (wend - 1u < end2) ? wend - 1u : end2;              // fills done by the literal tokens up to wend
				if (end2 && F <= lastp && lastp < end2) { F = (F + 0x2000u < end2) ? F + 0x2000u : end2; }
				cur = wend;
				break;
			}
			const uint32_t mp = rel + ctz64(rest);
			const P pm = wbase + mp;
			if (F <= pm) { F = (F + 0x2000u < end2) ? F + 0x2000u : end2; }            // fill by a token in (cur, pm]
			matchmask |= ((u64)1) << mp;
			P Lm = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)mp);
			if (Lm == 45u) {                                      // the finder capped this match at 48: extend it
				const P x = pm - (uint32_t)__builtin_amdgcn_readlane((int)off, (int)mp);
				const P lim = n - pm - 1u;                        // never count the buffer's final byte
				Lm = (P)(45u + wave_extend(d, x + 48u, pm + 48u, lim - 48u, n, lane));
				if (lane == mp) { L = (uint32_t)Lm; }
			}
			cur = pm + Lm + 3u;
		}
		XE_T(1)
		// tokens = positions of the window at/after the entry that no taken match covers
		const bool is_m = (matchmask >> lane) & (u64)1;
		const uint32_t mend = is_m ? (L < 0xFFFFFFu ? lane + L + 3u : 0xFFFFFFFFu) : 0u;       // match end, relative to the window
		const uint32_t reach = wave_incl_scan_max_u32(mend);
		const bool is_tok = p >= cur_entry && inr && (is_m || reach <= lane);
		const u64 tokmask = __ballot(is_tok);

		// ---- queue the window's tokens: rank order = position order (round 5; see emit_step above the loop) ----------------
		{
			const uint32_t tb = popc_below(tokmask);
			if (is_tok) {
				const uint32_t qi = (qt + tb) & (XE_RING - 1u);
				__hip_atomic_store(&s_qa[qi], (uint16_t)(is_m ? (0x8000u | off) : byte), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				__hip_atomic_store(&s_ql[qi], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
			}
			qt += (uint32_t)__popcll(tokmask);
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
		}
		XE_T(2)
		while (qt - qh >= 64u) { emit_step(64u); }
		XE_T(3)
	}
	if (qt != qh) { emit_step(qt - qh); }                       // the last, partial step
#ifdef XE_PROFILE
	if (lane == 0) { for (int i_ = 0; i_ < 6; ++i_) { atomicAdd(&g_xe_prof[i_], xe_acc[i_]); } }
#endif

	// ---- final flag word (:343-344), size, status ----------------------------------------------------------------
	const P gf = N / 32u;
	const uint32_t cnt = (uint32_t)(N & 31u);
	const P total = 4u * (gf + 1u) + S;
	if (lane == 0) {
		uint32_t wv; P fp;
		if (cnt) { wv = __builtin_bitreverse32(facc) | ((1u << (32u - cnt)) - 1u); fp = fposc; }
		else { wv = 0xFFFFFFFFu; fp = total - 4u; }
		put8(out, cap, fp, wv); put8(out, cap, fp + 1u, wv >> 8); put8(out, cap, fp + 2u, wv >> 16); put8(out, cap, fp + 3u, wv >> 24);
		const bool ok = total <= cap;
		d_out_len[u] = ok ? total : 0;
		d_status[u] = ok ? 0 : -5;
	}
}

__global__ __launch_bounds__(64) void xpress_emit_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                        S16 mlen3, S16 moff,
                                                        uint8_t* __restrict__ d_out, u64* __restrict__ d_out_len, int32_t* __restrict__ d_status)
{
	__shared__ uint16_t s_in_off[2][512];
	__shared__ uint16_t s_in_len[2][512];
	__shared__ uint8_t  s_in_byte[2][512];
	__shared__ uint16_t s_qa[XE_RING];                             // the token ring: 0x8000 | offset of a match, or the literal byte ...
	__shared__ uint32_t s_ql[XE_RING];                             // ... and len - 3
	const uint32_t u = blockIdx.x;
	if (bt.in_len[u] < ((u64)1 << 31)) { xpress_emit_body<uint32_t>(d_in, bt, mlen3, moff, d_out, d_out_len, d_status, u, s_in_off, s_in_len, s_in_byte, s_qa, s_ql); }
	else { xpress_emit_body<u64>(d_in, bt, mlen3, moff, d_out, d_out_len, d_status, u, s_in_off, s_in_len, s_in_byte, s_qa, s_ql); }
}

// ===================================================================================================================
// NW = 4 or 16 waves per unit (version 2 of the kernel above; same bytes)
// ===================================================================================================================
// The greedy walk is serial, but its state is small -- the position of the next token and the lazy-Fill boundary -- and
// it re-synchronises within a window or two wherever it is started. Per "super-block" of 1024 windows (64 KiB):
//   1. wave j walks its segment of 1024 / NW windows SPECULATIVELY, as if a token started at the first position of its
//      segment with the Fill boundary not lagging (wave 0 continues the true state); every window leaves its token mask,
//      match mask, lengths, byte counts and the state after it;
//   2. wave 0 repairs the seams in order: it re-walks a segment from the true entry state until the state after a
//      window equals the recorded speculative one;
//   3. wave 0 scans the windows: tokens, long matches (their rank parity decides who owns a length nibble) and bytes
//      before every window;
//   4. wave j EMITS its segment (every window knows its output position); what couples it to its neighbours -- the
//      flag word in progress at its first token, the half-filled nibble byte of the previous long match -- is not
//      written but recorded;
//   5. wave 0 stitches the seams: the shared flag words and nibble bytes.
// A unit longer than 64 KiB runs its super-blocks one after the other (the state is carried), each four waves wide.

// One window of the walk (xpress_compress.cpp:262-345 without the emission). In: the state (cur, F), this lane's
// candidate (off, L = len-3 as capped by the finder, or a length this function produced earlier: it is a fixed point).
// Out: the new state, the final L of the lanes whose match is taken, the token and match masks.
__device__ __forceinline__ void xe_walk_window(const uint8_t* __restrict__ d, u64 n, u64 end2, uint32_t lane, u64 wbase,
                                               u64& cur, u64& F, uint32_t off, uint32_t& L, u64& tokmask, u64& matchmask)
{
	const u64 wend = (wbase + 64u < n) ? wbase + 64u : n;
	tokmask = 0; matchmask = 0;
	if (cur >= wend) { return; }                                  // window wholly covered by a match
	const u64 p = wbase + lane;
	const bool inr = p < n;
	if (!inr) { off = 0; }
	const u64 mm = __ballot(inr && off != 0 && p >= cur);
	const u64 cur_entry = cur;
	const uint32_t wn = (uint32_t)(wend - wbase);
	if (F > wend || wbase >= end2) {
		// fast path: the lazy-Fill rule cannot fire in this window. Jump table + scalar loop as in the kernel above.
		const uint32_t nx = lane + L + 3u;
		const u64 restl = nx < 64u ? mm >> nx : (u64)0;
		const uint32_t J = nx >= wn ? nx : (restl ? nx + ctz64(restl) : wn);
		u64 capm = sgpr64(__ballot(L == 45u) & mm);
		uint32_t mp;
		{
			const uint32_t rel = (uint32_t)(cur - wbase);
			const u64 rest = mm >> rel;
			mp = rest ? rel + ctz64(rest) : wn;
		}
		mp = (uint32_t)__builtin_amdgcn_readfirstlane((int)mp);
		const uint32_t wn_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wn);
		bool far = false;
		while (mp < wn_s) {
			uint32_t st;
			matchmask = sgpr64(matchmask); mp = (uint32_t)__builtin_amdgcn_readfirstlane((int)mp);   // (uniform; tell the compiler)
			asm volatile(
				"s_nop 3\n\t"
				"1:\n\t"
				"s_bitcmp1_b64 %[cap], %[mp]\n\t"
				"s_cbranch_scc1 3f\n\t"
				"s_bitset1_b64 %[mk], %[mp]\n\t"
				"v_readlane_b32 %[mp], %[J], %[mp]\n\t"
				"s_cmp_lt_u32 %[mp], %[wn]\n\t"
				"s_cbranch_scc1 1b\n\t"
				"s_mov_b32 %[st], 0\n\t"
				"s_branch 4f\n\t"
				"3:\n\t"
				"s_mov_b32 %[st], 1\n\t"
				"4:\n\t"
				: [mp] "+s"(mp), [mk] "+s"(matchmask), [st] "=&s"(st)
				: [cap] "s"(sgpr64(capm)), [wn] "s"(wn_s), [J] "v"(J)
				: "scc");
			if (st == 0) { break; }
			matchmask |= ((u64)1) << mp;
			capm &= ~(((u64)1) << mp);
			const u64 pm = wbase + mp;
			const u64 x = pm - (uint32_t)__builtin_amdgcn_readlane((int)off, (int)mp);
			const u64 ext = 45u + wave_extend(d, x + 48u, pm + 48u, n - pm - 1u - 48u, n, lane);
			if (lane == mp) { L = (uint32_t)ext; }
			if (ext > 0x10000u) { cur = pm + ext + 3u; far = true; break; }
			const uint32_t nxs = mp + (uint32_t)ext + 3u;
			if (nxs >= wn_s) { mp = nxs; }
			else { const u64 rest = mm >> nxs; mp = rest ? nxs + ctz64(rest) : wn_s; }
		}
		if (!far) { cur = wbase + mp; }
	} else
	while (cur < wend) {                                          // exact loop with the ONE-lazy-Fill-per-token rule (:269)
		if (cur < end2) {
			if (F <= cur) { F = (F + 0x2000u < end2) ? F + 0x2000u : end2; }
			if (cur >= F) { ++cur; continue; }                      // lagging fill => literal
		}
		const uint32_t rel = (uint32_t)(cur - wbase);
		const u64 rest = (mm >> rel);
		if (rest == 0) {
			const u64 lastp = (wend - 1u < end2) ? wend - 1u : end2;
			if (end2 && F <= lastp && lastp < end2) { F = (F + 0x2000u < end2) ? F + 0x2000u : end2; }
			cur = wend;
			break;
		}
		const uint32_t mp = rel + ctz64(rest);
		const u64 pm = wbase + mp;
		if (F <= pm) { F = (F + 0x2000u < end2) ? F + 0x2000u : end2; }
		matchmask |= ((u64)1) << mp;
		u64 Lm = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)mp);
		if (Lm == 45u) {
			const u64 x = pm - (uint32_t)__builtin_amdgcn_readlane((int)off, (int)mp);
			const u64 lim = n - pm - 1u;
			Lm = 45u + wave_extend(d, x + 48u, pm + 48u, lim - 48u, n, lane);
			if (lane == mp) { L = (uint32_t)Lm; }
		}
		cur = pm + Lm + 3u;
	}
	const bool is_m = (matchmask >> lane) & (u64)1;
	const uint32_t mend = is_m ? (L < 0xFFFFFFu ? lane + L + 3u : 0xFFFFFFFFu) : 0u;
	const uint32_t reach = wave_incl_scan_max_u32(mend);
	const bool is_tok = p >= cur_entry && inr && (is_m || reach <= lane);
	tokmask = __ballot(is_tok);
}

// state after a window, as stored for the seam repair: position relative to the window (saturating) and the index of
// the Fill boundary (a multiple of 0x2000, or the end of the unit)
__device__ __forceinline__ uint32_t xe_pack_cur(u64 cur, u64 wbase) { const u64 r = cur - wbase; return r < 0xFFFFFFFFull ? (uint32_t)r : 0xFFFFFFFFu; }
__device__ __forceinline__ uint32_t xe_pack_F(u64 F, u64 end2) { return F == end2 ? 0xFFFFFFFFu : (uint32_t)(F >> 13); }

// record a walked window: final lengths (u16, 0xFFFF = see the window's far length), masks, byte counts, state after it
__device__ __forceinline__ void xe_store_window(uint32_t lane, u64 wbase, u64 n, u64 gw, uint32_t w, uint32_t off, uint32_t L, u64 tm, u64 mk,
                                                u64 cur, u64 F, u64 end2, S16 mlen3u, u64* __restrict__ wtoku, u64* __restrict__ wmatu,
                                                uint32_t* __restrict__ wfaru, uint32_t* s_ecur, uint32_t* s_eF, uint32_t* s_sum)
{
	const bool is_m = (mk >> lane) & (u64)1;
	if (is_m) { mlen3u[wbase + lane] = (uint16_t)(L < 0xFFFFu ? L : 0xFFFFu); if (L >= 0xFFFFu) { wfaru[gw] = L; } }
	const uint32_t nm = (uint32_t)__popcll(mk), nt = (uint32_t)__popcll(tm);
	const uint32_t nl = (uint32_t)__popcll(__ballot(is_m && L >= 7u));
	const uint32_t n22 = (uint32_t)__popcll(__ballot(is_m && L >= 22u)), n277 = (uint32_t)__popcll(__ballot(is_m && L >= 277u));
	const uint32_t nfar = (uint32_t)__popcll(__ballot(is_m && L > 0xFFFFu));
	const uint32_t fsz = (nt - nm) + 2u * nm + n22 + 2u * n277 + 4u * nfar;
	if (lane == 0) {
		wtoku[gw] = tm; wmatu[gw] = mk;
		s_ecur[w] = xe_pack_cur(cur, wbase); s_eF[w] = xe_pack_F(F, end2); s_sum[w] = nt | (nl << 8) | (fsz << 16);
	}
}

// Emission of one segment (windows [w0, w1) of the super-block that starts at window sb of the unit) from the recorded
// masks and lengths; s_nr[w] = tokens before window w of the super-block | long-match rank parity << 31, s_sr[w] = bytes
// before it (both relative to Nsb / Ssb). What couples the segment to its neighbours is recorded in seam[] / seampos[]:
// seam: lead bits, lead complete, tail bits, tail valid, lead nibble, tail pending, tail low nibble, tokens;
// seampos: slot of the flag word open at the segment's end, position of its pending nibble byte.
__device__ __forceinline__ void xe_emit_segment(uint32_t lane, u64 sb, uint32_t w0, uint32_t w1, u64 n, u64 cap, const uint8_t* __restrict__ d,
                                                uint8_t* __restrict__ out, S16 mlen3u, S16 moffu,
                                                const u64* wtoku, const u64* wmatu, const uint32_t* wfaru, const uint32_t* s_nr, const uint32_t* s_sr,
                                                u64 Nsb, u64 Ssb, uint16_t* s_off, uint16_t* s_len, uint8_t* s_byte, u64* s_mask,
                                                uint32_t* seam, u64* seampos)
{
	uint32_t g_off[4], g_len[4], g_byte[4];
#define XE4_LOAD(gb) { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { \
		const u64 q_ = (gb) + (u64)k_ * 64u + lane; const u64 c_ = q_ < n ? q_ : n - 1u; \
		{ const uint32_t w_ = mlen3u.word(c_); g_off[k_] = w_ >> 16; g_len[k_] = w_ & 0xFFFFu; } g_byte[k_] = d[c_]; } }
#define XE4_STORE() { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { \
		s_off[k_ * 64 + lane] = (uint16_t)g_off[k_]; s_len[k_ * 64 + lane] = (uint16_t)g_len[k_]; s_byte[k_ * 64 + lane] = (uint8_t)g_byte[k_]; } }
	if (w0 >= w1) { if (lane == 0) { seam[7] = 0; seam[4] = 0xFFu; seam[5] = 0; } return; }
	{
	bool pend = false; u64 pend_pos = 0; uint32_t pend_low = 0;
	uint32_t facc = 0; u64 fposc = 0;
	bool lead_open = false, lead_done = false; uint32_t lead_bits = 0, lead_nib = 0xFFu; bool tail_started = false;
	bool first = true; uint32_t ntok_seg = 0;
	// (the masks of the 4 windows of a burst travel with it: lanes 0-3 / 4-7 fetch them)
	u64 g_mask = 0;
#define XE2_LOAD_MASKS(wq) { g_mask = 0; if (lane < 8u && (wq) + (lane & 3u) < w1) { \
g_mask = __hip_atomic_load((lane < 4u ? wtoku : wmatu) + sb + (wq) + (lane & 3u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
	XE4_LOAD((sb + w0) * 64u) XE2_LOAD_MASKS(w0)
	for (uint32_t w = w0; w < w1; ++w) {
		const uint32_t wi = (w - w0) & 3u;
		if (wi == 0) {
			XE4_STORE()
			if (lane < 8u) { s_mask[lane] = g_mask; }
			if (w + 4u < w1) { XE4_LOAD((sb + w + 4u) * 64u) XE2_LOAD_MASKS(w + 4u) }
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
		}
		const u64 tokmask = s_mask[wi];
		if (tokmask == 0) { continue; }
		const u64 matchmask = s_mask[4u + wi];
		const uint32_t off = s_off[wi * 64u + lane], byte = s_byte[wi * 64u + lane];
		uint32_t L = s_len[wi * 64u + lane];
		if (L == 0xFFFFu) { L = __hip_atomic_load(&wfaru[sb + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		const bool is_tok = (tokmask >> lane) & (u64)1, is_m = (matchmask >> lane) & (u64)1;
		const u64 N = Nsb + (s_nr[w] & 0x7FFFFFFFu), S = Ssb + s_sr[w];
		const uint32_t Rpar = s_nr[w] >> 31;
		if (first) { first = false; lead_open = (N & 31u) != 0; }

		const uint32_t nt = (uint32_t)__popcll(tokmask);
		ntok_seg += nt;
		const bool lng = is_m && L >= 7u;
		const u64 longmask = __ballot(lng);
		const bool even = !((Rpar + popc_below(longmask)) & 1u);
		uint32_t sz = 0;
		if (is_tok) { sz = !is_m ? 1u : 2u + (uint32_t)(lng && even) + (uint32_t)(L >= 22u) + (L >= 277u ? (L <= 0xFFFFu ? 2u : 6u) : 0u); }
		const uint32_t incl = wave_incl_scan_add(sz);
		const uint32_t wsum = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		const uint32_t tb = popc_below(tokmask);
		const uint32_t sh = (uint32_t)(N & 31u);
		const uint32_t tq = sh + tb;
		const u64 base = 4u * (N / 32u + 1u) + S;
		const uint32_t posrel = 4u * (tq >> 5) + (incl - sz);
		const uint32_t kdone = (sh + nt) >> 5;
		const bool fits = base + 4u * kdone + wsum <= cap;
		const u64 above = (longmask >> lane) >> 1;
		const uint32_t nib = L >= 7u ? (L - 7u < 15u ? L - 7u : 15u) : 0u;
		const uint32_t partner = above ? lane + 1u + ctz64(above) : lane;
		const uint32_t pnib = (uint32_t)__shfl((int)nib, (int)partner, 64);
		// the first long match of the segment with an odd rank completes a byte owned by an earlier segment: recorded
		if (longmask && lead_nib == 0xFFu && !pend) {
			const uint32_t fl = ctz64(longmask);
			const uint32_t fn = (uint32_t)__builtin_amdgcn_readlane((int)nib, (int)fl);
			if (Rpar & 1u) { lead_nib = fn | 0x100u; } else { lead_nib = 0x200u; }   // 0x1xx: odd first rank (nibble xx); 0x200: nothing to complete
		}
		if (fits) { xe_emit_tokens<false>(out, cap, base, posrel, is_tok, is_m, byte, off, L, lng, even, above != 0, nib, pnib, pend, pend_pos, pend_low, longmask, lane); }
		else      { xe_emit_tokens<true >(out, cap, base, posrel, is_tok, is_m, byte, off, L, lng, even, above != 0, nib, pnib, pend, pend_pos, pend_low, longmask, lane); }
		{
			const uint32_t dst = is_tok ? tb : nt + (lane - tb);
			const uint32_t fm = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (is_tok && is_m) ? 1 : 0);
			const u64 M = __ballot(fm != 0);
			u64 sm = __ballot(is_tok && (tq & 31u) == 0);
			const u64 lo = (u64)facc | (M << sh);
			const uint32_t hi = sh ? (uint32_t)(M >> (64u - sh)) : 0u;
			u64 fp = fposc;
			uint32_t wd = (uint32_t)lo;
			bool started = tail_started;                            // does the word in `wd` start inside this segment?
			if (sh == 0) { const uint32_t l = ctz64(sm); fp = base + (uint32_t)__builtin_amdgcn_readlane((int)posrel, (int)l) - 4u; sm &= sm - 1u; started = true; }
			if (kdone >= 1u) {
				if (started) { if (lane == 0) { xe_store32(out, cap, fp, __builtin_bitreverse32(wd), !fits); } }
				else { lead_bits = wd; lead_done = true; }            // the word that was open at the segment's first token
				wd = (uint32_t)(lo >> 32); started = false;
				if (sm) { const uint32_t l = ctz64(sm); fp = base + (uint32_t)__builtin_amdgcn_readlane((int)posrel, (int)l) - 4u; sm &= sm - 1u; started = true; }
				if (kdone >= 2u) {
					if (lane == 0) { xe_store32(out, cap, fp, __builtin_bitreverse32(wd), !fits); }
					wd = hi; started = false;
					if (sm) { const uint32_t l = ctz64(sm); fp = base + (uint32_t)__builtin_amdgcn_readlane((int)posrel, (int)l) - 4u; started = true; }
				}
			}
			facc = wd; fposc = fp; tail_started = started;
		}
		if (longmask) {
			const uint32_t ll = 63u - (uint32_t)__builtin_clzll(longmask);
			const uint32_t rl = Rpar + (uint32_t)__popcll(longmask) - 1u;
			pend = !(rl & 1u);
			if (pend) {
				pend_pos = base + (uint32_t)__builtin_amdgcn_readlane((int)posrel, (int)ll) + 2u;
				pend_low = (uint32_t)__builtin_amdgcn_readlane((int)nib, (int)ll);
			}
		}
	}
	if (lane == 0) {
		// lead: bits of the word that was open at my first token (complete or not); tail: the word open at my end
		seam[0] = lead_open ? (lead_done ? lead_bits : facc) : 0u;
		seam[1] = (lead_open && lead_done) ? 1u : 0u;
		seam[2] = (lead_open && !lead_done) ? 0u : facc;
		seam[3] = ((!lead_open || lead_done) && tail_started) ? 1u : 0u;
		seam[4] = lead_nib;
		seam[5] = pend ? 1u : 0u;
		seam[6] = pend_low;
		seam[7] = ntok_seg;
		seampos[0] = fposc; seampos[1] = pend_pos;
	}
	}
#undef XE4_LOAD
#undef XE4_STORE
#undef XE2_LOAD_MASKS
}

#ifdef XE2_PROFILE
__device__ unsigned long long g_xe2_prof[16];
extern "C" void mscomp_amd_debug_xe2_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xe2_prof), 128); unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xe2_prof), z, 128); }
#define XE2_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); if (tid == 0) { atomicAdd(&g_xe2_prof[i], t_ - x2_prev); atomicMax(&g_xe2_prof[8 + (i)], t_ - x2_prev); } x2_prev = t_; }
#else
#define XE2_T(i)
#endif
template <uint32_t NW>                                         // waves per unit: 4 or 16 (segments of 1024 / NW windows)
__global__ __launch_bounds__(NW * 64u) void xpress_emit2_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                          S16 mlen3, S16 moff,
                                                          u64* __restrict__ wtok, u64* __restrict__ wmat, uint32_t* __restrict__ wfar,
                                                          uint8_t* __restrict__ d_out, u64* __restrict__ d_out_len, int32_t* __restrict__ d_status)
{
	__shared__ uint32_t s_ecur[1024], s_eF[1024];                  // state after every window (phases 1-2); then: tokens | rank parity, bytes before it
	__shared__ uint32_t s_sum[1024];                               // nt | nlong << 8 | fixed bytes << 16 of every window
	__shared__ uint16_t s_in_off[NW][256];
	__shared__ uint16_t s_in_len[NW][256];
	__shared__ uint8_t  s_in_byte[NW][256];
	__shared__ u64      s_in_mask[NW][8];                           // token masks [0..3] and match masks [4..7] of the staged windows
	__shared__ u64      s_seam_pos[NW][2];                          // per segment: [0] slot of the flag word open at its end, [1] pending nibble byte
	__shared__ uint32_t s_seam[NW][8];                              // per segment: lead bits, lead complete, tail bits, tail valid, lead nib, tail pend, tail low, tokens
	__shared__ u64      s_state[4];                                // carried: cur, F (written by wave 0 after the repair)
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
	const uint32_t u = blockIdx.x;
	const u64 n = bt.in_len[u];
	const u64 cap = bt.out_cap[u];
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	uint8_t* __restrict__ out = d_out + bt.out_off[u];
	const u64 mbase = (u64)bt.chunk_prefix[u] * 65536u;           // this unit's slice of the per-position match arrays
	S16 mlen3u = mlen3 + mbase;
	S16 moffu = moff + mbase;
	u64* __restrict__ wtoku = wtok + (mbase >> 6); u64* __restrict__ wmatu = wmat + (mbase >> 6); uint32_t* __restrict__ wfaru = wfar + (mbase >> 6);
	const u64 end2 = n >= 2u ? n - 2u : 0u;
	const u64 nwin = (n + 63u) >> 6;

	// carried over the super-blocks (wave 0 owns them; cur/F are broadcast through s_state)
	u64 g_cur = 0, g_F = 0, g_N = 0, g_S = 0, g_R = 0;
	uint32_t g_acc = 0; u64 g_fpos = 0; bool g_pend = false; u64 g_pend_pos = 0; uint32_t g_pend_low = 0;

	uint32_t g_off[4], g_len[4], g_byte[4];
#define XE2_LOAD(gb) { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { \
		const u64 q_ = (gb) + (u64)k_ * 64u + lane; const u64 c_ = q_ < n ? q_ : n - 1u; \
		{ const uint32_t w_ = mlen3u.word(c_); g_off[k_] = w_ >> 16; g_len[k_] = w_ & 0xFFFFu; } g_byte[k_] = d[c_]; } }
#define XE2_STORE() { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { \
		s_in_off[wv][k_ * 64 + lane] = (uint16_t)g_off[k_]; s_in_len[wv][k_ * 64 + lane] = (uint16_t)g_len[k_]; s_in_byte[wv][k_ * 64 + lane] = (uint8_t)g_byte[k_]; } }

#ifdef XE2_PROFILE
	unsigned long long x2_prev = __builtin_readcyclecounter();
#endif
	for (u64 sb = 0; sb < nwin; sb += 1024u) {                     // super-block = windows [sb, sb + nsb)
		const uint32_t nsb = (nwin - sb < 1024u) ? (uint32_t)(nwin - sb) : 1024u;
		const uint32_t w0 = wv * (1024u / NW), w1 = (w0 + (1024u / NW) < nsb) ? w0 + (1024u / NW) : nsb;   // my segment (may be empty)
		// ---- 1. speculative walk of my segment ---------------------------------------------------------------------
		if (w0 < w1) {
			u64 cur, F;
			if (wv == 0) { cur = g_cur; F = g_F; }
			else { cur = (sb + w0) * 64u; F = (cur < end2) ? cur : end2; }   // a token starts here, Fill not lagging
			XE2_LOAD((sb + w0) * 64u)
			for (uint32_t w = w0; w < w1; ++w) {
				const uint32_t wi = (w - w0) & 3u;
				if (wi == 0) {
					XE2_STORE()
					if (w + 4u < w1) { XE2_LOAD((sb + w + 4u) * 64u) }
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
				}
				const u64 wbase = (sb + w) * 64u;
				const uint32_t off = s_in_off[wv][wi * 64u + lane];
				uint32_t L = s_in_len[wv][wi * 64u + lane];
				u64 tm, mk;
				xe_walk_window(d, n, end2, lane, wbase, cur, F, off, L, tm, mk);
				xe_store_window(lane, wbase, n, sb + w, w, off, L, tm, mk, cur, F, end2, mlen3u, wtoku, wmatu, wfaru, s_ecur, s_eF, s_sum);
			}
		}
		XE2_T(0)
		__syncthreads();
		XE2_T(1)
		// ---- 2. repair the seams (wave 0) --------------------------------------------------------------------------
		if (wv == 0) {
			for (uint32_t j = 1; j * (1024u / NW) < nsb; ++j) {
				const uint32_t a = j * (1024u / NW), b = (a + (1024u / NW) < nsb) ? a + (1024u / NW) : nsb;
				const u64 wprev = (sb + a - 1u) * 64u;
				const uint32_t pc = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ecur[a - 1u]), pf = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_eF[a - 1u]);
				u64 cur = wprev + pc;                                  // true entry state of segment j
				u64 F = (pf == 0xFFFFFFFFu) ? end2 : ((u64)pf << 13);
				const u64 s0 = (sb + a) * 64u;
				if (cur == s0 && F == ((s0 < end2) ? s0 : end2)) { continue; }   // the speculation was right
				for (uint32_t w = a; w < b; ++w) {
					const u64 wbase = (sb + w) * 64u;
					const u64 q = wbase + lane, c = q < n ? q : n - 1u;
					const uint32_t off = moffu[c];
					uint32_t L = __hip_atomic_load(&mlen3u[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (written by another wave)
					if (L == 0xFFFFu) { L = __hip_atomic_load(&wfaru[sb + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
					u64 tm, mk;
					xe_walk_window(d, n, end2, lane, wbase, cur, F, off, L, tm, mk);
					const uint32_t sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ecur[w]), sf = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_eF[w]);
					xe_store_window(lane, wbase, n, sb + w, w, off, L, tm, mk, cur, F, end2, mlen3u, wtoku, wmatu, wfaru, s_ecur, s_eF, s_sum);
					if (xe_pack_cur(cur, wbase) == sc && xe_pack_F(F, end2) == sf) { break; }   // re-synchronised
				}
			}
			// the true state after the super-block
			const u64 wl = (sb + nsb - 1u) * 64u;
			const uint32_t pc = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ecur[nsb - 1u]), pf = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_eF[nsb - 1u]);
			g_cur = wl + pc; g_F = (pf == 0xFFFFFFFFu) ? end2 : ((u64)pf << 13);
			// ---- 3. scan: tokens / long-match rank parity / bytes before every window ---------------------------------
			uint32_t nc = 0, sc = 0, rc = (uint32_t)g_R;
			for (uint32_t b0 = 0; b0 < nsb; b0 += 64u) {
				const uint32_t w = b0 + lane;
				const uint32_t v = w < nsb ? s_sum[w] : 0u;
				const uint32_t nt = v & 0xFFu, nl = (v >> 8) & 0xFFu, fsz = v >> 16;
				const uint32_t ri = wave_incl_scan_add(nl);
				const uint32_t rb = rc + ri - nl;                       // long matches before this window
				const uint32_t sz = fsz + (((rb & 1u) == 0) ? (nl + 1u) >> 1 : nl >> 1);   // even ranks own a nibble byte
				const uint32_t ni = wave_incl_scan_add(nt), si = wave_incl_scan_add(sz);
				if (w < nsb) { s_ecur[w] = (nc + ni - nt) | ((rb & 1u) << 31); s_eF[w] = sc + si - sz; }
				nc += (uint32_t)__builtin_amdgcn_readlane((int)ni, 63); sc += (uint32_t)__builtin_amdgcn_readlane((int)si, 63); rc += (uint32_t)__builtin_amdgcn_readlane((int)ri, 63);
			}
			if (lane == 0) { s_state[0] = g_N; s_state[1] = g_S; s_state[2] = nc; s_state[3] = sc; }
			g_R += (u64)(rc - (uint32_t)g_R);
		}
		XE2_T(2)
		__syncthreads();
		XE2_T(3)
		// ---- 4. emission of my segment -----------------------------------------------------------------------------
		xe_emit_segment(lane, sb, w0, w1, n, cap, d, out, mlen3u, moffu, wtoku, wmatu, wfaru, s_ecur, s_eF, s_state[0], s_state[1],
		                s_in_off[wv], s_in_len[wv], s_in_byte[wv], s_in_mask[wv], s_seam[wv], s_seam_pos[wv]);
		XE2_T(4)
		__syncthreads();
		XE2_T(5)
		// ---- 5. stitch the seams (wave 0) --------------------------------------------------------------------------
		if (wv == 0) {
			for (uint32_t j = 0; j < NW; ++j) {
				if (s_seam[j][7] == 0) { continue; }                     // no token starts in this segment
				// flag word in progress
				if (s_seam[j][1]) {                                      // the open word was completed inside segment j
					const uint32_t wdv = g_acc | s_seam[j][0];
					if (lane == 0) { xe_store32(out, cap, g_fpos, __builtin_bitreverse32(wdv), true); }
					g_acc = 0;
				} else { g_acc |= s_seam[j][0]; }
				if (s_seam[j][3]) { g_acc = s_seam[j][2]; g_fpos = s_seam_pos[j][0]; }
				// nibble byte in progress
				const uint32_t ln = s_seam[j][4];
				if (ln & 0x100u) { if (g_pend && lane == 0) { put8(out, cap, g_pend_pos, g_pend_low | ((ln & 0xFu) << 4)); } g_pend = false; }
				if (ln != 0xFFu) { g_pend = s_seam[j][5] != 0; g_pend_pos = s_seam_pos[j][1]; g_pend_low = s_seam[j][6]; }
			}
			g_N += s_state[2]; g_S += s_state[3];
			if (lane == 0) { s_state[0] = g_cur; s_state[1] = g_F; }
		}
		__syncthreads();
	}
	// ---- final flag word (:343-344), size, status ----------------------------------------------------------------
	if (wv == 0) {
		const u64 gf = g_N / 32u;
		const uint32_t cnt = (uint32_t)(g_N & 31u);
		const u64 total = 4u * (gf + 1u) + g_S;
		if (lane == 0) {
			uint32_t wvv; u64 fp;
			if (cnt) { wvv = __builtin_bitreverse32(g_acc) | ((1u << (32u - cnt)) - 1u); fp = g_fpos; }
			else { wvv = 0xFFFFFFFFu; fp = total - 4u; }
			put8(out, cap, fp, wvv); put8(out, cap, fp + 1u, wvv >> 8); put8(out, cap, fp + 2u, wvv >> 16); put8(out, cap, fp + 3u, wvv >> 24);
			const bool ok = total <= cap;
			d_out_len[u] = ok ? total : 0;
			d_status[u] = ok ? 0 : -5;
		}
	}
#undef XE2_LOAD
#undef XE2_STORE
}

// ===================================================================================================================
// One block per 64 KiB super-block (mode 4): the same five phases, spread over four kernels, so that ONE long stream is
// parsed and emitted by the whole GPU
// ===================================================================================================================
//   xe3_walk_kernel   (block per super-block, 16 waves): speculative walk of the 16 segments -- the super-block's own
//                     entry is speculative too, except for the first one of a unit -- in-block seam repair, per-window
//                     records to global memory, in-block scans for both parities of the long-match rank at its entry;
//   xe3_seam_kernel   (wave per super-block): repairs the seam in FRONT of its super-block (re-walk until the recorded
//                     state is met, then re-scan the super-block);
//   xe3_fix_kernel    (wave per unit): cascades the rare repair that moved a super-block's end state, then the prefix
//                     of tokens / long matches / bytes over the super-blocks, 64 per step;
//   xe3_emit_kernel   (block per super-block): emission of its 16 segments, seam records to global memory;
//   xe3_stitch_kernel (wave per unit): stitches every seam of the unit in order, final flag word, size, status.
struct Xe3 {
	u64* wtok; u64* wmat; uint32_t* wfar;                          // per window (exist for the in-block kernel too)
	uint32_t* wecur; uint32_t* weF; uint32_t* wsum;               // per window: state after it, counts
	uint32_t* wnr; uint32_t* ws0; uint32_t* ws1;                  // per window: tokens before | rank parity << 31 (relative to the super-block), bytes before (entry parity 0 / 1)
	uint32_t* sbtot;                                              // per super-block: tokens, long matches, bytes (entry parity 0 / 1)
	u64* sbpre;                                                   // per super-block: tokens, bytes, long matches before it
	uint32_t* seam; u64* seampos;                                 // per super-block: 16 x 8, 16 x 2 (xe_emit_segment)
};

// scan of one super-block's windows (one wave): sum = nt | nlong << 8 | fixed bytes << 16 per window
__device__ __forceinline__ void xe3_scan_sb(uint32_t lane, uint32_t nsb, const uint32_t* sum, uint32_t* wnr, uint32_t* ws0, uint32_t* ws1, uint32_t* tot)
{
	uint32_t nc = 0, rc = 0, sc0 = 0, sc1 = 0;
	for (uint32_t b0 = 0; b0 < nsb; b0 += 64u) {
		const uint32_t w = b0 + lane;
		const uint32_t v = w < nsb ? sum[w] : 0u;
		const uint32_t nt = v & 0xFFu, nl = (v >> 8) & 0xFFu, fsz = v >> 16;
		const uint32_t ri = wave_incl_scan_add(nl);
		const uint32_t rb = rc + ri - nl;                             // long matches before this window inside the super-block
		const uint32_t z0 = fsz + (((rb & 1u) == 0) ? (nl + 1u) >> 1 : nl >> 1);   // entry parity 0: even relative ranks own a nibble byte
		const uint32_t z1 = fsz + (((rb & 1u) != 0) ? (nl + 1u) >> 1 : nl >> 1);   // entry parity 1
		const uint32_t ni = wave_incl_scan_add(nt), s0 = wave_incl_scan_add(z0), s1 = wave_incl_scan_add(z1);
		if (w < nsb) { wnr[w] = (nc + ni - nt) | ((rb & 1u) << 31); ws0[w] = sc0 + s0 - z0; ws1[w] = sc1 + s1 - z1; }
		nc += (uint32_t)__builtin_amdgcn_readlane((int)ni, 63); rc += (uint32_t)__builtin_amdgcn_readlane((int)ri, 63);
		sc0 += (uint32_t)__builtin_amdgcn_readlane((int)s0, 63); sc1 += (uint32_t)__builtin_amdgcn_readlane((int)s1, 63);
	}
	if (lane == 0) { tot[0] = nc; tot[1] = rc; tot[2] = sc0; tot[3] = sc1; }
}

__global__ __launch_bounds__(1024) void xe3_walk_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, S16 mlen3,
                                                       S16 moff, Xe3 x)
{
	__shared__ uint32_t s_ecur[1024], s_eF[1024], s_sum[1024];
	__shared__ uint16_t s_in_off[16][256];
	__shared__ uint16_t s_in_len[16][256];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
	const uint32_t lc = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	const uint32_t k = lc - bt.chunk_prefix[u];
	const u64 n = bt.in_len[u];
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	const u64 mbase = (u64)bt.chunk_prefix[u] * 65536u;
	S16 mlen3u = mlen3 + mbase;
	S16 moffu = moff + mbase;
	const u64 gwu = mbase >> 6;                                    // first window record of the unit
	const u64 end2 = n >= 2u ? n - 2u : 0u;
	const u64 nwin = (n + 63u) >> 6;
	const u64 sb = (u64)k * 1024u;
	const uint32_t nsb = sb >= nwin ? 0u : ((nwin - sb < 1024u) ? (uint32_t)(nwin - sb) : 1024u);
	const uint32_t w0 = wv * 64u, w1 = (w0 + 64u < nsb) ? w0 + 64u : nsb;
	uint32_t g_off[4], g_len[4];
#define XE3_LOAD(gb) { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { \
		const u64 q_ = (gb) + (u64)k_ * 64u + lane; const u64 c_ = q_ < n ? q_ : n - 1u; { const uint32_t w_ = mlen3u.word(c_); g_off[k_] = w_ >> 16; g_len[k_] = w_ & 0xFFFFu; } } }
#define XE3_STORE() { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { \
		s_in_off[wv][k_ * 64 + lane] = (uint16_t)g_off[k_]; s_in_len[wv][k_ * 64 + lane] = (uint16_t)g_len[k_]; } }
	if (w0 < w1) {                                                // ---- speculative walk of my segment
		u64 cur = (sb + w0) * 64u, F = (cur < end2) ? cur : end2;   // a token starts here, Fill not lagging (exact for the unit's first position)
		if (sb + w0 == 0) { F = 0; }
		XE3_LOAD((sb + w0) * 64u)
		for (uint32_t w = w0; w < w1; ++w) {
			const uint32_t wi = (w - w0) & 3u;
			if (wi == 0) {
				XE3_STORE()
				if (w + 4u < w1) { XE3_LOAD((sb + w + 4u) * 64u) }
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
			}
			const u64 wbase = (sb + w) * 64u;
			const uint32_t off = s_in_off[wv][wi * 64u + lane];
			uint32_t L = s_in_len[wv][wi * 64u + lane];
			u64 tm, mk;
			xe_walk_window(d, n, end2, lane, wbase, cur, F, off, L, tm, mk);
			xe_store_window(lane, wbase, n, sb + w, w, off, L, tm, mk, cur, F, end2, mlen3u, x.wtok + gwu, x.wmat + gwu, x.wfar + gwu, s_ecur, s_eF, s_sum);
		}
	}
#undef XE3_LOAD
#undef XE3_STORE
	__syncthreads();
	if (wv == 0) {                                                // ---- seams inside the super-block
		for (uint32_t j = 1; j * 64u < nsb; ++j) {
			const uint32_t a = j * 64u, b = (a + 64u < nsb) ? a + 64u : nsb;
			const u64 wprev = (sb + a - 1u) * 64u;
			const uint32_t pc = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ecur[a - 1u]), pf = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_eF[a - 1u]);
			u64 cur = wprev + pc;
			u64 F = (pf == 0xFFFFFFFFu) ? end2 : ((u64)pf << 13);
			const u64 s0 = (sb + a) * 64u;
			if (cur == s0 && F == ((s0 < end2) ? s0 : end2)) { continue; }
			for (uint32_t w = a; w < b; ++w) {
				const u64 wbase = (sb + w) * 64u;
				const u64 q = wbase + lane, c = q < n ? q : n - 1u;
				const uint32_t off = moffu[c];
				uint32_t L = __hip_atomic_load(&mlen3u[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (L == 0xFFFFu) { L = __hip_atomic_load(&x.wfar[gwu + sb + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
				u64 tm, mk;
				xe_walk_window(d, n, end2, lane, wbase, cur, F, off, L, tm, mk);
				const uint32_t sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ecur[w]), sf = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_eF[w]);
				xe_store_window(lane, wbase, n, sb + w, w, off, L, tm, mk, cur, F, end2, mlen3u, x.wtok + gwu, x.wmat + gwu, x.wfar + gwu, s_ecur, s_eF, s_sum);
				if (xe_pack_cur(cur, wbase) == sc && xe_pack_F(F, end2) == sf) { break; }
			}
		}
	}
	__syncthreads();
	for (uint32_t w = tid; w < nsb; w += 1024u) { x.wecur[gwu + sb + w] = s_ecur[w]; x.weF[gwu + sb + w] = s_eF[w]; x.wsum[gwu + sb + w] = s_sum[w]; }
	if (wv == 0) { xe3_scan_sb(lane, nsb, s_sum, x.wnr + gwu + sb, x.ws0 + gwu + sb, x.ws1 + gwu + sb, x.sbtot + (u64)lc * 4u); }
}

// Repair of the seam in front of super-block k of a unit (one wave): re-walk from the true entry state until the state
// after a window equals the recorded one, then re-scan the super-block. Returns without touching anything when the
// speculation was right. used[2 lc], used[2 lc + 1] remember the entry state this was decided on.
__device__ __forceinline__ void xe3_repair_seam(const uint8_t* __restrict__ d, u64 n, u64 end2, uint32_t lane, uint32_t lc, u64 sb, uint32_t nsb,
                                                S16 mlen3u, S16 moffu, u64 gwu, const Xe3& x, uint32_t* __restrict__ used)
{
	const u64 wprev = (sb - 1u) * 64u;
	const uint32_t pc = __hip_atomic_load(&x.wecur[gwu + sb - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const uint32_t pf = __hip_atomic_load(&x.weF[gwu + sb - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (lane == 0) { used[2u * lc] = pc; used[2u * lc + 1u] = pf; }
	u64 cur = wprev + pc;
	u64 F = (pf == 0xFFFFFFFFu) ? end2 : ((u64)pf << 13);
	const u64 s0 = sb * 64u;
	if (cur == s0 && F == ((s0 < end2) ? s0 : end2)) { return; }  // the speculation was right
	for (uint32_t w = 0; w < nsb; ++w) {
		const u64 wbase = (sb + w) * 64u;
		const u64 q = wbase + lane, c = q < n ? q : n - 1u;
		const uint32_t off = moffu[c];
		uint32_t L = __hip_atomic_load(&mlen3u[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (L == 0xFFFFu) { L = __hip_atomic_load(&x.wfar[gwu + sb + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		u64 tm, mk;
		xe_walk_window(d, n, end2, lane, wbase, cur, F, off, L, tm, mk);
		const uint32_t sc = __hip_atomic_load(&x.wecur[gwu + sb + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t sf = __hip_atomic_load(&x.weF[gwu + sb + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		xe_store_window(lane, wbase, n, sb + w, w, off, L, tm, mk, cur, F, end2, mlen3u, x.wtok + gwu, x.wmat + gwu, x.wfar + gwu,
		                x.wecur + gwu + sb, x.weF + gwu + sb, x.wsum + gwu + sb);
		if (xe_pack_cur(cur, wbase) == sc && xe_pack_F(F, end2) == sf) { break; }   // re-synchronised
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	xe3_scan_sb(lane, nsb, x.wsum + gwu + sb, x.wnr + gwu + sb, x.ws0 + gwu + sb, x.ws1 + gwu + sb, x.sbtot + (u64)lc * 4u);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// all seams between super-blocks at once, one wave each (a repair that does not re-synchronise inside its super-block
// changes that super-block's end state: xe3_fix_kernel notices and cascades)
__global__ __launch_bounds__(64) void xe3_seam_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, S16 mlen3,
                                                     S16 moff, Xe3 x, uint32_t* __restrict__ used)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t lc = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	const uint32_t k = lc - bt.chunk_prefix[u];
	if (k == 0) { return; }
	const u64 n = bt.in_len[u];
	const u64 mbase = (u64)bt.chunk_prefix[u] * 65536u;
	const u64 nwin = (n + 63u) >> 6;
	const u64 sb = (u64)k * 1024u;
	const uint32_t nsb = sb >= nwin ? 0u : ((nwin - sb < 1024u) ? (uint32_t)(nwin - sb) : 1024u);
	if (nsb == 0) { return; }
	xe3_repair_seam(d_in + bt.in_off[u], n, n >= 2u ? n - 2u : 0u, lane, lc, sb, nsb, mlen3 + mbase, moff + mbase, mbase >> 6, x, used);
}

// per unit (one wave): cascade check of the seams (serial, but it only reads two words per super-block unless a repair
// moved an end state), then the prefix of tokens / long matches / bytes over the super-blocks, 64 of them per step
__global__ __launch_bounds__(64) void xe3_fix_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, S16 mlen3,
                                                    S16 moff, Xe3 x, uint32_t* __restrict__ used,
                                                    u64* __restrict__ d_out_len, int32_t* __restrict__ d_status)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t u = blockIdx.x;
	const u64 n = bt.in_len[u];
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	const u64 mbase = (u64)bt.chunk_prefix[u] * 65536u;
	const u64 gwu = mbase >> 6;
	const u64 end2 = n >= 2u ? n - 2u : 0u;
	const u64 nwin = (n + 63u) >> 6;
	const uint32_t lc0 = bt.chunk_prefix[u], nk = bt.chunk_prefix[u + 1] - lc0;
	// ---- cascade check: does every seam still see the end state it was repaired against?
	for (uint32_t k0 = 1; k0 < nk; k0 += 64u) {
		const uint32_t k = k0 + lane;
		bool bad = false;
		if (k < nk && (u64)k * 1024u < nwin) {
			const u64 sb = (u64)k * 1024u;
			const uint32_t pc = __hip_atomic_load(&x.wecur[gwu + sb - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t pf = __hip_atomic_load(&x.weF[gwu + sb - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			bad = pc != used[2u * (lc0 + k)] || pf != used[2u * (lc0 + k) + 1u];
		}
		u64 bm = __ballot(bad);
		while (bm) {                                                // rare: repair in order; a repair may move the next seam too
			uint32_t kk = k0 + ctz64(bm);
			bm &= bm - 1u;
			for (;;) {
				const u64 sb = (u64)kk * 1024u;
				const uint32_t nsb = (nwin - sb < 1024u) ? (uint32_t)(nwin - sb) : 1024u;
				xe3_repair_seam(d, n, end2, lane, lc0 + kk, sb, nsb, mlen3 + mbase, moff + mbase, gwu, x, used);
				++kk;
				if (kk >= nk || (u64)kk * 1024u >= nwin) { break; }
				const u64 sb2 = (u64)kk * 1024u;
				const uint32_t pc = __hip_atomic_load(&x.wecur[gwu + sb2 - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const uint32_t pf = __hip_atomic_load(&x.weF[gwu + sb2 - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const uint32_t uc = __hip_atomic_load(&used[2u * (lc0 + kk)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const uint32_t uf = __hip_atomic_load(&used[2u * (lc0 + kk) + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (pc == uc && pf == uf) { break; }
				if (kk < k0 + 64u) { bm &= ~(((u64)1) << (kk - k0)); }
			}
		}
	}
	// ---- prefix over the super-blocks (lane = super-block)
	u64 N = 0, S = 0, R = 0;
	for (uint32_t k0 = 0; k0 < nk; k0 += 64u) {
		const uint32_t k = k0 + lane;
		const bool in = k < nk && (u64)k * 1024u < nwin;
		const u64 ti = (u64)(lc0 + k) * 4u;
		const uint32_t t0 = in ? __hip_atomic_load(&x.sbtot[ti], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
		const uint32_t t1 = in ? __hip_atomic_load(&x.sbtot[ti + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
		const uint32_t z0 = in ? __hip_atomic_load(&x.sbtot[ti + 2u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
		const uint32_t z1 = in ? __hip_atomic_load(&x.sbtot[ti + 3u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
		const uint32_t ri = wave_incl_scan_add(t1);
		const u64 Rb = R + (ri - t1);                                // long matches before my super-block
		const uint32_t sz = (Rb & 1u) ? z1 : z0;
		const uint32_t ni = wave_incl_scan_add(t0), si = wave_incl_scan_add(sz);
		if (k < nk) { x.sbpre[(u64)(lc0 + k) * 3u] = N + (ni - t0); x.sbpre[(u64)(lc0 + k) * 3u + 1u] = S + (si - sz); x.sbpre[(u64)(lc0 + k) * 3u + 2u] = Rb; }
		N += (uint32_t)__builtin_amdgcn_readlane((int)ni, 63); S += (uint32_t)__builtin_amdgcn_readlane((int)si, 63); R += (uint32_t)__builtin_amdgcn_readlane((int)ri, 63);
	}
	if (lane == 0) {
		const u64 total = 4u * (N / 32u + 1u) + S;
		const bool ok = total <= bt.out_cap[u];
		d_out_len[u] = ok ? total : 0;
		d_status[u] = ok ? 0 : -5;
	}
}

__global__ __launch_bounds__(1024) void xe3_emit_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, S16 mlen3,
                                                       S16 moff, Xe3 x, uint8_t* __restrict__ d_out)
{
	__shared__ uint32_t s_nr[1024], s_sr[1024];
	__shared__ uint16_t s_in_off[16][256];
	__shared__ uint16_t s_in_len[16][256];
	__shared__ uint8_t  s_in_byte[16][256];
	__shared__ u64      s_in_mask[16][8];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
	const uint32_t lc = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	const uint32_t k = lc - bt.chunk_prefix[u];
	const u64 n = bt.in_len[u];
	const u64 cap = bt.out_cap[u];
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	uint8_t* __restrict__ out = d_out + bt.out_off[u];
	const u64 mbase = (u64)bt.chunk_prefix[u] * 65536u;
	const u64 gwu = mbase >> 6;
	const u64 nwin = (n + 63u) >> 6;
	const u64 sb = (u64)k * 1024u;
	const uint32_t nsb = sb >= nwin ? 0u : ((nwin - sb < 1024u) ? (uint32_t)(nwin - sb) : 1024u);
	const u64 Nsb = x.sbpre[(u64)lc * 3u], Ssb = x.sbpre[(u64)lc * 3u + 1u];
	const uint32_t par = (uint32_t)(x.sbpre[(u64)lc * 3u + 2u] & 1u);     // rank parity of the long matches at the super-block's entry
	for (uint32_t w = tid; w < nsb; w += 1024u) {
		const uint32_t v = x.wnr[gwu + sb + w];
		s_nr[w] = v ^ (par << 31);
		s_sr[w] = par ? x.ws1[gwu + sb + w] : x.ws0[gwu + sb + w];
	}
	__syncthreads();
	const uint32_t w0 = wv * 64u, w1 = (w0 + 64u < nsb) ? w0 + 64u : nsb;
	xe_emit_segment(lane, sb, w0, w1, n, cap, d, out, mlen3 + mbase, moff + mbase, x.wtok + gwu, x.wmat + gwu, x.wfar + gwu, s_nr, s_sr, Nsb, Ssb,
	                s_in_off[wv], s_in_len[wv], s_in_byte[wv], s_in_mask[wv], x.seam + ((u64)lc * 16u + wv) * 8u, x.seampos + ((u64)lc * 16u + wv) * 2u);
}

__global__ __launch_bounds__(64) void xe3_stitch_kernel(BatchTables bt, Xe3 x, uint8_t* __restrict__ d_out)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t u = blockIdx.x;
	const u64 cap = bt.out_cap[u];
	uint8_t* __restrict__ out = d_out + bt.out_off[u];
	const uint32_t nk = bt.chunk_prefix[u + 1] - bt.chunk_prefix[u];
	const u64 seg0 = (u64)bt.chunk_prefix[u] * 16u, nseg = (u64)nk * 16u;
	uint32_t g_acc = 0; u64 g_fpos = 0; bool g_pend = false; u64 g_pend_pos = 0; uint32_t g_pend_low = 0;
	// 64 segments (4 super-blocks) per step, lane = segment. A segment's word-in-progress state after it is either a
	// reset (it completed the open word and / or opened one of its own) or a pass-through that only ORs a few bits in
	// (fewer than 32 tokens in 64 windows); likewise its pending-nibble state is a reset when it has long matches. With no
	// OR-ing pass-through in the step every seam only needs the nearest reset before it: one shuffle. Otherwise: serial.
	for (u64 sgb = 0; sgb < nseg; sgb += 64u) {
		const bool in = sgb + lane < nseg;
		uint32_t r[8]; u64 rp0 = 0, rp1 = 0;
		#pragma unroll
		for (int i = 0; i < 8; ++i) { r[i] = in ? x.seam[(seg0 + sgb + lane) * 8u + i] : 0u; }
		if (in) { rp0 = x.seampos[(seg0 + sgb + lane) * 2u]; rp1 = x.seampos[(seg0 + sgb + lane) * 2u + 1u]; }
		const bool nonempty = in && r[7] != 0;
		const bool wreset = nonempty && (r[1] != 0 || r[3] != 0);
		const bool wpass = nonempty && !wreset;
		const bool nreset = nonempty && r[4] != 0xFFu;
		if (__ballot(wpass && r[0] != 0) == 0) {
			const u64 below = (((u64)1) << lane) - 1u;
			// word state in front of me
			const u64 wm = __ballot(wreset) & below;
			const uint32_t wp = wm ? 63u - (uint32_t)__builtin_clzll(wm) : 0u;
			const uint32_t a_after = r[3] ? r[2] : 0u;                 // my state after me if I am a reset
			uint32_t acc_in = (uint32_t)__shfl((int)a_after, (int)wp, 64);
			u64 fpos_in = ((u64)(uint32_t)__shfl((int)(uint32_t)(rp0 >> 32), (int)wp, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)rp0, (int)wp, 64);
			const uint32_t pv = (uint32_t)__shfl((int)r[3], (int)wp, 64);    // did that reset open a word (else the slot is not needed: acc is 0 and the next word opens later)
			if (!wm) { acc_in = g_acc; fpos_in = g_fpos; } else if (!pv) { acc_in = 0; }
			if (nonempty && r[1]) { xe_store32(out, cap, fpos_in, __builtin_bitreverse32(acc_in | r[0]), true); }
			// pending nibble in front of me
			const u64 nm = __ballot(nreset) & below;
			const uint32_t np = nm ? 63u - (uint32_t)__builtin_clzll(nm) : 0u;
			uint32_t pend_in = (uint32_t)__shfl((int)r[5], (int)np, 64), low_in = (uint32_t)__shfl((int)r[6], (int)np, 64);
			u64 ppos_in = ((u64)(uint32_t)__shfl((int)(uint32_t)(rp1 >> 32), (int)np, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)rp1, (int)np, 64);
			if (!nm) { pend_in = g_pend ? 1u : 0u; low_in = g_pend_low; ppos_in = g_pend_pos; }
			if (nonempty && (r[4] & 0x100u) && pend_in) { put8(out, cap, ppos_in, low_in | ((r[4] & 0xFu) << 4)); }
			// carry out of the step
			const u64 wall = __ballot(wreset);
			if (wall) {
				const uint32_t l = 63u - (uint32_t)__builtin_clzll(wall);
				const uint32_t v3 = (uint32_t)__builtin_amdgcn_readlane((int)r[3], (int)l);
				g_acc = v3 ? (uint32_t)__builtin_amdgcn_readlane((int)r[2], (int)l) : 0u;
				if (v3) { g_fpos = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rp0 >> 32), (int)l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rp0, (int)l); }
			}
			const u64 nall = __ballot(nreset);
			if (nall) {
				const uint32_t l = 63u - (uint32_t)__builtin_clzll(nall);
				g_pend = __builtin_amdgcn_readlane((int)r[5], (int)l) != 0;
				g_pend_low = (uint32_t)__builtin_amdgcn_readlane((int)r[6], (int)l);
				g_pend_pos = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rp1 >> 32), (int)l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rp1, (int)l);
			}
			continue;
		}
		for (uint32_t j = 0; j < 64u; ++j) {                       // serial: a word spans more than two segments here
			uint32_t q[8];
			#pragma unroll
			for (int i = 0; i < 8; ++i) { q[i] = (uint32_t)__builtin_amdgcn_readlane((int)r[i], (int)j); }
			const u64 p0 = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rp0 >> 32), (int)j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rp0, (int)j);
			const u64 p1 = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rp1 >> 32), (int)j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rp1, (int)j);
			if (q[7] == 0) { continue; }                             // no token starts in this segment
			if (q[1]) {                                              // the open word was completed inside segment j
				const uint32_t wdv = g_acc | q[0];
				if (lane == 0) { xe_store32(out, cap, g_fpos, __builtin_bitreverse32(wdv), true); }
				g_acc = 0;
			} else { g_acc |= q[0]; }
			if (q[3]) { g_acc = q[2]; g_fpos = p0; }
			const uint32_t ln = q[4];
			if (ln & 0x100u) { if (g_pend && lane == 0) { put8(out, cap, g_pend_pos, g_pend_low | ((ln & 0xFu) << 4)); } g_pend = false; }
			if (ln != 0xFFu) { g_pend = q[5] != 0; g_pend_pos = p1; g_pend_low = q[6]; }
		}
	}
	// final flag word (:343-344); size and status were written by xe3_fix_kernel
	if (lane == 0) {
		u64 Nall = 0, Sall = 0;
		if (nk) {                                                   // totals = before the last super-block + its own
			const uint32_t lcl = bt.chunk_prefix[u + 1] - 1u;
			Nall = x.sbpre[(u64)lcl * 3u] + x.sbtot[(u64)lcl * 4u];
			Sall = x.sbpre[(u64)lcl * 3u + 1u] + x.sbtot[(u64)lcl * 4u + 2u + (uint32_t)(x.sbpre[(u64)lcl * 3u + 2u] & 1u)];
		}
		const uint32_t cnt = (uint32_t)(Nall & 31u);
		const u64 total = 4u * (Nall / 32u + 1u) + Sall;
		uint32_t wvv; u64 fp;
		if (cnt) { wvv = __builtin_bitreverse32(g_acc) | ((1u << (32u - cnt)) - 1u); fp = g_fpos; }
		else { wvv = 0xFFFFFFFFu; fp = total - 4u; }
		put8(out, cap, fp, wvv); put8(out, cap, fp + 1u, wvv >> 8); put8(out, cap, fp + 2u, wvv >> 16); put8(out, cap, fp + 3u, wvv >> 24);
	}
}

static int g_xpress_emit_mode = 0;                               // 0 = by batch size, 1 = one wave per unit, 2 / 3 = 4 / 16 waves per unit, 4 = a block per super-block (tests)
// Which parse/emit kernels a batch gets. Few units, or long ones (on average more than four 64 KiB super-blocks each):
// a block per super-block, so that a long stream is spread over the GPU (needs 24 B of records per 64 input bytes).
// Otherwise the per-unit kernels: four waves per unit up to 1024 units, one wave per unit beyond (a full GPU is
// issue-bound and the multi-wave kernels spend more wave-cycles; at 3 239 units of 64 KiB the three variants are within
// 10 % of each other: 5.5 / 5.6 / 6.2 ms per pass).
int xpress_emit_mode_for(uint32_t n_units, uint32_t n_chunks)
{
	if (g_xpress_emit_mode) { return g_xpress_emit_mode; }
	if ((n_units <= 64u || n_chunks >= 4u * n_units) && n_chunks <= 32768u) { return 4; }
	// (round 6 sweep, units of 64 KiB spread over the corpus, ms per pass one wave / four waves per unit -- tools/dev/gpu_emit_sweep.py: 256 units 1.59 / 1.03,
	// 512: 1.72 / 1.19, 768: 2.44 / 2.13, 1024: 2.64 / 2.49, 1536: 3.02 / 3.16, 2048: 3.09 / 3.22, 3239: 4.05 / 4.76, 6478: 7.25 / 8.97 -- the crossover lies
	// between 1024 and 1536 units: four waves shorten a unit's serial chain while wave slots are free, one wave per unit wins once the CUs are full)
	return n_units <= 1024u ? 2 : 1;
}
void launch_xpress_emit(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint16_t* mlen3, const uint16_t* moff,
                        const XpressWinBufs& wb, uint8_t* d_out, u64* d_out_len, int32_t* d_status)
{
	if (bt.n_units == 0) { return; }
	int mode = xpress_emit_mode_for(bt.n_units, bt.n_chunks);
	if (mode == 4 && !wb.wecur) { mode = 2; }                    // (records not reserved: the plan was made under another forced mode)
	u64* wtok = wb.wtok; u64* wmat = wb.wmat; uint32_t* wfar = wb.wfar;
	if (mode == 1) { hipLaunchKernelGGL(xpress_emit_kernel, dim3(bt.n_units), dim3(64), 0, st, d_in, bt, mlen3, moff, d_out, d_out_len, d_status); }
	else if (mode == 2) { hipLaunchKernelGGL(xpress_emit2_kernel<4u>, dim3(bt.n_units), dim3(256), 0, st, d_in, bt, mlen3, moff, wtok, wmat, wfar, d_out, d_out_len, d_status); }
	else if (mode == 3) { hipLaunchKernelGGL(xpress_emit2_kernel<16u>, dim3(bt.n_units), dim3(1024), 0, st, d_in, bt, mlen3, moff, wtok, wmat, wfar, d_out, d_out_len, d_status); }
	else {
		Xe3 x = { wb.wtok, wb.wmat, wb.wfar, wb.wecur, wb.weF, wb.wsum, wb.wnr, wb.ws0, wb.ws1, wb.sbtot, wb.sbpre, wb.seam, wb.seampos };
		if (bt.n_chunks) { hipLaunchKernelGGL(xe3_walk_kernel, dim3(bt.n_chunks), dim3(1024), 0, st, d_in, bt, mlen3, moff, x); }
		if (bt.n_chunks) { hipLaunchKernelGGL(xe3_seam_kernel, dim3(bt.n_chunks), dim3(64), 0, st, d_in, bt, mlen3, moff, x, wb.used); }
		hipLaunchKernelGGL(xe3_fix_kernel, dim3(bt.n_units), dim3(64), 0, st, d_in, bt, mlen3, moff, x, wb.used, d_out_len, d_status);
		if (bt.n_chunks) { hipLaunchKernelGGL(xe3_emit_kernel, dim3(bt.n_chunks), dim3(1024), 0, st, d_in, bt, mlen3, moff, x, d_out); }
		hipLaunchKernelGGL(xe3_stitch_kernel, dim3(bt.n_units), dim3(64), 0, st, bt, x, d_out);
	}
}
void set_xpress_emit_mode(int mode) { g_xpress_emit_mode = mode; }

} // namespace msc

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

__global__ __launch_bounds__(64) void xpress_emit_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                        const uint16_t* __restrict__ mlen3, const uint16_t* __restrict__ moff,
                                                        uint8_t* __restrict__ d_out, u64* __restrict__ d_out_len, int32_t* __restrict__ d_status)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t u = blockIdx.x;
	const u64 n = bt.in_len[u];
	const u64 cap = bt.out_cap[u];
	const uint8_t* __restrict__ d = d_in + bt.in_off[u];
	uint8_t* __restrict__ out = d_out + bt.out_off[u];
	const u64 mbase = (u64)bt.chunk_prefix[u] * 65536u;           // this unit's slice of the per-position match arrays
	const u64 end2 = n >= 2u ? n - 2u : 0u;

	u64 cur = 0, F = 0, S = 0, N = 0, R = 0;                      // next token start, filled, sum sizes, tokens, long matches
	bool pend = false; u64 pend_pos = 0; uint32_t pend_low = 0;   // length nibble byte waiting for its high half
	uint32_t facc = 0; u64 fposc = 0;                             // flag word in progress (first token = bit 0) and its slot

	// Inputs are burst-loaded 8 windows (512 positions) at a time into a double-buffered LDS stage: one wait on global
	// memory per 512 positions instead of one per window (loads are unconditional with a clamped index).
	__shared__ uint16_t s_in_off[2][512];
	__shared__ uint16_t s_in_len[2][512];
	__shared__ uint8_t  s_in_byte[2][512];
	uint32_t g_off[8], g_len[8], g_byte[8];
#define XE_BURST_LOAD(gbase) { _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) { \
		const u64 q_ = (gbase) + (u64)k_ * 64u + lane; const u64 c_ = q_ < n ? q_ : n - 1u; \
		g_off[k_] = moff[mbase + c_]; g_len[k_] = mlen3[mbase + c_]; g_byte[k_] = d[c_]; } }
#define XE_BURST_STORE(buf) { _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) { \
		s_in_off[buf][k_ * 64 + lane] = (uint16_t)g_off[k_]; s_in_len[buf][k_ * 64 + lane] = (uint16_t)g_len[k_]; s_in_byte[buf][k_ * 64 + lane] = (uint8_t)g_byte[k_]; } }
	if (n) { XE_BURST_LOAD((u64)0) }
#ifdef XE_PROFILE
	unsigned long long xe_acc[6] = {0, 0, 0, 0, 0, 0}, xe_prev = __builtin_readcyclecounter();
#endif
	for (u64 wbase = 0; wbase < n; wbase += 64u) {
		XE_T(5)
		const uint32_t wi = (uint32_t)((wbase >> 6) & 7u), buf = (uint32_t)((wbase >> 9) & 1u);
		if (wi == 0) {                                            // group start: publish this group's inputs, start loading the next
			XE_BURST_STORE(buf)
			if (wbase + 512u < n) { XE_BURST_LOAD(wbase + 512u) }
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
		}
		const u64 wend = (wbase + 64u < n) ? wbase + 64u : n;
		const u64 p = wbase + lane;
		const bool inr = p < n;
		if (cur >= wend) { continue; }                            // window wholly covered by a match
		uint32_t off = inr ? (uint32_t)s_in_off[buf][wi * 64u + lane] : 0u, L = s_in_len[buf][wi * 64u + lane];
		const uint32_t byte = s_in_byte[buf][wi * 64u + lane];
		const u64 mm = __ballot(inr && off != 0 && p >= cur);
		XE_T(0)
		// The serial loop only decides which candidates are TAKEN (incl. the lagging-fill rule); the token mask is derived
		// in parallel afterwards.
		u64 matchmask = 0;
		const u64 cur_entry = cur;
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
		while (cur < wend) {
			if (cur < end2) {
				if (F <= cur) { F = (F + 0x2000u < end2) ? F + 0x2000u : end2; }          // this token's lazy Fill (:269)
				if (cur >= F) { ++cur; continue; }                                         // lagging fill => literal
			}
			const uint32_t rel = (uint32_t)(cur - wbase);
			const u64 rest = (mm >> rel);
			if (rest == 0) {
				const u64 lastp = (wend - 1u < end2) ? wend - 1u : end2;              // fills done by the literal tokens up to wend
				if (end2 && F <= lastp && lastp < end2) { F = (F + 0x2000u < end2) ? F + 0x2000u : end2; }
				cur = wend;
				break;
			}
			const uint32_t mp = rel + ctz64(rest);
			const u64 pm = wbase + mp;
			if (F <= pm) { F = (F + 0x2000u < end2) ? F + 0x2000u : end2; }            // fill by a token in (cur, pm]
			matchmask |= ((u64)1) << mp;
			u64 Lm = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)mp);
			if (Lm == 45u) {                                      // the finder capped this match at 48: extend it
				const u64 x = pm - (uint32_t)__builtin_amdgcn_readlane((int)off, (int)mp);
				const u64 lim = n - pm - 1u;                        // never count the buffer's final byte
				Lm = 45u + wave_extend(d, x + 48u, pm + 48u, lim - 48u, n, lane);
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

		// ---- emit ----------------------------------------------------------------------------------------------
		const uint32_t nt = (uint32_t)__popcll(tokmask);
		const bool lng = is_m && L >= 7u;
		const u64 longmask = __ballot(lng);
		const bool even = !(((uint32_t)R + popc_below(longmask)) & 1u);
		uint32_t sz = 0;
		if (is_tok) { sz = !is_m ? 1u : 2u + (uint32_t)(lng && even) + (uint32_t)(L >= 22u) + (L >= 277u ? (L <= 0xFFFFu ? 2u : 6u) : 0u); }
		const uint32_t incl = wave_incl_scan_add(sz);
		const uint32_t wsum = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		const uint32_t tb = popc_below(tokmask);
		const uint32_t sh = (uint32_t)(N & 31u);                   // tokens already in the flag word in progress
		const uint32_t tq = sh + tb;                               // my token's index counted from that word's first token
		// every position of this window is base + a 32-bit offset: pos(t) = 4*(t div 32 + 1) + sum size(u<t)
		const u64 base = 4u * (N / 32u + 1u) + S;
		const uint32_t posrel = 4u * (tq >> 5) + (incl - sz);
		const uint32_t kdone = (sh + nt) >> 5;                     // flag words completed by this window (0..2)
		const bool fits = base + 4u * kdone + wsum <= cap;         // uniform: no store of this window can pass the capacity
		// the nibble of the NEXT long match in this window (it shares my byte when my rank is even)
		const u64 above = (longmask >> lane) >> 1;
		const uint32_t nib = L >= 7u ? (L - 7u < 15u ? L - 7u : 15u) : 0u;
		const uint32_t partner = above ? lane + 1u + ctz64(above) : lane;
		const uint32_t pnib = (uint32_t)__shfl((int)nib, (int)partner, 64);
		if (fits) { xe_emit_tokens<false>(out, cap, base, posrel, is_tok, is_m, byte, off, L, lng, even, above != 0, nib, pnib, pend, pend_pos, pend_low, longmask, lane); }
		else      { xe_emit_tokens<true >(out, cap, base, posrel, is_tok, is_m, byte, off, L, lng, even, above != 0, nib, pnib, pend, pend_pos, pend_low, longmask, lane); }
		XE_T(2)
		// ---- flag words: the match bits in TOKEN order are a ballot after moving every token's bit to lane = its rank
		// (tokens to [0,nt), the other lanes behind them: a permutation). The word in progress lives in scalars, first
		// token in bit 0; a completed word is bit-reversed into its slot (the first token is the MSB, :317-324). ----------
		{
			const uint32_t dst = is_tok ? tb : nt + (lane - tb);
			const uint32_t fm = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (is_tok && is_m) ? 1 : 0);
			const u64 M = __ballot(fm != 0);
			u64 sm = __ballot(is_tok && (tq & 31u) == 0);             // tokens that open a flag word: its slot is the 4 bytes before them
			const u64 lo = (u64)facc | (M << sh);
			const uint32_t hi = sh ? (uint32_t)(M >> (64u - sh)) : 0u;
			u64 fp = fposc;
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
			const uint32_t ll = 63u - (uint32_t)__builtin_clzll(longmask);        // last long match of the window
			const u64 rl = R + (uint32_t)__popcll(longmask) - 1u;
			pend = !(rl & 1u);
			if (pend) {
				pend_pos = base + (uint32_t)__builtin_amdgcn_readlane((int)posrel, (int)ll) + 2u;
				pend_low = (uint32_t)__builtin_amdgcn_readlane((int)nib, (int)ll);
			}
			R += (uint32_t)__popcll(longmask);
		}
		N += nt;
		S += wsum;
		XE_T(3)
	}
#ifdef XE_PROFILE
	if (lane == 0) { for (int i_ = 0; i_ < 6; ++i_) { atomicAdd(&g_xe_prof[i_], xe_acc[i_]); } }
#endif

	// ---- final flag word (:343-344), size, status ----------------------------------------------------------------
	const u64 gf = N / 32u;
	const uint32_t cnt = (uint32_t)(N & 31u);
	const u64 total = 4u * (gf + 1u) + S;
	if (lane == 0) {
		uint32_t wv; u64 fp;
		if (cnt) { wv = __builtin_bitreverse32(facc) | ((1u << (32u - cnt)) - 1u); fp = fposc; }
		else { wv = 0xFFFFFFFFu; fp = total - 4u; }
		put8(out, cap, fp, wv); put8(out, cap, fp + 1u, wv >> 8); put8(out, cap, fp + 2u, wv >> 16); put8(out, cap, fp + 3u, wv >> 24);
		const bool ok = total <= cap;
		d_out_len[u] = ok ? total : 0;
		d_status[u] = ok ? 0 : -5;
	}
}

void launch_xpress_emit(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* mlen3, const uint16_t* moff,
                        uint8_t* d_out, u64* d_out_len, int32_t* d_status)
{
	if (bt.n_units == 0) { return; }
	hipLaunchKernelGGL(xpress_emit_kernel, dim3(bt.n_units), dim3(64), 0, st, d_in, bt, mlen3, moff, d_out, d_out_len, d_status);
}

} // namespace msc

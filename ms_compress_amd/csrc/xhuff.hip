// xhuff.hip -- Xpress+Huffman chunk pipeline for gfx950, bit-exact with the reference CPU encoder.
//
// Replaces, per 64 KiB chunk of a unit (/root/reference/src/xpress_huff_compress.cpp:247-331):
//   xh_parse_kernel    xh_compress_lz77 (:52-153): greedy tokens over the matches of xp_find_kernel (0xFFFF window
//                      reaching into the previous chunk), length clipped to the chunk (:93), 512-symbol histogram
//                      (+ EOS 0x100 in the unit's last chunk, :127-144). One wavefront per chunk; scalar walk over
//                      ballot masks; histogram by LDS atomics. The reference's private intermediate byte buffer is not
//                      reproduced: tokens stay as (token-start bit mask, per-position len-3 / offset).
//   xh_huff_kernel     HuffmanEncoder<15,512>::CreateCodes (/root/reference/include/mscomp/HuffmanEncoder.h:58-127):
//                      the bzip2-style heap is simulated operation for operation (tie-breaking defines the lengths) in
//                      LDS, every sift by the whole wave in one round trip, depths / size test (xh_calc_compressed_len :181-188) / canonical codes (:109-123)
//                      by the whole wave; flags chunks that need the fallback (:274, :310).
//   xh_fallback_kernel xh_compress_no_matching + CreateCodesSlow (:155-180, HuffmanEncoder.h:129-226): literals only,
//                      package-merge lengths (packages = per-symbol multiplicity vectors in an LDS pool).
//   xh_encode_kernel   xh_compress_encode + OutputBitstream (:195-245, Bitstream.h:111-148): every bit/byte position is
//                      a prefix sum: a WriteBits call that raises flushes(T)=ceil(T/16)-1 to f allocates
//                      the slot of word f+1 at 4+2(f-1)+R; raw length bytes sit at 4+2*flushes+R. Bits are OR-ed into a
//                      256-word LDS ring, completed words leave to their slots every window.
#include "common.h"
#include "kernels.h"

namespace msc {

__device__ __forceinline__ uint32_t xh_incl_scan_add(uint32_t v) { return wave_incl_scan_add_u32(v); }
__device__ __forceinline__ uint32_t xh_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t xh_wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)xh_incl_scan_add(v), 63); }

__device__ __forceinline__ uint32_t xh_ldg32(const uint8_t* __restrict__ d, u64 pos, u64 n)
{
	if (pos + 4u <= n) { return ld32(d + pos); }
	uint32_t v = 0;
	for (uint32_t k = 0; k < 4u && pos + k < n; ++k) { v |= (uint32_t)d[pos + k] << (8u * k); }
	return v;
}
// equal bytes of d[a..] and d[b..] (a < b), at most maxadd; 256 bytes per wave step
__device__ __forceinline__ uint32_t xh_extend(const uint8_t* __restrict__ d, u64 a, u64 b, uint32_t maxadd, u64 n, uint32_t lane)
{
	uint32_t done = 0;
	while (done < maxadd) {
		const uint32_t off = done + 4u * lane;
		uint32_t x = 0xFFFFFFFFu;
		if (off < maxadd) { x = xh_ldg32(d, a + off, n) ^ xh_ldg32(d, b + off, n); }
		const u64 mis = __ballot(x != 0);
		if (mis) {
			const uint32_t l = ctz64(mis);
			const uint32_t xl = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)l);
			const uint32_t r = done + 4u * l + ((uint32_t)__builtin_ctz(xl) >> 3);
			return r < maxadd ? r : maxadd;
		}
		done += 256u;
	}
	return maxadd;
}

struct ChunkGeom { uint32_t u, k, cn; u64 n, cbase; bool last; };
__device__ __forceinline__ ChunkGeom chunk_geom(const BatchTables& bt, uint32_t lc)
{
	ChunkGeom g;
	g.u = unit_of_chunk(bt.chunk_prefix, bt.n_units, lc);
	g.k = lc - bt.chunk_prefix[g.u];
	g.n = bt.in_len[g.u];
	g.cbase = (u64)g.k * 65536u;
	g.cn = (g.n - g.cbase < 65536u) ? (uint32_t)(g.n - g.cbase) : 65536u;
	g.last = (lc + 1u == bt.chunk_prefix[g.u + 1]);
	return g;
}

// ===================================================================================================================
// parse: tokens + histogram
// ===================================================================================================================
// One window of the greedy parse (xh_compress_lz77, xpress_huff_compress.cpp:60-126): positions [wbase, wbase+64) of the
// chunk, entered at `cur`; off / L are this lane's candidate (0 = none; L = len-3 as capped by the finder). Writes the
// window's token mask and the final lengths of the taken matches; returns the parse position after the window.
__device__ __forceinline__ uint32_t xh_parse_window(const ChunkGeom& g, const uint8_t* __restrict__ d, uint32_t lane, uint32_t wbase,
                                                    uint32_t cur, uint32_t off, uint32_t L, S16 mlen3c, u64* __restrict__ tokc,
                                                    u64& tokmask_out, uint32_t& L_out)
{
	const uint32_t wend = (wbase + 64u < g.cn) ? wbase + 64u : g.cn;
	if (cur >= wend) { if (lane == 0) { tokc[wbase >> 6] = 0; } tokmask_out = 0; L_out = L; return cur; }      // window wholly covered by a match
	const uint32_t o = wbase + lane;
	const bool inr = o < g.cn;
	if (!inr) { off = 0; }
	const u64 mm = __ballot(inr && off != 0 && o >= cur);
	// the serial loop only decides which candidates are TAKEN; the token mask is derived in parallel afterwards
	u64 matchmask = 0;
	const uint32_t entry = cur;
	const uint32_t wn = wend - wbase;
	// Every candidate lane clips its length to the chunk end (:93) and precomputes where the walk goes after taking it:
	// the first candidate at or after its end (relative to the window; >= wn leaves the window). The scalar loop is
	// then one v_readlane per taken match; candidates the finder capped at 48 are extended on demand.
	const uint32_t remL = g.cn - o;                              // bytes left in the chunk (>= 3 for a candidate)
	const uint32_t Lin = L;                                      // what the finder's word holds
	const bool cappedL = (L == 45u) && remL > 48u;
	{ const uint32_t lc_ = (L + 3u < remL) ? L + 3u : remL; if (inr && off != 0) { L = lc_ - 3u; } }
	const uint32_t nx = lane + L + 3u;
	const u64 restl = nx < 64u ? mm >> nx : (u64)0;
	const uint32_t J = nx >= wn ? nx : (restl ? nx + ctz64(restl) : wn);
	u64 capm = sgpr64(__ballot(cappedL) & mm);
	uint32_t mp;
	{
		const uint32_t rel = cur - wbase;
		const u64 rest = mm >> rel;
		mp = rest ? rel + ctz64(rest) : wn;
	}
	mp = (uint32_t)__builtin_amdgcn_readfirstlane((int)mp);
	const uint32_t wn_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wn);
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
		// capped by the finder: extend, at most to the chunk end
		matchmask |= ((u64)1) << mp;
		capm &= ~(((u64)1) << mp);
		const uint32_t rem = g.cn - (wbase + mp);
		const u64 P = g.cbase + wbase + mp;
		const u64 X = P - (uint32_t)__builtin_amdgcn_readlane((int)off, (int)mp);
		const u64 lim = g.n - P - 1u;                              // never count the buffer's final byte
		const uint32_t maxadd = (uint32_t)((lim < rem ? lim : rem) - 48u);
		uint32_t len = 48u + xh_extend(d, X + 48u, P + 48u, maxadd, g.n, lane);
		if (len > rem) { len = rem; }                              // :93
		if (lane == mp) { L = len - 3u; }
		const uint32_t nxs = mp + len;
		if (nxs >= wn_s) { mp = nxs; }
		else { const u64 rest = mm >> nxs; mp = rest ? nxs + ctz64(rest) : wn_s; }
	}
	const bool is_m = (matchmask >> lane) & (u64)1;
	const uint32_t mend = is_m ? lane + L + 3u : 0u;              // match end, relative to the window
	const uint32_t reach = wave_incl_scan_max(mend);
	const bool is_tok = o >= entry && inr && (is_m || reach <= lane);
	const u64 tokmask = __ballot(is_tok);
	if (is_m && L != Lin) { mlen3c[o] = (uint16_t)L; }           // only a clipped or extended length is written back (round 4 rewrote every taken match: each
	                                                             // scattered 2-byte store dirties a line the kernel has just read -- 13 GB of write-backs per pass)
	if (lane == 0) { tokc[wbase >> 6] = tokmask; }
	tokmask_out = tokmask; L_out = L;
	return wbase + mp;
}

// Adds (sign = +1) or removes (sign = -1) the tokens of one window from the symbol histogram and the raw-length-byte
// count. off / L / byte: this lane's position (L = final length - 3 when the position is a match token).
__device__ __forceinline__ int xh_count_window(u64 tokmask, uint32_t lane, uint32_t off, uint32_t L, uint32_t byte, uint32_t* s_cnt, uint32_t sign)
{
	int raw = 0;
	if ((tokmask >> lane) & (u64)1) {
		uint32_t sym = byte;
		if (off != 0) {
			const uint32_t ob = 31u - (uint32_t)__builtin_clz(off);
			sym = 0x100u | (ob << 4) | (L < 15u ? L : 15u);
			raw = L >= 270u ? 3 : (L >= 15u ? 1 : 0);
		}
		atomicAdd(&s_cnt[sym], sign);
	}
	return raw;
}

// Four waves per 64 KiB chunk. The parse is a serial walk whose only state is the position of the next token, so:
//   1. wave j parses the windows [256 j, 256 j + 256) SPECULATIVELY, as if a token started exactly at the first position of
//      its segment (wave 0's segment is exact), and records the parse position after every window;
//   2. wave 0 repairs the seams in order: it re-parses segment j from the true entry position until the position after
//      a window equals the recorded speculative one -- from there on both parses are identical (typically after one or
//      two windows; a segment that never re-synchronises is simply parsed again, serially);
//      The histogram and the raw-length-byte count are accumulated while parsing; a repaired window first takes its
//      speculative tokens back.
// A taken match's final length depends only on its position (chunk clipping, extension), so speculative writes of
// mlen3 are consistent with the final parse.
#define XH_SEGW 256u
#ifdef XP_PROFILE
__device__ unsigned long long g_xp_prof[8];
extern "C" void mscomp_amd_debug_xp_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xp_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xp_prof), z, 64); }
#define XP_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); if (tid == 0) { atomicAdd(&g_xp_prof[i], t_ - xp_prev); atomicMax(&g_xp_prof[4 + (i)], t_ - xp_prev); } xp_prev = t_; }
#else
#define XP_T(i)
#endif
__global__ __launch_bounds__(256) void xh_parse_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                      S16 mlen3, S16 moff,
                                                      u64* __restrict__ tokbits, uint32_t* __restrict__ counts, uint32_t* __restrict__ extra)
{
	__shared__ uint32_t s_cnt[512];
	__shared__ uint32_t s_end[1024];                               // parse position after every window
	__shared__ uint32_t s_xtra[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = xh_uniform(tid >> 6);
	const uint32_t lc = blockIdx.x;
	const ChunkGeom g = chunk_geom(bt, lc);
	const uint8_t* __restrict__ d = d_in + bt.in_off[g.u];
	const u64 gbase = (u64)lc * 65536u;
	S16 mlen3c = mlen3 + gbase;
	S16 moffc = moff + gbase;
	u64* __restrict__ tokc = tokbits + (u64)lc * 1024u;
	for (uint32_t i = tid; i < 512u; i += 256u) { s_cnt[i] = 0; }
	const uint32_t nwin = (g.cn + 63u) >> 6;
#ifdef XP_PROFILE
	unsigned long long xp_prev = __builtin_readcyclecounter();
#endif

	// ---- 1. speculative parse of my segment. Inputs are burst-loaded 4 windows (256 positions) at a time into a
	// double-buffered LDS stage (one wait on global memory per 256 positions; unconditional loads, clamped index).
	__shared__ uint16_t s_in_off[4][2][256];
	__shared__ uint16_t s_in_len[4][2][256];
	__shared__ uint8_t  s_in_byte[4][2][256];
	uint32_t g_off[4], g_len[4], g_byte[4];
	int xtra = 0;                                                  // raw length bytes of my tokens (per lane, reduced at the end)
#define XP_BURST_LOAD(gb) { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { \
		const uint32_t q_ = (gb) + (uint32_t)k_ * 64u + lane; const uint32_t c_ = q_ < g.cn ? q_ : g.cn - 1u; \
		{ const uint32_t w_ = mlen3c.word(c_); g_off[k_] = w_ >> 16; g_len[k_] = w_ & 0xFFFFu; } g_byte[k_] = d[g.cbase + c_]; } }
#define XP_BURST_STORE(buf) { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { \
		s_in_off[wv][buf][k_ * 64 + lane] = (uint16_t)g_off[k_]; s_in_len[wv][buf][k_ * 64 + lane] = (uint16_t)g_len[k_]; s_in_byte[wv][buf][k_ * 64 + lane] = (uint8_t)g_byte[k_]; } }
	{
		const uint32_t w0 = wv * XH_SEGW, w1 = (w0 + XH_SEGW < nwin) ? w0 + XH_SEGW : nwin;
		uint32_t cur = w0 * 64u;
		if (w0 < w1) { XP_BURST_LOAD(w0 * 64u) }
		for (uint32_t w = w0; w < w1; ++w) {
			const uint32_t wi = (w - w0) & 3u, buf = ((w - w0) >> 2) & 1u;
			if (wi == 0) {
				XP_BURST_STORE(buf)
				if (w + 4u < w1) { XP_BURST_LOAD((w + 4u) * 64u) }
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
			}
			const uint32_t off = s_in_off[wv][buf][wi * 64u + lane];
			u64 tm; uint32_t Lf;
			cur = xh_parse_window(g, d, lane, w * 64u, cur, off, s_in_len[wv][buf][wi * 64u + lane], mlen3c, tokc, tm, Lf);
			if (lane == 0) { s_end[w] = cur; }
			xtra += xh_count_window(tm, lane, off, Lf, s_in_byte[wv][buf][wi * 64u + lane], s_cnt, 1u);
		}
	}
#undef XP_BURST_LOAD
#undef XP_BURST_STORE
	__syncthreads();
	XP_T(0)
	// ---- 2. repair the seams (wave 0) ------------------------------------------------------------------------------
	if (wv == 0) {
		for (uint32_t j = 1; j * XH_SEGW < nwin; ++j) {
			const uint32_t w0 = j * XH_SEGW, w1 = (w0 + XH_SEGW < nwin) ? w0 + XH_SEGW : nwin;
			uint32_t cur = xh_uniform(s_end[w0 - 1u]);                 // true entry of segment j
			if (cur == w0 * 64u) { continue; }                         // the speculation was right
			for (uint32_t w = w0; w < w1; ++w) {
				const uint32_t q = w * 64u + lane, c = q < g.cn ? q : g.cn - 1u;
				// (a length already finalised by the speculative pass is a fixed point of the clipping / extension)
				// (written by another wave a moment ago: read past this CU's L1)
				const u64 old_tm = __hip_atomic_load(&tokc[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const uint32_t off = moffc[c], Lold = __hip_atomic_load(&mlen3c[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), byte = d[g.cbase + c];
				xtra -= xh_count_window(old_tm, lane, off, Lold, byte, s_cnt, 0xFFFFFFFFu);      // take the speculative tokens back
				u64 tm; uint32_t Lf;
				cur = xh_parse_window(g, d, lane, w * 64u, cur, off, Lold, mlen3c, tokc, tm, Lf);
				xtra += xh_count_window(tm, lane, off, Lf, byte, s_cnt, 1u);
				const uint32_t spec = xh_uniform(s_end[w]);
				if (lane == 0) { s_end[w] = cur; }
				if (cur == spec) { break; }                               // re-synchronised: the rest of the segment stands
			}
		}
	}
	__syncthreads();
	XP_T(1)
	// ---- 3. reduce ------------------------------------------------------------------------------------------------
	const uint32_t xsum = xh_wave_sum((uint32_t)xtra);              // (two's complement: the corrections of wave 0 may be negative)
	if (lane == 0) { s_xtra[wv] = xsum; }
	__syncthreads();
	if (g.last && tid == 0) { s_cnt[0x100] += 1u; }                // EOS (:127-144)
	__syncthreads();
	for (uint32_t i = tid; i < 512u; i += 256u) { counts[(u64)lc * 512u + i] = s_cnt[i]; }
	if (tid == 0) { extra[lc] = s_xtra[0] + s_xtra[1] + s_xtra[2] + s_xtra[3]; }
	XP_T(2)
}


// ===================================================================================================================
// Huffman lengths (heap), size, canonical codes
// ===================================================================================================================
struct HuffLds {
	__attribute__((aligned(16))) uint2 heap[516]; // entry = {weight, node}; weight = (count<<8)|depth; heap[0] = {0,0}; every entry behind the last one is {0xFFFFFFFF,0}
	uint32_t wleaf[512];   // leaf weights (kept for the >15-bit rescale loop)
	uint16_t parent[1024];
	uint32_t cnt[512];
	uint8_t  lens[512];
	uint16_t codes[512];
	uint32_t flag;
};

// What xh_huff_kernel needs of it: the counts stay in registers (8 per lane) and the results (lens, codes) take the heap's
// place once the tree is built, the leaf weights are 8 registers per lane -- 6.2 KiB instead of 11.8 KiB: 24 chunks in flight per
// CU instead of 13.
struct HuffLdsFast {
	union {
		__attribute__((aligned(16))) uint2 heap[516];
		struct { __attribute__((aligned(16))) uint8_t lens[512]; __attribute__((aligned(16))) uint16_t codes[512]; };
	};
	uint16_t parent[1024];
};

// HEAP_PUSH / HEAP_POP (HuffmanEncoder.h:31-55) by the WHOLE wave, one LDS round trip per sift instead of one per level.
// The reference's order of operations (and with it every tie-break) is kept, because both sifts move along a path that does
// not depend on the item being sifted:
//   push at slot j: the path is j's ancestors j>>1, j>>2, ...; lane l reads ancestor l, the item passes every ancestor with a
//       larger weight (`weights[x] < weights[heap[j>>1]]`; the weights of the ancestors fall towards the root, so these
//       are the first m of them), lanes 1..m move their ancestor one level down, lane 0 drops the item at j>>m.
//   pop: the last entry t goes to the root and sinks along the "smaller child" path (right child only if STRICTLY
//       smaller, :48), which is a property of the heap alone: every lane compares the two children of 4 nodes, four ballots
//       give the choice at every inner node, the scalar unit follows the 8 choices down to the leaf level F; the path is
//       F>>8, F>>7, ..., F. Lane l reads path node l; t passes every node whose weight is not larger
//       (`weights[t] < weights[heap[j]]` stops, :49; the weights grow along the path), lanes 1..m-1 move their node one
//       level up, lane 0 drops t at path node m-1.
// Entries behind the last one hold the largest weight, so "no child" needs no index test: such a child never wins a
// comparison against a real entry and always stops the sift. Keys are compared on the weight only, exactly like the reference.
// The DS instructions of a wave execute in program order, so "read t, overwrite its slot, read the children" needs no wait in
// between -- only the compiler has to keep the order (HH_ORDER: no instruction, a scheduling fence).
// (One lane replaying the heap level by level: 3.1 ms for the 3 239 chunks of the bench corpus; this: 1.4 ms, bound by
// instruction issue -- 13 single-purpose waves per CU at the time, 24 now, ~200 instructions per merge step.)
#define HH_SENT 0xFFFFFFFFu
#define HH_ORDER() asm volatile("" ::: "memory")
// Round 5: the kernel is bound by its SCALAR instructions (a CU has one scalar unit for all its waves; it ran at 0.66 instructions per cycle,
// profiles/r04_sq_counters.json), so the two operations are written to need few of them (HH_DIET; 0 = the round-4 code): the eight choices
// of the path are followed with s_bitcmp1_b64 + s_addc_u32 (i = 2 i + bit: 2 instructions per level, the compiler's shift / and / shift / or
// took 4), the path reads are unconditional (lanes outside 1..8 compute slot 0 or 1: no exec masking) and the two predicated stores of an
// operation are one. HH_DIET = 2: the follow by the vector unit instead (every lane computes the same path; 23 vector instructions, no scalar one).
#ifndef HH_DIET
#define HH_DIET 1
#endif
template <class H> __device__ __forceinline__ void hh_push(H& h, uint32_t hl_new, uint2 e, uint32_t lane)     // hl_new = slot of the new entry (heap length after the push)
{
#if HH_DIET
	const uint2 par = h.heap[hl_new >> (lane & 31u)];              // lane l: ancestor l (0 = the sentinel in front of the heap, weight 0; lanes 10.. read slot 0 or a real slot: unused)
	const u64 up = __ballot(lane - 1u < 9u && e.x < par.x) >> 1;
	const uint32_t m = (uint32_t)__builtin_ctzll(~up);             // ancestors passed (heap[0] has weight 0: never passed)
	// lanes 1..m move their ancestor one level down, lane 0 drops the item at hl_new >> m: one store
	if (lane <= m) { h.heap[hl_new >> (lane ? lane - 1u : m)] = lane ? par : e; }
	HH_ORDER();
#else
	const uint32_t a = hl_new >> lane;                             // lane l: ancestor l (0 = the sentinel in front of the heap, weight 0)
	uint2 par = make_uint2(0u, 0u);
	if (lane >= 1u && lane <= 9u) { par = h.heap[a]; }
	const u64 up = __ballot(lane >= 1u && lane <= 9u && e.x < par.x) >> 1;
	const uint32_t m = (uint32_t)__builtin_ctzll(~up);             // ancestors passed (heap[0] has weight 0: never passed)
	if (lane >= 1u && lane <= m) { h.heap[hl_new >> (lane - 1u)] = par; }
	if (lane == 0u) { h.heap[hl_new >> m] = e; }
	HH_ORDER();
#endif
}
template <int NQ, class H> __device__ __forceinline__ void hh_masks(H& h, uint32_t lane, u64 (&mask)[4])
{
	uint4 ch[NQ];
	#pragma unroll
	for (uint32_t q = 0; q < (uint32_t)NQ; ++q) { ch[q] = *reinterpret_cast<const uint4*>(&h.heap[2u * (64u * q + lane)]); }
	#pragma unroll
	for (uint32_t q = 0; q < (uint32_t)NQ; ++q) { mask[q] = __ballot(ch[q].z < ch[q].x); }
}
template <class H> __device__ __forceinline__ uint2 hh_pop(H& h, uint32_t hl_old, uint32_t lane)               // hl_old = heap length before the pop
{
	const uint2 top = h.heap[1], t = h.heap[hl_old];               // (all lanes read the same two entries: broadcast)
	HH_ORDER();
	if (lane == 0u) { h.heap[hl_old] = make_uint2(HH_SENT, 0u); }
	HH_ORDER();
	// the smaller child of every inner node: node i = 64 q + lane reads its children 2 i, 2 i + 1 (one 16-byte aligned pair)
	u64 mask[4] = { 0, 0, 0, 0 };
#if HH_DIET
	// ... of every inner node that HAS children: node i has one iff 2 i <= the heap's length, so the nodes 64 q .. 64 q + 63 are only looked at
	// while the heap holds at least 128 q entries (it shrinks from 512 to 1: 2.5 of the 4 KiB-reads per pop on average; the LDS pipe is what
	// the kernel waits for once its scalar instructions are fewer)
	if (hl_old >= 384u) { hh_masks<4>(h, lane, mask); } else if (hl_old >= 256u) { hh_masks<3>(h, lane, mask); }
	else if (hl_old >= 128u) { hh_masks<2>(h, lane, mask); } else { hh_masks<1>(h, lane, mask); }
#else
	hh_masks<4>(h, lane, mask);
#endif
#if HH_DIET == 2
	// the follow on the VECTOR unit (every lane computes the same path; no scalar instruction at all): 32-bit halves of the masks, bit i mod 32
	uint32_t F;                                                    // -> 256..511: the path's node on the leaf level
	asm volatile("v_mov_b32 %0, 1" : "=v"(F));                     // (opaque: keeps the chain in vector registers)
	{
		const uint32_t m0l = (uint32_t)mask[0], m0h = (uint32_t)(mask[0] >> 32), m1l = (uint32_t)mask[1], m1h = (uint32_t)(mask[1] >> 32);
		const uint32_t m2l = (uint32_t)mask[2], m2h = (uint32_t)(mask[2] >> 32), m3l = (uint32_t)mask[3], m3h = (uint32_t)(mask[3] >> 32);
		#pragma unroll
		for (int s = 0; s < 5; ++s) { F = (F << 1) | ((m0l >> (F & 31u)) & 1u); }           // levels 0..4: nodes 1..31
		F = (F << 1) | ((m0h >> (F & 31u)) & 1u);                                          // level 5: nodes 32..63
		{ const uint32_t w = (F & 32u) ? m1h : m1l; F = (F << 1) | ((w >> (F & 31u)) & 1u); }   // level 6: nodes 64..127
		{ const uint32_t wa = (F & 32u) ? m2h : m2l, wb = (F & 32u) ? m3h : m3l; const uint32_t w = (F & 64u) ? wb : wa; F = (F << 1) | ((w >> (F & 31u)) & 1u); }   // level 7: 128..255
	}
#elif HH_DIET
	uint32_t F = 1;                                                // -> 256..511: the path's node on the leaf level
	u64 mk_;                                                       // (scratch of the asm block: the mask word of level 7)
	asm volatile(
		"s_bitcmp1_b64 %[m0], %[i]\n\ts_addc_u32 %[i], %[i], %[i]\n\t"      // levels 0..5: nodes 1..63 (s_bitcmp1_b64 takes bit i mod 64)
		"s_bitcmp1_b64 %[m0], %[i]\n\ts_addc_u32 %[i], %[i], %[i]\n\t"
		"s_bitcmp1_b64 %[m0], %[i]\n\ts_addc_u32 %[i], %[i], %[i]\n\t"
		"s_bitcmp1_b64 %[m0], %[i]\n\ts_addc_u32 %[i], %[i], %[i]\n\t"
		"s_bitcmp1_b64 %[m0], %[i]\n\ts_addc_u32 %[i], %[i], %[i]\n\t"
		"s_bitcmp1_b64 %[m0], %[i]\n\ts_addc_u32 %[i], %[i], %[i]\n\t"
		"s_bitcmp1_b64 %[m1], %[i]\n\ts_addc_u32 %[i], %[i], %[i]\n\t"      // level 6: nodes 64..127
		"s_bitcmp1_b32 %[i], 6\n\ts_cselect_b64 %[mk], %[m3], %[m2]\n\t"    // level 7: nodes 128..191 | 192..255
		"s_bitcmp1_b64 %[mk], %[i]\n\ts_addc_u32 %[i], %[i], %[i]"
		: [i] "+s"(F), [mk] "=&s"(mk_)
		: [m0] "s"(mask[0]), [m1] "s"(mask[1]), [m2] "s"(mask[2]), [m3] "s"(mask[3])
		: "scc");
#endif
#if HH_DIET
	const uint2 k = h.heap[F >> ((8u - lane) & 31u)];              // lane l (1..8): path node l; the other lanes read slot 0 or 1 (unused)
	const u64 stop = __ballot(lane - 1u < 8u && t.x < k.x);
	const uint32_t m = stop ? (uint32_t)__builtin_ctzll(stop) : 9u;   // t comes to rest on level m - 1
	// lanes 1..m-1 move their node one level up, lane 0 drops t at path node m - 1: one store
	if (lane < m) { h.heap[F >> (9u - (lane ? lane : m))] = lane ? k : t; }
	HH_ORDER();
#else
	uint32_t i = 1;
	#pragma unroll
	for (int s = 0; s < 6; ++s) { i = 2u * i + (uint32_t)((mask[0] >> i) & 1u); }          // levels 0..5: nodes 1..63
	i = 2u * i + (uint32_t)((mask[1] >> (i - 64u)) & 1u);                                   // level 6: nodes 64..127
	{ const u64 mk = i < 192u ? mask[2] : mask[3]; i = 2u * i + (uint32_t)((mk >> (i & 63u)) & 1u); }   // level 7: nodes 128..255
	const uint32_t F = i;                                          // 256..511: the path's node on the leaf level
	uint2 k = make_uint2(HH_SENT, 0u);
	if (lane >= 1u && lane <= 8u) { k = h.heap[F >> (8u - lane)]; }
	const u64 stop = __ballot(lane >= 1u && lane <= 8u && t.x < k.x);
	const uint32_t m = stop ? (uint32_t)__builtin_ctzll(stop) : 9u;   // t comes to rest on level m - 1
	if (lane >= 1u && lane < m) { h.heap[F >> (9u - lane)] = k; }
	if (lane == 0u) { h.heap[F >> (9u - m)] = t; }
	HH_ORDER();
#endif
	return top;
}

// CreateCodes lengths -> h.lens (one wave). wl[k] = leaf weight of symbol lane + 64 k = max(count, 1) << 8 (:69), in registers.
template <class H> __device__ void huff_lengths_fast(H& h, uint32_t lane, uint32_t (&wl)[8])
{
	__syncthreads();
	for (;;) {
		for (uint32_t i = lane; i < 1024u; i += 64u) { h.parent[i] = 0; }
		for (uint32_t i = lane; i < 516u; i += 64u) { h.heap[i] = i ? make_uint2(HH_SENT, 0) : make_uint2(0, 0); }
		__syncthreads();
		#pragma unroll
		for (uint32_t k = 0; k < 8u; ++k) {                         // :74, symbols in order
			for (uint32_t j = 0; j < 64u; ++j) { hh_push(h, k * 64u + j + 1u, make_uint2((uint32_t)__builtin_amdgcn_readlane((int)wl[k], (int)j), k * 64u + j + 1u), lane); }
		}
		uint32_t hl = 512, nn = 512;
		while (hl > 1u) {                                           // :78-85
			const uint2 ea = hh_pop(h, hl, lane); --hl;
			const uint2 eb = hh_pop(h, hl, lane); --hl;
			const uint32_t wa = ea.x, wb = eb.x;
			const uint32_t da = wa & 0xFFu, db = wb & 0xFFu;
			++nn;
			if (lane == 0u) { h.parent[ea.y] = (uint16_t)nn; h.parent[eb.y] = (uint16_t)nn; }
			const uint32_t wn = ((wa & ~0xFFu) + (wb & ~0xFFu)) | (1u + (da > db ? da : db));
			++hl;
			hh_push(h, hl, make_uint2(wn, nn), lane);
		}
		__syncthreads();
		bool too_long = false;
		for (uint32_t i = lane + 1u; i <= 512u; i += 64u) {
			uint32_t depth = 0;
			for (uint32_t k = i; h.parent[k]; k = h.parent[k]) { ++depth; }
			h.lens[i - 1u] = (uint8_t)depth;
			too_long |= depth > 15u;
		}
		if (!__ballot(too_long)) { break; }
		__syncthreads();
		#pragma unroll
		for (uint32_t k = 0; k < 8u; ++k) { wl[k] = (1u + (wl[k] >> 9)) << 8; }                        // :100-105
		__syncthreads();
	}
	__syncthreads();
}

// canonical codes by (length, symbol) for the symbols with lens != 0 (HuffmanEncoder.h:109-123 / :214-222 agree)
template <class H> __device__ void huff_canonical(H& h, uint32_t lane)
{
	uint32_t mylen[8];
	#pragma unroll
	for (int k = 0; k < 8; ++k) { mylen[k] = h.lens[lane * 8u + k]; }
	uint32_t code = 0;
	for (uint32_t l = 1; l <= 15u; ++l) {
		uint32_t mine = 0;
		#pragma unroll
		for (int k = 0; k < 8; ++k) { mine += (mylen[k] == l); }
		const uint32_t incl = xh_incl_scan_add(mine);
		uint32_t c = code + incl - mine;
		#pragma unroll
		for (int k = 0; k < 8; ++k) { if (mylen[k] == l) { h.codes[lane * 8u + k] = (uint16_t)c++; } }
		code = (code + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63)) << 1;
	}
	#pragma unroll
	for (int k = 0; k < 8; ++k) { if (mylen[k] == 0) { h.codes[lane * 8u + k] = 0; } }
}

template <class H> __device__ __forceinline__ void huff_store(const H& h, uint32_t lane, uint8_t* __restrict__ lens_out, uint16_t* __restrict__ codes_out)
{
	reinterpret_cast<uint2*>(lens_out)[lane] = reinterpret_cast<const uint2*>(h.lens)[lane];
	reinterpret_cast<uint4*>(codes_out)[lane] = reinterpret_cast<const uint4*>(h.codes)[lane];
}

__global__ __launch_bounds__(64) void xh_huff_kernel(BatchTables bt, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ extra,
                                                    uint8_t* __restrict__ lens_out, uint16_t* __restrict__ codes_out,
                                                    uint32_t* __restrict__ chunk_size, uint32_t* __restrict__ fb_list, uint32_t* __restrict__ fb_count,
                                                    uint32_t* __restrict__ fbflag)
{
	__shared__ HuffLdsFast h;
	const uint32_t lane = threadIdx.x;
	const uint32_t lc = blockIdx.x;
	const ChunkGeom g = chunk_geom(bt, lc);
	uint32_t mycnt[8], wl[8];                                     // counts / leaf weights of the symbols lane + 64 k
	#pragma unroll
	for (uint32_t k = 0; k < 8u; ++k) { const uint32_t c = counts[(u64)lc * 512u + lane + 64u * k]; mycnt[k] = c; wl[k] = (c ? c : 1u) << 8; }
	huff_lengths_fast(h, lane, wl);
	// xh_calc_compressed_len (:181-188): 16 + sum (len + offset bits) * count, rounded to 16-bit words, + raw length bytes
	uint32_t bits = 0;
	#pragma unroll
	for (uint32_t k = 0; k < 8u; ++k) { const uint32_t s = lane + 64u * k; bits += ((uint32_t)h.lens[s] + (s >= 0x100u ? ((s >> 4) & 0xFu) : 0u)) * mycnt[k]; }
	bits = 16u + xh_wave_sum(bits);
	const uint32_t comp = (bits + 15u) / 16u * 2u + extra[lc];
	const uint32_t limit = g.last ? g.cn + 36u : 65538u;         // :310 / :274
	if (comp > limit) {
		if (lane == 0) { fb_list[atomicAdd(fb_count, 1u)] = lc; chunk_size[lc] = 0; fbflag[lc] = 1; }
		return;
	}
	huff_canonical(h, lane);
	__syncthreads();
	huff_store(h, lane, lens_out + (u64)lc * 512u, codes_out + (u64)lc * 512u);
	if (lane == 0) { chunk_size[lc] = 256u + comp; fbflag[lc] = 0; }
}

// Stage-level test hook: code lengths of HuffmanEncoder<15,512>::CreateCodes for histograms given directly (one wave each)
__global__ __launch_bounds__(64) void xh_huff_debug_kernel(const uint32_t* __restrict__ counts, uint8_t* __restrict__ lens_out)
{
	__shared__ HuffLdsFast h;
	const uint32_t lane = threadIdx.x;
	uint32_t wl[8];
	#pragma unroll
	for (uint32_t k = 0; k < 8u; ++k) { const uint32_t c = counts[(u64)blockIdx.x * 512u + lane + 64u * k]; wl[k] = (c ? c : 1u) << 8; }
	huff_lengths_fast(h, lane, wl);
	reinterpret_cast<uint2*>(lens_out + (u64)blockIdx.x * 512u)[lane] = reinterpret_cast<const uint2*>(h.lens)[lane];
}
void launch_xh_huff_debug(hipStream_t st, const uint32_t* counts, uint8_t* lens, uint32_t n)
{
	if (n) { hipLaunchKernelGGL(xh_huff_debug_kernel, dim3(n), dim3(64), 0, st, counts, lens); }
}

// ===================================================================================================================
// fallback: literals only + package-merge (CreateCodesSlow)
// ===================================================================================================================
// package pool in (dynamic) LDS: two generations of <= 256 packages, each a multiplicity vector over the 257
// literal/EOS symbols (stride 264) -> 132 KiB beside the 24 KiB of static LDS
#define XH_FB_STRIDE 264u
#define XH_FB_GEN_BYTES (256u * XH_FB_STRIDE)
#define XH_FB_POOL_BYTES (2u * XH_FB_GEN_BYTES)

#ifdef XF_PROFILE
__device__ unsigned long long g_fb_prof[8];
extern "C" void mscomp_amd_debug_fb_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fb_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fb_prof), z, 64); }
#define FB_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); if (tid == 0) { atomicAdd(&g_fb_prof[i], t_ - fb_prev); } fb_prev = t_; }
#else
#define FB_T(i)
#endif
__global__ __launch_bounds__(512) void xh_fallback_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                         const uint32_t* __restrict__ fb_list, const uint32_t* __restrict__ fb_count,
                                                         u64* __restrict__ tokbits,
                                                         uint8_t* __restrict__ lens_out, uint16_t* __restrict__ codes_out, uint32_t* __restrict__ chunk_size)
{
	__shared__ HuffLds h;
	__shared__ uint16_t s_leaf[512];
	__shared__ uint32_t s_lcnt[512];                               // counts of the sorted leaves
	__shared__ uint16_t s_item[1024];                              // merged sequence of a round: package index, or 0x8000 | leaf symbol
	__shared__ u64 s_pc[2][512];                                   // package counts, two generations
	__shared__ uint32_t s_n[4];
	const uint32_t tid = threadIdx.x;
	extern __shared__ __attribute__((aligned(16))) uint8_t fb_pool_lds[];
	uint8_t* const gen0 = fb_pool_lds;                             // two generations
	const uint32_t nfb = *fb_count;
	for (uint32_t it = blockIdx.x; it < nfb; it += gridDim.x) {
		const uint32_t lc = fb_list[it];
		const ChunkGeom g = chunk_geom(bt, lc);
		const uint8_t* __restrict__ d = d_in + bt.in_off[g.u] + g.cbase;
#ifdef XF_PROFILE
		unsigned long long fb_prev = __builtin_readcyclecounter(); if (tid == 0) { atomicAdd(&g_fb_prof[7], 1ull); }
#endif
		__syncthreads();
		h.cnt[tid] = 0;
		__syncthreads();
		for (uint32_t i0 = tid; i0 < g.cn; i0 += 512u * 16u) {                            // xh_compress_no_matching (:155-180)
			uint32_t by[16];                                                              // 16 loads in flight per thread
			#pragma unroll
			for (int j = 0; j < 16; ++j) { const uint32_t i = i0 + (uint32_t)j * 512u; by[j] = d[i < g.cn ? i : g.cn - 1u]; }
			#pragma unroll
			for (int j = 0; j < 16; ++j) { if (i0 + (uint32_t)j * 512u < g.cn) { atomicAdd(&h.cnt[by[j]], 1u); } }
		}
		for (uint32_t w = tid; w < 1024u; w += 512u) {                                   // every position is a literal token
			const uint32_t lo = w * 64u;
			tokbits[(u64)lc * 1024u + w] = lo >= g.cn ? 0 : (g.cn - lo >= 64u ? ~(u64)0 : ((((u64)1) << (g.cn - lo)) - 1u));
		}
		__syncthreads();
		if (tid == 0 && g.last) { h.cnt[0x100] += 1u; }
		__syncthreads();
		FB_T(0)
		// present symbols, stable-sorted by count (ties: symbol order)  -- rank = #{(count, sym) smaller}
		const uint32_t myc = h.cnt[tid];
		h.lens[tid] = myc ? 15 : 0;
		uint32_t rank = 0;
		if (myc) { for (uint32_t s = 0; s < 512u; ++s) { const uint32_t c = h.cnt[s]; rank += (c != 0) && (c < myc || (c == myc && s < tid)); } }
		if (tid == 0) { s_n[0] = 0; }
		__syncthreads();
		if (myc) { s_leaf[rank] = (uint16_t)tid; s_lcnt[rank] = myc; atomicAdd(&s_n[0], 1u); }
		__syncthreads();
		const uint32_t nleaf = s_n[0];
		FB_T(1)
		if (nleaf == 1) { if (myc) { h.lens[tid] = 1; } }
		else {
			// Package-merge, 15 rounds. A round merges the sorted leaves with the sorted packages of the previous round
			// (a leaf goes first on equal counts) and pairs consecutive items. The merge is rank arithmetic -- every
			// leaf and every package finds its place with one binary search, in parallel -- and the per-symbol
			// multiplicity vectors of the new packages are then summed column-wise (thread s = symbol s) with
			// independent, coalesced loads. (The serial two-finger merge this replaces was one dependent L2 round trip
			// per item: 1.2 ms for a single chunk.)
			uint32_t ncur = 0;
			int mylen = myc ? 15 : 0;
			for (uint32_t round = 0; round < 15u; ++round) {
				const uint8_t* cur = gen0 + (round & 1u) * XH_FB_GEN_BYTES;
				uint8_t* nxt = gen0 + ((round & 1u) ^ 1u) * XH_FB_GEN_BYTES;
				const u64* pc = s_pc[round & 1u]; u64* pn = s_pc[(round & 1u) ^ 1u];
				const uint32_t M = ncur + nleaf, nn = M >> 1;
				if (tid < nleaf) {                                   // my leaf: after the packages that are strictly lighter
					const u64 x = s_lcnt[tid];
					uint32_t lo = 0, hi = ncur;
					while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (pc[mid] < x) { lo = mid + 1u; } else { hi = mid; } }
					s_item[tid + lo] = (uint16_t)(0x8000u | s_leaf[tid]);
				}
				if (tid < ncur) {                                    // my package: after the leaves that are not heavier
					const u64 x = pc[tid];
					uint32_t lo = 0, hi = nleaf;
					while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((u64)s_lcnt[mid] <= x) { lo = mid + 1u; } else { hi = mid; } }
					s_item[tid + lo] = (uint16_t)tid;
				}
				__syncthreads();
				FB_T(2)
				if (tid < nn) {
					const uint32_t a = s_item[2u * tid], b2 = s_item[2u * tid + 1u];
					pn[tid] = ((a & 0x8000u) ? (u64)h.cnt[a & 0x7FFFu] : pc[a]) + ((b2 & 0x8000u) ? (u64)h.cnt[b2 & 0x7FFFu] : pc[b2]);
				}
				// multiplicity vectors of the new packages: wave w sums the packages k = w, w+8, ...; lane l owns the dword
				// of symbols 4l..4l+3 (byte-wise sums, no carries: a multiplicity stays below 256) and lane 0 the EOS byte
				{
					const uint32_t wv = tid >> 6, lane = tid & 63u;
					const uint32_t* cur32 = reinterpret_cast<const uint32_t*>(cur);
					uint32_t* nxt32 = reinterpret_cast<uint32_t*>(nxt);
					for (uint32_t k = wv; k < nn; k += 8u) {
						const uint32_t a = xh_uniform(s_item[2u * k]), b2 = xh_uniform(s_item[2u * k + 1u]);
						uint32_t va, vb, ea, eb;
						if (a & 0x8000u) { const uint32_t sy = a & 0x7FFFu; va = (sy < 256u && lane == (sy >> 2)) ? 1u << ((sy & 3u) * 8u) : 0u; ea = (sy == 256u); }
						else { va = cur32[a * (XH_FB_STRIDE / 4u) + lane]; ea = cur[a * XH_FB_STRIDE + 256u]; }
						if (b2 & 0x8000u) { const uint32_t sy = b2 & 0x7FFFu; vb = (sy < 256u && lane == (sy >> 2)) ? 1u << ((sy & 3u) * 8u) : 0u; eb = (sy == 256u); }
						else { vb = cur32[b2 * (XH_FB_STRIDE / 4u) + lane]; eb = cur[b2 * XH_FB_STRIDE + 256u]; }
						nxt32[k * (XH_FB_STRIDE / 4u) + lane] = va + vb;
						if (lane == 0) { nxt[k * XH_FB_STRIDE + 256u] = (uint8_t)(ea + eb); }
					}
				}
				if ((M & 1u) && tid <= 0x100u) {                     // the leftover item is dropped
					const uint32_t a = s_item[M - 1u];
					mylen -= (a & 0x8000u) ? (int)((a & 0x7FFFu) == tid) : (int)cur[a * XH_FB_STRIDE + tid];
				}
				ncur = nn;
				__syncthreads();
				FB_T(3)
			}
			h.lens[tid] = (uint8_t)mylen;
		}
		__syncthreads();
		// size: xh_calc_compressed_len_no_matching (:189-194)
		if (tid < 64u) {
			uint32_t bits = 0;
			for (uint32_t s = tid; s <= 0x100u; s += 64u) { bits += (uint32_t)h.lens[s] * h.cnt[s]; }
			bits = 16u + xh_wave_sum(bits);
			if (tid == 0) { chunk_size[lc] = 256u + (bits + 15u) / 16u * 2u; }
		}
		__syncthreads();
		if (tid < 64u) { huff_canonical(h, tid); }
		__syncthreads();
		if (tid < 64u) { huff_store(h, tid, lens_out + (u64)lc * 512u, codes_out + (u64)lc * 512u); }
		FB_T(4)
	}
}

// ===================================================================================================================
// encode
// ===================================================================================================================
__global__ __launch_bounds__(64) void xh_encode_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                      S16 mlen3, S16 moff,
                                                      const u64* __restrict__ tokbits, const uint8_t* __restrict__ lens_in, const uint16_t* __restrict__ codes_in,
                                                      const uint32_t* __restrict__ fbflag,
                                                      const u64* __restrict__ prefix, uint8_t* __restrict__ d_out)
{
	__shared__ __attribute__((aligned(16))) uint8_t  s_lens[512];
	__shared__ __attribute__((aligned(16))) uint16_t s_codes[512];
	__shared__ uint32_t s_bits[128];                               // ring of 256 16-bit words (word w = half (w&1) of dword (w&255)>>1)
	__shared__ uint32_t s_slot[256];                               // byte position of word w (mod 256)
	const uint32_t lane = threadIdx.x;
	const uint32_t lc = blockIdx.x;
	const ChunkGeom g = chunk_geom(bt, lc);
	const u64 ustart = prefix[bt.chunk_prefix[g.u]];
	const u64 utotal = prefix[bt.chunk_prefix[g.u + 1]] - ustart;
	if (utotal > bt.out_cap[g.u]) { return; }                     // BUF_ERROR unit
	uint8_t* __restrict__ out = d_out + bt.out_off[g.u] + (prefix[lc] - ustart);
	const uint8_t* __restrict__ d = d_in + bt.in_off[g.u] + g.cbase;
	const u64 gbase = (u64)lc * 65536u;
	const bool fallback = fbflag[lc] != 0;                        // literals only: ignore the match arrays

	reinterpret_cast<uint2*>(s_lens)[lane] = reinterpret_cast<const uint2*>(lens_in + (u64)lc * 512u)[lane];
	reinterpret_cast<uint4*>(s_codes)[lane] = reinterpret_cast<const uint4*>(codes_in + (u64)lc * 512u)[lane];
	s_bits[lane] = 0; s_bits[lane + 64u] = 0;
	if (lane == 0) { s_slot[0] = 0; s_slot[1] = 2; }
	__syncthreads();
	// 256-byte table: lens[2i] | lens[2i+1] << 4   (:284 / :320)
	{
		const uint32_t a = reinterpret_cast<const uint32_t*>(s_lens)[lane * 2u], b = reinterpret_cast<const uint32_t*>(s_lens)[lane * 2u + 1u];
		const uint32_t pa = (a & 0xFu) | ((a >> 4) & 0xF0u) | ((a >> 8) & 0xF00u) | ((a >> 12) & 0xF000u);
		const uint32_t pb = (b & 0xFu) | ((b >> 4) & 0xF0u) | ((b >> 8) & 0xF00u) | ((b >> 12) & 0xF000u);
		const uint32_t v = pa | (pb << 16);
		out[lane * 4u] = (uint8_t)v; out[lane * 4u + 1u] = (uint8_t)(v >> 8); out[lane * 4u + 2u] = (uint8_t)(v >> 16); out[lane * 4u + 3u] = (uint8_t)(v >> 24);
	}
	uint8_t* __restrict__ bs = out + 256u;                         // the chunk's bitstream

	uint32_t T = 0, R = 0, Wdone = 0;                            // bits so far, raw bytes so far, words already stored
#define XH_F(t) ((t) ? (((t) - 1u) >> 4) : 0u)                    /* flushes after t bits */
#define XH_ORBITS(val, len, at) { const uint32_t w0_ = (at) >> 4, b_ = (at) & 15u; \
		if (b_ + (len) <= 16u) { atomicOr(&s_bits[(w0_ & 255u) >> 1], (((val) << (16u - b_ - (len))) & 0xFFFFu) << ((w0_ & 1u) * 16u)); } \
		else { atomicOr(&s_bits[(w0_ & 255u) >> 1], (((val) >> (b_ + (len) - 16u)) & 0xFFFFu) << ((w0_ & 1u) * 16u)); \
		       atomicOr(&s_bits[((w0_ + 1u) & 255u) >> 1], (((val) << (32u - b_ - (len))) & 0xFFFFu) << (((w0_ + 1u) & 1u) * 16u)); } }
	const uint32_t nwin = (g.cn + 63u) >> 6;
	// Inputs are burst-loaded XE_GRP windows at a time into registers (one wait on global memory per group instead of two
	// dependent round trips per window) and parked in LDS when their group begins: a window indexes them by its number, which
	// registers cannot do. One stage is enough -- the wave has read the previous group's entries before it stores the next --
	// and with groups of 4 windows the block needs 4.4 KiB of LDS: 32 chunks in flight per CU instead of 19.
#ifndef XE_GRP
#define XE_GRP 4
#endif
	__shared__ uint16_t s_in_off[XE_GRP * 64];
	__shared__ uint16_t s_in_len[XE_GRP * 64];
	__shared__ uint8_t  s_in_byte[XE_GRP * 64];
	__shared__ u64      s_in_tok[XE_GRP];
	uint32_t g_off[XE_GRP], g_len[XE_GRP], g_byte[XE_GRP]; u64 g_tok = 0;
#define XE_BURST_LOAD(gb) { _Pragma("unroll") for (int k_ = 0; k_ < XE_GRP; ++k_) { \
		const uint32_t q_ = (gb) + (uint32_t)k_ * 64u + lane; const uint32_t c_ = q_ < g.cn ? q_ : g.cn - 1u; \
		{ const uint32_t w_ = mlen3.word(gbase + c_); g_off[k_] = w_ >> 16; g_len[k_] = w_ & 0xFFFFu; } g_byte[k_] = d[c_]; } \
		g_tok = (lane < (uint32_t)XE_GRP && ((gb) >> 6) + lane < nwin) ? tokbits[(u64)lc * 1024u + ((gb) >> 6) + lane] : (u64)0; }
#define XE_BURST_STORE() { _Pragma("unroll") for (int k_ = 0; k_ < XE_GRP; ++k_) { \
		s_in_off[k_ * 64 + lane] = (uint16_t)g_off[k_]; s_in_len[k_ * 64 + lane] = (uint16_t)g_len[k_]; s_in_byte[k_ * 64 + lane] = (uint8_t)g_byte[k_]; } \
		if (lane < (uint32_t)XE_GRP) { s_in_tok[lane] = g_tok; } }
#ifndef XE_COMPACT
#define XE_COMPACT 1
#endif
#if XE_COMPACT
	// Round 5: a step works on 64 TOKENS, not on the 64 positions of a window. On the bench corpus a window holds 3-24 tokens (0.04-0.38 per
	// byte), so the scans, the bit ORs and the word stores of a step ran with a fifth of the lanes; the tokens of a group of XE_GRP windows are
	// first compacted -- token of rank r (mbcnt over the window masks) writes its place in the group to s_tokpos[r] -- and then taken 64 at a
	// time in rank order, which is position order: everything behind the lane -> token mapping is the code it was.
	static_assert(XE_GRP * 64 <= 256, "s_tokpos holds a position of the group in a byte");
	__shared__ uint8_t s_tokpos[XE_GRP * 64];
	if (nwin) { XE_BURST_LOAD(0u) }
	const uint32_t ngrp = (nwin + (uint32_t)XE_GRP - 1u) / (uint32_t)XE_GRP;
	for (uint32_t gi = 0; gi <= ngrp; ++gi) {
		uint32_t ntok;
		if (gi < ngrp) {                                          // publish this group's inputs, start loading the next, compact its tokens
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
			XE_BURST_STORE()
			if ((gi + 1u) * (uint32_t)XE_GRP < nwin) { XE_BURST_LOAD((gi + 1u) * (uint32_t)XE_GRP * 64u) }
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
			uint32_t base = 0;
			#pragma unroll
			for (uint32_t k = 0; k < (uint32_t)XE_GRP; ++k) {
				const u64 tm = __hip_atomic_load(&s_in_tok[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				if ((tm >> lane) & (u64)1) { s_tokpos[base + popc_below(tm)] = (uint8_t)(k * 64u + lane); }
				base += (uint32_t)__popcll(tm);
			}
			ntok = xh_uniform(base);
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
		} else {
			if (!g.last) { break; }
			ntok = 1u;                                              // the EOS token of the unit's last chunk (lane 0)
		}
		for (uint32_t r0 = 0; r0 < ntok; r0 += 64u) {
		uint32_t clen = 0, code = 0, ob = 0, offlow = 0, rawn = 0, L = 0;
		bool is_tok = false;
		if (gi < ngrp) {
			is_tok = r0 + lane < ntok;
			if (is_tok) {
				const uint32_t pos = s_tokpos[r0 + lane];
				const uint32_t off = s_in_off[pos];
				uint32_t sym = s_in_byte[pos];
				if (off != 0 && !fallback) {
					L = s_in_len[pos];
					ob = 31u - (uint32_t)__builtin_clz(off);
					sym = 0x100u | (ob << 4) | (L < 15u ? L : 15u);
					offlow = off ^ (1u << ob);
					rawn = L >= 270u ? 3u : (L >= 15u ? 1u : 0u);
				}
				clen = s_lens[sym]; code = s_codes[sym];
			}
		} else {
			is_tok = (lane == 0);
			if (is_tok) { clen = s_lens[0x100]; code = s_codes[0x100]; }
		}
		const uint32_t tb = clen + ob;
		const uint32_t ib = xh_incl_scan_add(tb), ir = xh_incl_scan_add(rawn);
		const uint32_t T0 = T + ib - tb, T1 = T0 + clen, T2 = T1 + ob;
		const uint32_t R0 = R + ir - rawn;
		if (is_tok) {
			if (clen) { XH_ORBITS(code, clen, T0) }
			if (ob) { XH_ORBITS(offlow, ob, T1) }
			const uint32_t f0 = XH_F(T0), f1 = XH_F(T1), f2 = XH_F(T2);
			if (f1 > f0) { s_slot[(f1 + 1u) & 255u] = 4u + 2u * (f1 - 1u) + R0; }
			if (f2 > f1) { s_slot[(f2 + 1u) & 255u] = 4u + 2u * (f2 - 1u) + R0 + rawn; }
			if (rawn) {
				uint8_t* q = bs + 4u + 2u * f1 + R0;
				if (rawn == 1u) { q[0] = (uint8_t)(L - 15u); }
				else { q[0] = 0xFF; q[1] = (uint8_t)L; q[2] = (uint8_t)(L >> 8); }
			}
		}
		T += (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
		R += (uint32_t)__builtin_amdgcn_readlane((int)ir, 63);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
		// store the words that are complete now
		const uint32_t wcomplete = T >> 4;
		for (uint32_t base = Wdone; base < wcomplete; base += 64u) {
			const uint32_t ww = base + lane;
			if (ww < wcomplete) {
				const uint32_t v = (__hip_atomic_load(&s_bits[(ww & 255u) >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> ((ww & 1u) * 16u)) & 0xFFFFu;
				const uint32_t sp = __hip_atomic_load(&s_slot[ww & 255u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				bs[sp] = (uint8_t)v; bs[sp + 1u] = (uint8_t)(v >> 8);
				atomicAnd(&s_bits[(ww & 255u) >> 1], ~(0xFFFFu << ((ww & 1u) * 16u)));
			}
		}
		if (wcomplete > Wdone) { Wdone = wcomplete; }
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
		}
	}
#else
	if (nwin) { XE_BURST_LOAD(0u) }
	for (uint32_t w = 0; w <= nwin; ++w) {
		// window w < nwin: the tokens starting in it; w == nwin: the EOS token of the unit's last chunk (lane 0)
		uint32_t clen = 0, code = 0, ob = 0, offlow = 0, rawn = 0, L = 0;
		bool is_tok = false;
		if (w < nwin) {
			const uint32_t wi = w % (uint32_t)XE_GRP;
			if (wi == 0) {                                          // group start: publish this group's inputs, start loading the next
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
				XE_BURST_STORE()
				if ((w + (uint32_t)XE_GRP) < nwin) { XE_BURST_LOAD((w + (uint32_t)XE_GRP) * 64u) }
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
			}
			const u64 tm = __hip_atomic_load(&s_in_tok[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
			if (tm == 0) { continue; }
			is_tok = (tm >> lane) & (u64)1;
			if (is_tok) {
				const uint32_t off = s_in_off[wi * 64u + lane];
				uint32_t sym = s_in_byte[wi * 64u + lane];
				if (off != 0 && !fallback) {
					L = s_in_len[wi * 64u + lane];
					ob = 31u - (uint32_t)__builtin_clz(off);
					sym = 0x100u | (ob << 4) | (L < 15u ? L : 15u);
					offlow = off ^ (1u << ob);
					rawn = L >= 270u ? 3u : (L >= 15u ? 1u : 0u);
				}
				clen = s_lens[sym]; code = s_codes[sym];
			}
		} else {
			if (!g.last) { break; }
			is_tok = (lane == 0);
			if (is_tok) { clen = s_lens[0x100]; code = s_codes[0x100]; }
		}
		const uint32_t tb = clen + ob;
		const uint32_t ib = xh_incl_scan_add(tb), ir = xh_incl_scan_add(rawn);
		const uint32_t T0 = T + ib - tb, T1 = T0 + clen, T2 = T1 + ob;
		const uint32_t R0 = R + ir - rawn;
		if (is_tok) {
			if (clen) { XH_ORBITS(code, clen, T0) }
			if (ob) { XH_ORBITS(offlow, ob, T1) }
			const uint32_t f0 = XH_F(T0), f1 = XH_F(T1), f2 = XH_F(T2);
			if (f1 > f0) { s_slot[(f1 + 1u) & 255u] = 4u + 2u * (f1 - 1u) + R0; }
			if (f2 > f1) { s_slot[(f2 + 1u) & 255u] = 4u + 2u * (f2 - 1u) + R0 + rawn; }
			if (rawn) {
				uint8_t* q = bs + 4u + 2u * f1 + R0;
				if (rawn == 1u) { q[0] = (uint8_t)(L - 15u); }
				else { q[0] = 0xFF; q[1] = (uint8_t)L; q[2] = (uint8_t)(L >> 8); }
			}
		}
		T += (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
		R += (uint32_t)__builtin_amdgcn_readlane((int)ir, 63);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
		// store the words that are complete now
		const uint32_t wcomplete = T >> 4;
		for (uint32_t base = Wdone; base < wcomplete; base += 64u) {
			const uint32_t ww = base + lane;
			if (ww < wcomplete) {
				const uint32_t v = (__hip_atomic_load(&s_bits[(ww & 255u) >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> ((ww & 1u) * 16u)) & 0xFFFFu;
				const uint32_t sp = __hip_atomic_load(&s_slot[ww & 255u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				bs[sp] = (uint8_t)v; bs[sp + 1u] = (uint8_t)(v >> 8);
				atomicAnd(&s_bits[(ww & 255u) >> 1], ~(0xFFFFu << ((ww & 1u) * 16u)));
			}
		}
		if (wcomplete > Wdone) { Wdone = wcomplete; }
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
	}
#endif
	// Finish (Bitstream.h:142-147): the current word (zero padded) and one zero word
	const uint32_t fe = XH_F(T);
	for (uint32_t ww = Wdone + lane; ww <= fe + 1u; ww += 64u) {
		const uint32_t v = (s_bits[(ww & 255u) >> 1] >> ((ww & 1u) * 16u)) & 0xFFFFu;
		const uint32_t sp = s_slot[ww & 255u];
		bs[sp] = (uint8_t)v; bs[sp + 1u] = (uint8_t)(v >> 8);
	}
#undef XH_F
#undef XH_ORBITS
#undef XE_BURST_LOAD
#undef XE_BURST_STORE
}

// ===================================================================================================================
// launchers
// ===================================================================================================================
void launch_xh_parse(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint16_t* mlen3, const uint16_t* moff,
                     u64* tokbits, uint32_t* counts, uint32_t* extra)
{
	if (bt.n_chunks == 0) { return; }
	hipLaunchKernelGGL(xh_parse_kernel, dim3(bt.n_chunks), dim3(256), 0, st, d_in, bt, mlen3, moff, tokbits, counts, extra);
}
void launch_xh_huff(hipStream_t st, const BatchTables& bt, const uint32_t* counts, const uint32_t* extra, uint8_t* lens, uint16_t* codes,
                    uint32_t* chunk_size, uint32_t* fb_list, uint32_t* fb_count, uint32_t* fbflag)
{
	if (bt.n_chunks == 0) { return; }
	(void)hipMemsetAsync(fb_count, 0, sizeof(uint32_t), st);
	hipLaunchKernelGGL(xh_huff_kernel, dim3(bt.n_chunks), dim3(64), 0, st, bt, counts, extra, lens, codes, chunk_size, fb_list, fb_count, fbflag);
}
void launch_xh_fallback(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint32_t* fb_list, const uint32_t* fb_count,
                        uint32_t blocks, u64* tokbits, uint8_t* lens, uint16_t* codes, uint32_t* chunk_size)
{
	if (bt.n_chunks == 0) { return; }
	const uint32_t grid = bt.n_chunks < blocks ? bt.n_chunks : blocks;
	static PerDeviceOnce attr;
	if (attr.needed()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xh_fallback_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XH_FB_POOL_BYTES); attr.done(); }
	hipLaunchKernelGGL(xh_fallback_kernel, dim3(grid), dim3(512), XH_FB_POOL_BYTES, st, d_in, bt, fb_list, fb_count, tokbits, lens, codes, chunk_size);
}
void launch_xh_encode(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const uint16_t* mlen3, const uint16_t* moff,
                      const u64* tokbits, const uint8_t* lens, const uint16_t* codes, const uint32_t* fbflag, const u64* prefix, uint8_t* d_out)
{
	if (bt.n_chunks == 0) { return; }
	hipLaunchKernelGGL(xh_encode_kernel, dim3(bt.n_chunks), dim3(64), 0, st, d_in, bt, mlen3, moff, tokbits, lens, codes, fbflag, prefix, d_out);
}

} // namespace msc

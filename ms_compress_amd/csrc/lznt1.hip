// lznt1.hip -- LZNT1 one-shot compression for gfx950 (MI355X), bit-exact with the reference CPU encoder.
//
// Replaces: lznt1_compress / lznt1_compress_chunk (/root/reference/src/lznt1_compress.cpp:233-273, :49-94) and
// LZNT1Dictionary::Fill/Find (/root/reference/include/mscomp/LZNT1Dictionary.h:93-106, :114-143).
//
// One wavefront (64-thread block) = one 4 KiB chunk, everything staged in LDS (17.6 KiB -> 9 chunks in flight per CU):
//   A. coalesced 16 B/lane load of the chunk into LDS;
//   B. dictionary = the reference's per-key position arrays as ONE position-sorted bucket array in LDS, built by a stable
//      counting sort on an 11-bit hash of the 3-byte key: histogram by LDS atomics, exclusive scan of the counts (DPP),
//      then an ORDERED scatter in 64 ascending batches (one LDS gather + scatter of the bucket cursors per batch,
//      intra-batch conflicts resolved with ballots). Afterwards cursor[h] = end of bucket h, so the candidates of a
//      position are simply the entries of its bucket that are smaller than the position itself;
//   C. window by window (64 positions, lane = position), LAZILY like the reference (Find runs only where the greedy parse
//      can start a token):
//        1. every lane at or after the parse position scans the 4 OLDEST candidates of its position itself (unconditional
//           loads in flight together, own bytes in registers, 16-byte compares, strictly-longer wins, early exit at max_len);
//        2. the greedy walk runs on the scalar unit over ballot masks (s_ff1 over literal runs). When it lands on a
//           position that still has unexamined candidates, the whole wave finishes that ONE position: 64 candidates per
//           step, DPP max of (len, -position) = longest, oldest on ties, stop when max_len is reached. Positions covered
//           by a match are never finished;
//        3. tokens go straight to the chunk's scratch slot, placed by the closed form
//           pos(t) = (t div 8 + 1) + sum size(u<t)   (mbcnt prefix popcounts); flag bits collect in a 16-entry LDS ring;
//   D. header (0xB000|size-1), or the raw chunk (0x3000|n-1) when the running size reaches n; util.hip concatenates slots.
#include "common.h"
#include "kernels.h"

namespace msc {

#ifdef LZ_PROFILE   // dev-only phase timers (s_memtime cycles summed over blocks); not in the production build
__device__ unsigned long long g_lz_prof[16];
#define LZ_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); t_acc[i] += t_ - t_prev; t_prev = t_; }
#define LZ_T0   unsigned long long t_prev = __builtin_readcyclecounter(); unsigned long long t_acc[16] = {0};
#define LZ_CNT(i, v) { t_acc[i] += (unsigned long long)(v); }
#define LZ_TEND if (lane == 0) { for (int i_ = 0; i_ < 16; ++i_) { atomicAdd(&g_lz_prof[i_], t_acc[i_]); } }
#else
#define LZ_T(i)
#define LZ_T0
#define LZ_CNT(i, v)
#define LZ_TEND
#endif

#ifndef LZ_TBL_BITS
#ifdef LZ_TBL_GLOBAL
#define LZ_TBL_BITS 12
#else
#define LZ_TBL_BITS 11
#endif
#endif
#define LZ_TBL      (1u << LZ_TBL_BITS)
#ifndef LZ_SELF
#define LZ_SELF     4u       // candidates each lane scans by itself before the wave cooperates
#endif

// (never 0: bucket 0 stays empty, so "the end of the bucket before mine" is always cnt[h - 1] -- no special case in the parse.
// Keys that hash to 0 share bucket 1: a bucket may hold several keys anyway, the 3-byte compare sorts them out.)
__device__ __forceinline__ uint32_t lz_hash(uint32_t key24) { const uint32_t h = (key24 * 0x9E3779B1u) >> (32 - LZ_TBL_BITS); return h ? h : 1u; }

// position-dependent split of the 16-bit match token (lznt1_compress.cpp:51,66) -- pure function of pos
__device__ __forceinline__ uint32_t lz_shift(uint32_t pos)
{
	// = pos <= 16 ? 12 : 12 - (bits(pos - 1) - 4), as one unsigned minimum: clz(pos - 1) - 16 is >= 12 for pos <= 16 and wraps to a huge
	// value for pos <= 1 (v_ffbh_u32 gives -1 for 0)
	uint32_t z; asm("v_ffbh_u32 %0, %1" : "=v"(z) : "v"(pos - 1u));
	return (z - 16u) < 12u ? z - 16u : 12u;
}

// racing LDS accesses between lanes of one wave: relaxed wavefront-scope atomics (plain ds_read/ds_write in the ISA;
// DS operations of a wave execute in program order) so the compiler neither forwards nor reorders them.
__device__ __forceinline__ void     wst16(uint16_t* p, uint32_t v) { __hip_atomic_store(p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ uint32_t wld16(uint16_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ void     wst32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ uint32_t wld32(uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ void     wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local"); }

// 16 bytes at any byte offset of the chunk in LDS (common.h: lds_ld128), the offset taken modulo 4096: the AND that aligns the dword
// address also keeps it inside the chunk, so callers need no select for candidates that do not exist (they are masked afterwards).
__device__ __forceinline__ uint4 lz_ld128(const uint8_t* base, uint32_t off)
{
	const uint32_t* a = reinterpret_cast<const uint32_t*>(base + (off & 0xFFCu));
	const uint32_t sh = off & 3u;
	const uint32_t w0 = a[0], w1 = a[1], w2 = a[2], w3 = a[3], w4 = a[4];
	return make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
	                  __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh));
}
// Common prefix beyond the first 16 (equal) bytes of d[q..] and d[p..]: 16 bytes per step; the result may exceed maxlen
// (callers clamp).
__device__ __forceinline__ uint32_t lz_lcp_tail(const uint8_t* d, uint32_t q, uint32_t p, uint32_t maxlen)
{
	uint32_t l = 16u;
	for (;;) {
		uint4 a, b;
		a = lz_ld128(d, q + l); b = lz_ld128(d, p + l);
		const uint32_t f = first_nz_byte16(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
		l += f;
		if (f < 16u || l >= maxlen) { break; }
	}
	return l;
}
// First difference of two 16-byte blocks given their XOR, as a BIT index limited to capbits (<= 128): common.h first_nz_byte16 without
// its final shift -- the limit (8 x min(max_len, 16)) rides in the minimum that finds the difference, so the byte count needs no clamp.
__device__ __forceinline__ uint32_t lz_diff_bits16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t capbits)
{
	const uint32_t a = ffbl_raw(x1) | 32u, b = ffbl_raw(x2) | 64u, c = ffbl_raw(x3) | 96u;
	return min3u(min3u(ffbl_raw(x0), a, b), c, capbits);
}
// One window (64 positions, lane = position) of the lazy parse of a chunk: Find for the positions at / after the parse
// position `entry` (the 4 oldest candidates per lane, the rest on demand), greedy walk, token mask. Returns the parse
// position after the window; key = (len << 12) | (4095 - q) of this lane's match (valid where matchmask is set),
// o0 = the 4 bytes at this lane's position, shift = its token split.
struct LzWin { uint32_t key, o0, shift; u64 tokmask, matchmask; };
__device__ __forceinline__ uint32_t lz_window(const uint8_t* s_data, const uint16_t* s_cnt, const uint16_t* s_bucket, uint32_t n, uint32_t lane,
                                              uint32_t wbase, uint32_t entry, LzWin& r)
{
	entry = (uint32_t)__builtin_amdgcn_readfirstlane((int)entry); wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);   // (uniform: keeps the walk's bookkeeping on the scalar unit)
	const uint32_t wend = (wbase + 64u < n) ? wbase + 64u : n;
	const uint32_t p = wbase + lane;
	const uint4 own = lz_ld128(s_data, p);                       // (aligned dword reads: a misaligned 16-byte read is replayed)
	const uint32_t o0 = own.x, o1 = own.y, o2 = own.z, o3 = own.w;
	const uint32_t shift = lz_shift(p);
	uint32_t maxlen = 0, s = 0, e = 0;                        // my candidates: bucket[s..e) entries that are < p (ascending)
	if (p >= entry && p > 0 && p + 3u <= n) {
		const uint32_t mask3 = (1u << shift) + 2u;
		maxlen = (n - p < mask3) ? n - p : mask3;
		const uint32_t h = lz_hash(o0 & 0xFFFFFFu);          // >= 1
#ifdef LZ_TBL_GLOBAL
		const uint32_t se = ld32(reinterpret_cast<const uint8_t*>(s_cnt + (h - 1u)));   // (the table lies in global memory here: both ends in ONE 4-byte gather)
		s = se & 0xFFFFu; e = se >> 16;
#else
		e = s_cnt[h];                                          // bucket h = [end[h-1], end[h])
		s = s_cnt[h - 1u];
#endif
	}
	// 1. the oldest LZ_SELF candidates (LZNT1Dictionary.h:124-135: in order, strictly longer wins, stop at max_len). With
	// key = (len << 12) | (4095 - q) the reference's choice is simply the MAXIMUM of the candidates' keys: longest first,
	// then oldest (the candidates come oldest first, so "strictly longer" keeps the oldest of equals), and nothing can
	// beat a candidate that reached max_len. A key below 3 << 12 (hash collision) is "no match" to everything downstream.
	uint32_t key = 0;
	// All loads are UNCONDITIONAL so that they issue back to back and are waited for once: the five bucket entries from one
	// address (entries past the bucket's end -- the next buckets', or the first words of the table behind the array -- are
	// masked by the count), the candidates' bytes with the offset taken modulo 4096.
	uint32_t q[LZ_SELF + 1u];
	bool ex[LZ_SELF + 1u];                                    // candidate j exists: j < e - s and it lies before me
	const uint32_t cnt = e - s;
	#pragma unroll
	// (five 2-byte reads, kept apart by the relaxed-atomic form: merged into an 8-byte + a 2-byte read, as the compiler does with plain loads, the 8-byte
	// one is only 2-byte aligned and is replayed -- SQ_LDS_UNALIGNED_STALL 14 % of the LDS pipe's busy cycles, 54.4 against 53.3 ms on configs[4])
	for (uint32_t j = 0; j <= LZ_SELF; ++j) { q[j] = wld16(const_cast<uint16_t*>(s_bucket) + s + j); }
	#pragma unroll
	for (uint32_t j = 0; j <= LZ_SELF; ++j) { ex[j] = j < cnt && q[j] < p; }
	const bool longer = maxlen > 16u;
	const uint32_t capb = (maxlen < 16u ? maxlen : 16u) << 3;
#if defined(LZ_PROBE) && LZ_PROBE == 5      /* dev probe (SUBTRACTIVE, not bit-exact): the eager scan of the odd windows is skipped (their positions are literals unless finished) */
	if (!((wbase >> 6) & 1u))
#endif
	{
		uint4 c[LZ_SELF];                                     // (all loads in flight together)
		#pragma unroll
		for (uint32_t k = 0; k < LZ_SELF; ++k) { c[k] = lz_ld128(s_data, q[k]); }
		#pragma unroll
		for (uint32_t k = 0; k < LZ_SELF; ++k) {
			uint32_t lb = lz_diff_bits16(c[k].x ^ o0, c[k].y ^ o1, c[k].z ^ o2, c[k].w ^ o3, capb);     // length in bits (the low three dropped below)
			if (ex[k] && lb >= 128u && longer) { const uint32_t l = lz_lcp_tail(s_data, q[k], p, maxlen); lb = (l < maxlen ? l : maxlen) << 3; }   // long match
			const uint32_t kk = ex[k] ? (((lb & ~7u) << 9) | (q[k] ^ 4095u)) : 0u;
			key = kk > key ? kk : key;
		}
	}
#if defined(LZ_PROBE) && LZ_PROBE == 1      /* dev probe: 40 more VALU instructions per window */
	{ uint32_t y_ = key; _Pragma("unroll") for (int q_ = 0; q_ < 40; ++q_) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(y_) : "v"(o0)); } asm volatile("" :: "v"(y_)); }
#elif defined(LZ_PROBE) && LZ_PROBE == 2    /* dev probe: four more 16-byte LDS reads per window */
	{ _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) { const uint4 y_ = lds_ld128(s_data, (p * 7u + 64u * q_) & 4095u); asm volatile("" :: "v"(y_.x), "v"(y_.y), "v"(y_.z), "v"(y_.w)); } }
#endif
	// 2. greedy walk; unresolved positions (a fifth older candidate exists and max_len was not reached) are finished by the whole wave
	// when (and only when) the walk lands on them. (The masks are built from ballots of single compares, combined on the scalar
	// unit: the ballot of a combined condition costs two more vector instructions. len <= maxlen, so "not reached" is key < maxlen << 12.)
	u64 un = __builtin_amdgcn_ballot_w64(cnt > LZ_SELF) & __builtin_amdgcn_ballot_w64(q[LZ_SELF] < p) & __builtin_amdgcn_ballot_w64(key < (maxlen << 12));
#if defined(LZ_PROBE) && LZ_PROBE == 6      /* dev probe (SUBTRACTIVE, not bit-exact): no position is ever finished by the wave -- the 4 eager candidates are all there is */
	un = 0;
#endif
	u64 mm = __builtin_amdgcn_ballot_w64(key >= (3u << 12)) & ~un;              // resolved positions that have a match
	// The serial loop only decides which candidates are TAKEN; everything else (which positions are literal tokens)
	// is derived in parallel afterwards.
	u64 matchmask = 0;
	const uint32_t wn = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wend - wbase));
	// Every resolved match lane precomputes where the walk goes after taking it: the first stop (match or unresolved
	// position) at or after its end, relative to the window (>= wn leaves it). The scalar walk is then one
	// v_readlane per taken match; it leaves the asm block on an unresolved position (st = 1), which is finished by
	// the whole wave. Stops are only ever removed at the walk's own position, so the table never goes stale ahead.
	un = sgpr64(un); mm = sgpr64(mm);
	// (a target at or beyond wn only has to be >= wn: the walk ends there and the position after the window comes from the
	// match ends below. So no guard for nx >= 64: the shift count wraps, and nx + anything is already >= 64 >= wn.)
	const uint32_t nx = lane + (key >> 12);
	const u64 restl = (un | mm) >> (nx & 63u);
	const uint32_t zl = min3u(ffbl_raw((uint32_t)restl), ffbl_raw((uint32_t)(restl >> 32)) | 32u, 64u);   // 64 = no stop left
	const uint32_t J = (nx + zl < wn) ? nx + zl : wn;
	uint32_t mp, rel;                                            // rel: the next token start, relative to the window (entry < wend here: < 64)
	{
		// first stop at or after the next token start, on the scalar unit: the asm output pins the chain there
		asm("s_max_u32 %0, %1, %2\n\ts_sub_u32 %0, %0, %2" : "=&s"(rel) : "s"(entry), "s"(wbase) : "scc");   // (a saturating subtract would go to the vector unit)
		u64 rest; asm("s_lshr_b64 %0, %1, %2" : "=s"(rest) : "s"(un | mm), "s"(rel) : "scc");
		mp = rest ? rel + ctz64(rest) : wn;
	}
	while (mp < wn) {
		uint32_t st;
		matchmask = sgpr64(matchmask);
		// v_readlane needs 4 wait states after the write of its lane select (mp): on the loop edge the five scalar
		// instructions in between provide them, on entry the s_nop does.
		asm volatile(
			"s_nop 3\n\t"
			"1:\n\t"
			"s_bitcmp1_b64 %[un], %[mp]\n\t"
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
			: [un] "s"(un), [wn] "s"(wn), [J] "v"(J)
			: "scc");
		if (st == 0) { break; }
		{
			// finish position wbase+mp: the candidates after the first LZ_SELF of its bucket, oldest first, 64 per step
			const uint32_t sL = (uint32_t)__builtin_amdgcn_readlane((int)s, (int)mp);
			const uint32_t eL = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)mp);
			const uint32_t maxL = (uint32_t)__builtin_amdgcn_readlane((int)maxlen, (int)mp);
			const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)o0, (int)mp), a1 = (uint32_t)__builtin_amdgcn_readlane((int)o1, (int)mp);
			const uint32_t a2 = (uint32_t)__builtin_amdgcn_readlane((int)o2, (int)mp), a3 = (uint32_t)__builtin_amdgcn_readlane((int)o3, (int)mp);
			uint32_t kbest = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)mp);
			const uint32_t pL = wbase + mp;
			const bool longL = maxL > 16u;
			const uint32_t capL = (maxL < 16u ? maxL : 16u) << 3;
			for (uint32_t base = sL + LZ_SELF; base < eL; base += 64u) {
				const uint32_t qq = s_bucket[base + lane];         // unconditional load (past the array's end it reads the table), masked below
				const bool v1 = lane < eL - base, v2 = qq < pL;     // else: this lane is at or beyond pL's own entry
				const bool valid = v1 && v2;
				const u64 vmask = __builtin_amdgcn_ballot_w64(v1) & __builtin_amdgcn_ballot_w64(v2);
				const uint4 c = lz_ld128(s_data, qq);
				uint32_t l2 = lz_diff_bits16(c.x ^ a0, c.y ^ a1, c.z ^ a2, c.w ^ a3, capL);   // in bits
				if (valid && l2 >= 128u && longL) { const uint32_t l = lz_lcp_tail(s_data, qq, pL, maxL); l2 = (l < maxL ? l : maxL) << 3; }
#if defined(LZ_PROBE) && LZ_PROBE == 3      /* dev probe: 20 more VALU instructions per finishing step */
				{ uint32_t y_ = l2; _Pragma("unroll") for (int q_ = 0; q_ < 20; ++q_) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(y_) : "v"(qq)); } asm volatile("" :: "v"(y_)); }
#elif defined(LZ_PROBE) && LZ_PROBE == 4    /* dev probe: one more 16-byte LDS read per finishing step */
				{ const uint4 y_ = lds_ld128(s_data, (qq * 7u + 64u) & 4095u); asm volatile("" :: "v"(y_.x), "v"(y_.y), "v"(y_.z), "v"(y_.w)); }
#endif
				uint32_t k2; asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(k2) : "v"(((l2 & ~7u) << 9) | (qq ^ 4095u)), "s"(vmask));   // (a key below 3 << 12 is no match)
				const uint32_t m = wave_max_u32(k2);
				kbest = m > kbest ? m : kbest;                   // longest, then oldest (older blocks hold the larger 4095 - q)
				if ((kbest >> 12) == maxL || ~vmask) { break; }  // max_len reached / all older candidates seen
			}
			if (lane == mp) { key = kbest; }
			un = sgpr64(un & ~(((u64)1) << mp));
			// literals before mp are settled; mp itself is now resolved: take its match, or step over it as a literal
			uint32_t nxs = mp + 1u;
			if ((kbest >> 12) >= 3u) { matchmask |= ((u64)1) << mp; nxs = mp + (kbest >> 12); }
			const u64 rest = nxs < 64u ? (un | mm) >> nxs : (u64)0;
			mp = (uint32_t)__builtin_amdgcn_readfirstlane((int)(nxs >= wn ? nxs : (rest ? nxs + ctz64(rest) : wn)));
		}
	}
	// tokens of the window = positions >= entry that no taken match covers: covered <=> the furthest end of the taken
	// matches starting at or before me lies beyond me and I am not such a start myself
	matchmask = sgpr64(matchmask);
	uint32_t mend; asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(mend) : "v"(p + (key >> 12)), "s"(matchmask));   // the mask itself selects: no per-lane bit test
	const uint32_t reach = wave_incl_scan_max(mend);
	const u64 range = (~(u64)0 << rel) & (wn >= 64u ? ~(u64)0 : (((u64)1 << wn) - 1u));     // entry <= p < wend, on the scalar unit
	const u64 tokmask = range & (matchmask | __builtin_amdgcn_ballot_w64(reach <= p));
	const uint32_t wreach = (uint32_t)__builtin_amdgcn_readlane((int)reach, 63);
	const uint32_t cur = wreach > wend ? wreach : wend;

	r.key = key; r.o0 = o0; r.shift = shift; r.tokmask = tokmask; r.matchmask = matchmask;
	return cur;
}

// One returning LDS atomic for the 64 positions of a batch; `serial`: the same, one lane at a time in lane order (kernels.h: a device whose
// same-address atomics are not served in lane order -- DS operations of a wave execute in program order)
template <bool serial>
__device__ __forceinline__ uint32_t lz_ordered_add(uint32_t* addr, uint32_t v, bool active, uint32_t lane)
{
	uint32_t old = 0;
	if (!serial) { if (active) { old = __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); } }
	else { for (uint32_t l = 0; l < 64u; ++l) { if (lane == l && active) { old = __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); } wave_fence(); } }
	return old;
}

template <bool serial>    // (a template, not an argument: the default kernels are the code they were)
__global__ __launch_bounds__(64) void lznt1_chunk_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                        uint8_t* __restrict__ slots, uint32_t* __restrict__ slot_size)
{
	// ONE LDS object, the chunk FIRST: its 16-byte reads then need no address arithmetic beyond the AND that aligns them (the DS offset
	// fields reach 1 KiB / 64 KiB from the register address), and the bucket array is followed by the count table (lz_window reads up to
	// four entries past a bucket's end)
	struct __attribute__((aligned(16))) Lds {
		uint8_t  data[4096 + 32];
		uint16_t bucket[4096];      // positions sorted by (hash, position)
		uint16_t cnt[LZ_TBL];       // counts -> bucket ends
		uint32_t flagacc[16];       // flag bits of the groups in flight
		uint32_t flagpos[16];       // their byte position in the image
	};
	__shared__ Lds L;
	uint8_t* const s_data = L.data; uint16_t* const s_cnt = L.cnt; uint16_t* const s_bucket = L.bucket;
	uint32_t* const s_flagacc = L.flagacc; uint32_t* const s_flagpos = L.flagpos;

	const uint32_t lane = threadIdx.x;
	const uint32_t c = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, c);
	const u64 coff = (u64)(c - bt.chunk_prefix[u]) * 4096u;
	const u64 left = bt.in_len[u] - coff;
	const uint32_t n = left < 4096u ? (uint32_t)left : 4096u;
	const uint8_t* __restrict__ src = d_in + bt.in_off[u] + coff;
	uint8_t* __restrict__ img = slots + (u64)c * LZNT1_SLOT;              // chunk image: 2-byte header + payload

	LZ_T0
	// ---- A. stage the chunk, clear the count table -----------------------------------------------------------
	{
		const uint32_t nvec = (((uintptr_t)src & 15u) == 0) ? (n & ~15u) : 0u;
		for (uint32_t i = lane * 16u; i < nvec; i += 1024u) {
			*reinterpret_cast<uint4*>(s_data + i) = *reinterpret_cast<const uint4*>(src + i);
		}
		for (uint32_t i = nvec + lane; i < n; i += 64u) { s_data[i] = src[i]; }
		for (uint32_t i = n + lane; i < 4096u + 32u; i += 64u) { s_data[i] = 0; }
		for (uint32_t i = lane * 8u; i < LZ_TBL; i += 512u) { *reinterpret_cast<uint4*>(s_cnt + i) = make_uint4(0, 0, 0, 0); }
		if (lane < 16u) { s_flagacc[lane] = 0; }
	}
	__syncthreads();
	LZ_T(0)

	// ---- B1. histogram of the position hashes (two u16 counters per dword); the value each atomic RETURNS is the
	// position's rank inside its bucket: batches are issued in ascending position order, and within one DS instruction
	// the LDS unit serialises same-address atomics in lane order (gfx950 behaviour, checked by tools/dev/lds_order_test.hip
	// and, indirectly, by every parity test: a different order would change which candidate wins a tie) ------------------
	const uint32_t nb = (n + 63u) >> 6;
	uint32_t rk[32];                                             // ranks of my 64 positions, two per register
	#pragma unroll
	for (uint32_t b = 0; b < 64u; ++b) {
		uint32_t r = 0;
		if (b < nb) {
			const uint32_t p = b * 64u + lane;
			{
				const bool act = p + 2u < n;
				const uint32_t h = lz_hash(ld32(s_data + (act ? p : 0u)) & 0xFFFFFFu);
				const uint32_t old = lz_ordered_add<serial>(reinterpret_cast<uint32_t*>(s_cnt) + (h >> 1), (h & 1u) ? 0x10000u : 1u, act, lane);
				r = act ? ((h & 1u) ? old >> 16 : old & 0xFFFFu) : 0u;
			}
		}
		rk[b >> 1] = (b & 1u) ? (rk[b >> 1] | (r << 16)) : r;
	}
	__syncthreads();
	LZ_T(1)
	// ---- B2. inclusive scan of the 2048 counts -> bucket ENDS (bucket h = [end[h-1], end[h])): coalesced rounds of 8
	// bins per lane -----------------------------------------------------------------------------------------------------
	{
		uint32_t run = 0;
		for (uint32_t k = 0; k < LZ_TBL / 512u; ++k) {
			uint4 a = reinterpret_cast<uint4*>(s_cnt)[k * 64u + lane];
			uint32_t w[4] = { a.x, a.y, a.z, a.w };
			uint32_t sum = 0;
			#pragma unroll
			for (int i = 0; i < 4; ++i) { sum += (w[i] & 0xFFFFu) + (w[i] >> 16); }
			const uint32_t incl = wave_incl_scan_add_u32(sum);
			uint32_t r = run + incl - sum;
			#pragma unroll
			for (int i = 0; i < 4; ++i) { const uint32_t lo = w[i] & 0xFFFFu, hi = w[i] >> 16; w[i] = (r + lo) | ((r + lo + hi) << 16); r += lo + hi; }
			reinterpret_cast<uint4*>(s_cnt)[k * 64u + lane] = make_uint4(w[0], w[1], w[2], w[3]);
			run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		}
	}
	__syncthreads();
	// ---- B3. scatter: bucket[start[h] + rank] = p (positions of a bucket end up in ascending order) --------------------
	#pragma unroll
	for (uint32_t b = 0; b < 64u; ++b) {
		if (b < nb) {
			const uint32_t p = b * 64u + lane;
			if (p + 2u < n) {
				const uint32_t h = lz_hash(ld32(s_data + p) & 0xFFFFFFu);
				const uint32_t start = s_cnt[h - 1u];                  // (h >= 1)
				const uint32_t r = (b & 1u) ? rk[b >> 1] >> 16 : rk[b >> 1] & 0xFFFFu;
				s_bucket[start + r] = (uint16_t)p;
			}
		}
	}
	__syncthreads();
	LZ_T(2)

	// ---- C. lazy windows: Find -> greedy walk (finishing positions on demand) -> emit --------------------------------
	uint32_t entry = 0, T = 0, S = 0;                            // next token start, tokens so far, sum of token sizes so far
	bool raw = false;
	const uint32_t nw = (n + 63u) >> 6;
	for (uint32_t w = 0; w < nw; ++w) {
		const uint32_t wbase = w * 64u;
		const uint32_t wend = (wbase + 64u < n) ? wbase + 64u : n;
		if (entry >= wend) { continue; }                         // window wholly covered by a match
		LzWin r;
		const uint32_t cur = lz_window(s_data, s_cnt, s_bucket, n, lane, wbase, entry, r);
		const uint32_t p = wbase + lane;
		const uint32_t key = r.key, o0 = r.o0, shift = r.shift;
		const u64 tokmask = r.tokmask, matchmask = r.matchmask;
		const bool is_m = (matchmask >> lane) & (u64)1, is_tok = (tokmask >> lane) & (u64)1;
		entry = cur;
		LZ_T(4)

		// 3. emit: pos(t) = 2 (header) + (t div 8 + 1) + sum size(u<t)
		const uint32_t tb = popc_below(tokmask), mbl = popc_below(matchmask);
		const uint32_t t = T + tb;
		const uint32_t pos = 3u + (t >> 3) + S + tb + mbl;
		if (is_tok) {
			if ((t & 7u) == 0) { wst32(&s_flagpos[(t >> 3) & 15u], pos - 1u); }
			if (is_m) {
				const uint32_t best = key >> 12;
				const uint32_t tok = ((p - (4095u - (key & 0xFFFu)) - 1u) << shift) | (best - 3u);
				img[pos] = (uint8_t)tok; img[pos + 1u] = (uint8_t)(tok >> 8);
			} else { img[pos] = (uint8_t)o0; }
		}
		if (is_m) { atomicOr(&s_flagacc[(t >> 3) & 15u], 1u << (t & 7u)); }
		wave_fence();
		const uint32_t nt = (uint32_t)__popcll(tokmask), nm = (uint32_t)__popcll(matchmask);
		// groups completed in this window: their flag byte is final
		if (is_tok && (t & 7u) == 7u) {
			const uint32_t g = (t >> 3) & 15u;
			img[wld32(&s_flagpos[g])] = (uint8_t)wld32(&s_flagacc[g]);
			wst32(&s_flagacc[g], 0u);
		}
		wave_fence();
		T += nt; S += nt + nm;
		LZ_T(5)
		if (((T + 7u) >> 3) + S >= n) { raw = true; break; }    // running size reached n (:85-86) => store raw
	}

	// ---- D. header / raw chunk -----------------------------------------------------------------------------------------
	const uint32_t csize = ((T + 7u) >> 3) + S;
	uint32_t total;
	if (!raw && csize < n) {
		if (lane == 0) {
			if (T & 7u) { const uint32_t g = (T >> 3) & 15u; img[s_flagpos[g]] = (uint8_t)s_flagacc[g]; }   // the last, partial group
			img[0] = (uint8_t)(0xB000u | (csize - 1u)); img[1] = (uint8_t)((0xB000u | (csize - 1u)) >> 8);
		}
		total = 2u + csize;
	} else {
		const uint32_t hdr = 0x3000u | (n - 1u);
		for (uint32_t k = lane; k < (n + 2u + 3u) / 4u; k += 64u) {
			reinterpret_cast<uint32_t*>(img)[k] = (k == 0) ? (hdr | (ld16(s_data) << 16)) : ld32(s_data + 4u * k - 2u);
		}
		total = 2u + n;
	}
	if (lane == 0) { slot_size[c] = total; }
	LZ_T(6)
	LZ_TEND
}
#ifdef LZ_PROFILE
extern "C" void mscomp_amd_debug_lz_prof(unsigned long long* out, int reset)
{
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lz_prof), sizeof(unsigned long long) * 16);
	if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lz_prof), z, sizeof z); }
}
#endif

// ===================================================================================================================
// Four waves per chunk (same bytes as the kernel above)
// ===================================================================================================================
// The sort (B) stays with wave 0 -- the rank trick needs the positions in order -- but the lazy parse (C), 85 % of the
// time, is split: the parse's only state is the position of the next token, so wave j parses the windows
// [16 j, 16 j + 16) SPECULATIVELY as if a token started at position 1024 j, records per window the token mask, the match
// mask, the 16-bit match tokens (a match token depends on its position only) and the parse position after it; then wave
// j repairs the seam BEHIND its segment (it continues with its true position into segment j+1 until the position after
// a window equals the recorded one), wave 0 checks that no repair ran through a whole segment (else it cascades,
// serially), scans tokens / bytes per window, and the four waves emit their segments at known offsets; the flag bytes
// are OR-ed into a per-group LDS array and stored at the end.
// Segments (in windows): 4 of shrinking length (22 / 18 / 13 / 11) -- a window costs more the later it lies in the chunk
// (fuller buckets) and every seam costs a repair. Round 5, after the window parse lost a quarter of its vector instructions (the
// finishing steps, which grow along the chunk, lost less), configs[4]: boundaries 16/32/48: 56.9 ms, 18/34/49: 54.9, 20/37/51: 52.6,
// 22/40/53: 52.5, 24/43/55 (rounds 3-4): 53.5, 26/46/57: 54.4.
#define LZ4_NSEG 4u
#ifndef LZ4_B1
#define LZ4_B1 22u
#define LZ4_B2 40u
#define LZ4_B3 53u
#endif
__device__ __forceinline__ uint32_t lz4_seg_start(uint32_t j) { return j == 0 ? 0u : j == 1 ? LZ4_B1 : j == 2 ? LZ4_B2 : j == 3 ? LZ4_B3 : 64u; }
#define LZ4_MAXM 22u                                             // matches that can START in one window of 64 positions
#ifdef LZ4_PROFILE
__device__ unsigned long long g_lz4_prof[8];
extern "C" void mscomp_amd_debug_lz4_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lz4_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lz4_prof), z, 64); }
#endif
template <bool serial>
__global__ __launch_bounds__(256) void lznt1_chunk4_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                          uint8_t* __restrict__ slots, uint32_t* __restrict__ slot_size)
{
#ifdef LZ_TBL_GLOBAL
	// dev variant: 12-bit hash. 20 480 B exactly -> still 8 blocks per CU: the bucket-end table (8 KiB) shares its LDS with the parse records -- it is
	// written to the chunk's scratch slot in global memory after the sort and looked up there by the parse (one 4-byte gather per lane and window) --
	// the prefixes and flags of the emission take the place of the (then dead) bucket array, and the chunk has no pad behind it (a read behind the
	// chunk falls into bucket[]: every length is capped by what the chunk has left).
	struct __attribute__((aligned(16))) Lds {
		uint8_t  data[4096];
		uint16_t bucket[4096];
		union {
			uint16_t cnt[LZ_TBL];
			struct { u64 tok[64], mat[64]; uint16_t endc[64]; uint16_t ptok[64][LZ4_MAXM]; uint32_t prog[LZ4_NSEG]; uint32_t used[LZ4_NSEG]; uint32_t segctr; uint32_t total[2]; } r;
		};
	};
	static_assert(sizeof(Lds) <= 20480, "eight blocks per CU");
	__shared__ Lds L;
	uint8_t* const s_data = L.data; uint16_t* const s_cnt = L.cnt; uint16_t* const s_bucket = L.bucket;
	u64* const s_tok = L.r.tok; u64* const s_mat = L.r.mat; uint16_t* const s_endc = L.r.endc; uint16_t (* const s_ptok)[LZ4_MAXM] = L.r.ptok;
	uint32_t* const s_prog = L.r.prog; uint32_t* const s_used = L.r.used; uint32_t& s_segctr = L.r.segctr; uint32_t* const s_total = L.r.total;
	uint16_t* const s_T = s_bucket;                                        // [64] tokens before window w          } after the parse (bucket[] is dead then)
	uint16_t* const s_S = s_bucket + 64;                                   // [64] token bytes before window w     }
	uint16_t* const s_flagpos = s_bucket + 128;                            // [512] byte position of group g's flag byte
	uint32_t* const s_flagacc = reinterpret_cast<uint32_t*>(s_bucket + 640);   // [512] its bits
#else
	struct __attribute__((aligned(16))) Lds {                              // one object, the chunk first (see the kernel above); 20 432 B -> 8 blocks per CU
		uint8_t  data[4096 + 32];
		uint16_t bucket[4096];
		uint16_t cnt[LZ_TBL];                                              // counts -> bucket ends; after the parse: prefixes and flags (below)
		u64      tok[64], mat[64];                                         // per window
		uint16_t endc[64];                                                 // parse position after the window
		uint16_t ptok[64][LZ4_MAXM];                                       // its match tokens, in order
		uint32_t prog[LZ4_NSEG];                                           // windows finished in segment j (index + 1)
		uint32_t used[LZ4_NSEG];                                           // entry position the seam in front of segment j was repaired against
		uint32_t segctr;                                                   // next segment to hand out
		uint32_t total[2];
	};
	static_assert(sizeof(Lds) <= (LZ_TBL_BITS == 11 ? 20480 : 27306), "eight blocks per CU (dev variants with a wider hash: six)");
	__shared__ Lds L;
	uint8_t* const s_data = L.data; uint16_t* const s_cnt = L.cnt; uint16_t* const s_bucket = L.bucket;
	u64* const s_tok = L.tok; u64* const s_mat = L.mat; uint16_t* const s_endc = L.endc; uint16_t (* const s_ptok)[LZ4_MAXM] = L.ptok;
	uint32_t* const s_prog = L.prog; uint32_t* const s_used = L.used; uint32_t& s_segctr = L.segctr; uint32_t* const s_total = L.total;
	uint16_t* const s_T = s_cnt;                                           // [64] tokens before window w          } after the parse
	uint16_t* const s_S = s_cnt + 64;                                      // [64] token bytes before window w     }
	uint16_t* const s_flagpos = s_cnt + 128;                               // [512] byte position of group g's flag byte
	uint32_t* const s_flagacc = reinterpret_cast<uint32_t*>(s_cnt + 640);  // [512] its bits (LZ_TBL u16 = 4096 B >= 1280 + 2048)

#endif
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
	const uint32_t c = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, c);
	const u64 coff = (u64)(c - bt.chunk_prefix[u]) * 4096u;
	const u64 left = bt.in_len[u] - coff;
	const uint32_t n = left < 4096u ? (uint32_t)left : 4096u;
	const uint8_t* __restrict__ src = d_in + bt.in_off[u] + coff;
	uint8_t* __restrict__ img = slots + (u64)c * LZNT1_SLOT;

#ifdef LZ4_PROFILE
	unsigned long long z_prev = __builtin_readcyclecounter();
#define LZ4_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); if (tid == 0) { atomicAdd(&g_lz4_prof[i], t_ - z_prev); } z_prev = t_; }
#else
#define LZ4_T(i)
#endif
	// ---- A. stage the chunk, clear the count table (all waves) ------------------------------------------------------
	{
		const uint32_t nvec = (((uintptr_t)src & 15u) == 0) ? (n & ~15u) : 0u;
		for (uint32_t i = tid * 16u; i < nvec; i += 4096u) { *reinterpret_cast<uint4*>(s_data + i) = *reinterpret_cast<const uint4*>(src + i); }
		for (uint32_t i = nvec + tid; i < n; i += 256u) { s_data[i] = src[i]; }
#ifdef LZ_TBL_GLOBAL
		for (uint32_t i = n + tid; i < 4096u; i += 256u) { s_data[i] = 0; }
		for (uint32_t i = tid * 8u; i < LZ_TBL; i += 2048u) { *reinterpret_cast<uint4*>(s_cnt + i) = make_uint4(0, 0, 0, 0); }
#else
		for (uint32_t i = n + tid; i < 4096u + 32u; i += 256u) { s_data[i] = 0; }
		for (uint32_t i = tid * 8u; i < LZ_TBL; i += 2048u) { *reinterpret_cast<uint4*>(s_cnt + i) = make_uint4(0, 0, 0, 0); }
		if (tid < LZ4_NSEG) { s_prog[tid] = 0; s_used[tid] = 0; }
		if (tid == 0) { s_segctr = 0; }
#endif
	}
	__syncthreads();
	LZ4_T(0)
	// ---- B. position-sorted buckets (wave 0; see the kernel above) ---------------------------------------------------
	// histogram (all waves: the counts commute) -> exclusive scan (bucket starts) -> ordered scatter through the start
	// cursors (wave 0): the returning atomic hands out the slots of a bucket in position order (batches ascending, lane
	// order inside a batch) and leaves every cursor at its bucket's END, which is what the parse looks up. (No
	// per-position ranks to keep in registers: this kernel wants 8 blocks of 4 waves per CU.)
	{
		const uint32_t nb = (n + 63u) >> 6;
		for (uint32_t b = wv; b < nb; b += 4u) {
			const uint32_t p = b * 64u + lane;
			if (p + 2u < n) {
				const uint32_t h = lz_hash(lds_ld32(s_data, p) & 0xFFFFFFu);
				atomicAdd(reinterpret_cast<uint32_t*>(s_cnt) + (h >> 1), (h & 1u) ? 0x10000u : 1u);
			}
		}
	}
	__syncthreads();
	if (wv == 0) {
		const uint32_t nb = (n + 63u) >> 6;
		wave_fence();
		{
			uint32_t run = 0;
			for (uint32_t k = 0; k < LZ_TBL / 512u; ++k) {
				uint4 a = reinterpret_cast<uint4*>(s_cnt)[k * 64u + lane];
				uint32_t w[4] = { a.x, a.y, a.z, a.w };
				uint32_t sum = 0;
				#pragma unroll
				for (int i = 0; i < 4; ++i) { sum += (w[i] & 0xFFFFu) + (w[i] >> 16); }
				const uint32_t incl = wave_incl_scan_add_u32(sum);
				uint32_t r = run + incl - sum;
				#pragma unroll
				for (int i = 0; i < 4; ++i) { const uint32_t lo = w[i] & 0xFFFFu, hi = w[i] >> 16; w[i] = r | ((r + lo) << 16); r += lo + hi; }
				reinterpret_cast<uint4*>(s_cnt)[k * 64u + lane] = make_uint4(w[0], w[1], w[2], w[3]);
				run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
			}
		}
		wave_fence();
		// 8 batches per round: their 8 atomics are issued back to back (DS operations of a wave execute in order; the
		// wavefront-scope form keeps the compiler from waiting for each one)
		for (uint32_t b0 = 0; b0 < nb; b0 += 8u) {
			uint32_t h[8], old[8];
			#pragma unroll
			for (int j = 0; j < 8; ++j) { h[j] = lz_hash(lds_ld32(s_data, ((b0 + j) * 64u + lane) & 4095u) & 0xFFFFFFu); }
			#pragma unroll
			for (int j = 0; j < 8; ++j) {
				old[j] = lz_ordered_add<serial>(reinterpret_cast<uint32_t*>(s_cnt) + (h[j] >> 1), (h[j] & 1u) ? 0x10000u : 1u, (b0 + j) * 64u + lane + 2u < n, lane);
			}
			#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const uint32_t p = (b0 + j) * 64u + lane;
				if (p + 2u < n) { s_bucket[(h[j] & 1u) ? old[j] >> 16 : old[j] & 0xFFFFu] = (uint16_t)p; }
			}
		}
	}
	__syncthreads();
#ifdef LZ_TBL_GLOBAL
	// the table leaves LDS: 8 KiB behind the chunk's image in its scratch slot (same CU writes and reads it: L2-resident), the parse records take its place
	uint16_t* __restrict__ const g_tbl = reinterpret_cast<uint16_t*>(img + 4352u);
	for (uint32_t i = tid * 8u; i < LZ_TBL; i += 2048u) { *reinterpret_cast<uint4*>(g_tbl + i) = *reinterpret_cast<const uint4*>(s_cnt + i); }
	__syncthreads();
	if (tid < LZ4_NSEG) { s_prog[tid] = 0; s_used[tid] = 0; }
	if (tid == 0) { s_segctr = 0; }
	__syncthreads();
	const uint16_t* const tbl = g_tbl;
#else
	const uint16_t* const tbl = s_cnt;
#endif
	LZ4_T(1)

	// ---- C1. speculative parse of my segment ------------------------------------------------------------------------
	const uint32_t nw = (n + 63u) >> 6;
	const uint32_t w0 = wv * 16u, w1 = (w0 + 16u < nw) ? w0 + 16u : nw;
	// one window: parse, record. Returns the parse position after it.
#define LZ4_WINDOW(w_, entry_, cur_out) { \
		const uint32_t wb_ = (w_) * 64u; const uint32_t we_ = (wb_ + 64u < n) ? wb_ + 64u : n; \
		if ((entry_) >= we_) { if (lane == 0) { s_tok[w_] = 0; s_mat[w_] = 0; } cur_out = (entry_); } \
		else { \
			LzWin r_; cur_out = lz_window(s_data, tbl, s_bucket, n, lane, wb_, (entry_), r_); \
			if ((r_.matchmask >> lane) & (u64)1) { \
				const uint32_t p_ = wb_ + lane, best_ = r_.key >> 12; \
				s_ptok[w_][popc_below(r_.matchmask)] = (uint16_t)(((p_ - (4095u - (r_.key & 0xFFFu)) - 1u) << r_.shift) | (best_ - 3u)); \
			} \
			if (lane == 0) { s_tok[w_] = r_.tokmask; s_mat[w_] = r_.matchmask; } \
		} }
	// the segments are handed out in order (with four of them: one per wave)
	for (;;) {
		uint32_t seg = 0;
		if (lane == 0) { seg = atomicAdd(&s_segctr, 1u); }
		seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg);
		const uint32_t a0 = lz4_seg_start(seg);
		if (a0 >= nw) { break; }
		const uint32_t a1 = (lz4_seg_start(seg + 1u) < nw) ? lz4_seg_start(seg + 1u) : nw;
		uint32_t entry = a0 * 64u;                                 // (exact for segment 0)
		for (uint32_t w = a0; w < a1; ++w) {
			uint32_t cur;
			LZ4_WINDOW(w, entry, cur)
			entry = cur;
			if (lane == 0) { s_endc[w] = (uint16_t)cur; }
			wave_fence();
			if (lane == 0) { __hip_atomic_store(&s_prog[seg], w + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
		}
		// repair the seam behind my segment: continue with my position (true unless my own segment never re-synchronised)
		// into the next segment until the position after a window equals the recorded one
		if (a1 < nw) {
			if (lane == 0) { s_used[seg + 1u] = entry; }
			if (entry != a1 * 64u) {
				const uint32_t wlim = (lz4_seg_start(seg + 2u) < nw) ? lz4_seg_start(seg + 2u) : nw;
				for (uint32_t w = a1; w < wlim; ++w) {
					while (__hip_atomic_load(&s_prog[seg + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= w) { __builtin_amdgcn_s_sleep(2); }
					wave_fence();
					const uint32_t spec = s_endc[w];
					uint32_t cur;
					LZ4_WINDOW(w, entry, cur)
					entry = cur;
					if (lane == 0) { s_endc[w] = (uint16_t)cur; }
					if (cur == spec) { break; }
				}
			}
		}
	}
	LZ4_T(2)
	LZ4_T(3)
	__syncthreads();
	LZ4_T(4)
	// ---- C3. (wave 0) cascade check, then tokens / bytes before every window -----------------------------------------
	if (wv == 0) {
		for (uint32_t j = 1; j < LZ4_NSEG && lz4_seg_start(j) < nw; ++j) {
			uint32_t e2 = s_endc[lz4_seg_start(j) - 1u];
			if (e2 == s_used[j]) { continue; }                        // (else a repair ran through the whole of segment j-1: rare)
			uint32_t wl = lz4_seg_start(j);                           // first window NOT walked
			for (uint32_t w = lz4_seg_start(j); w < nw; ++w) {
				const uint32_t spec = s_endc[w];
				uint32_t cur;
				LZ4_WINDOW(w, e2, cur)
				e2 = cur;
				if (lane == 0) { s_endc[w] = (uint16_t)cur; }
				wave_fence();
				wl = w + 1u;
				if (cur == spec) { break; }
			}
			// every seam this walk crossed is consistent now
			if (lane == 0) { for (uint32_t jj = j + 1u; jj < LZ4_NSEG && lz4_seg_start(jj) < wl; ++jj) { s_used[jj] = s_endc[lz4_seg_start(jj) - 1u]; } }
			wave_fence();
		}
	}
	__syncthreads();                                              // s_cnt is dead from here on: prefixes and flags take its place
	for (uint32_t i = tid; i < 512u; i += 256u) { s_flagacc[i] = 0; }
	if (wv == 0) {
		const u64 tm = lane < nw ? s_tok[lane] : (u64)0, mk = lane < nw ? s_mat[lane] : (u64)0;
		const uint32_t nt = (uint32_t)__popcll(tm), ns = nt + (uint32_t)__popcll(mk);
		const uint32_t ti = wave_incl_scan_add_u32(nt), si = wave_incl_scan_add_u32(ns);
		s_T[lane] = (uint16_t)(ti - nt); s_S[lane] = (uint16_t)(si - ns);
		if (lane == 63u) { s_total[0] = ti; s_total[1] = si; }
	}
	__syncthreads();
	LZ4_T(5)
	const uint32_t T = s_total[0], S = s_total[1];
	const uint32_t csize = ((T + 7u) >> 3) + S;
	uint32_t total;
	if (csize < n) {
		// ---- D1. emission of my segment: pos(t) = 2 (header) + (t div 8 + 1) + sum size(u<t) --------------------------
		for (uint32_t w = w0; w < w1; ++w) {
			const u64 tokmask = s_tok[w];
			if (tokmask == 0) { continue; }
			const u64 matchmask = s_mat[w];
			const bool is_tok = (tokmask >> lane) & (u64)1, is_m = (matchmask >> lane) & (u64)1;
			const uint32_t tb = popc_below(tokmask), mbl = popc_below(matchmask);
			const uint32_t t = (uint32_t)s_T[w] + tb;
			const uint32_t pos = 3u + (t >> 3) + (uint32_t)s_S[w] + tb + mbl;
			if (is_tok) {
				if ((t & 7u) == 0) { s_flagpos[t >> 3] = (uint16_t)(pos - 1u); }
				if (is_m) {
					const uint32_t tok = s_ptok[w][mbl];
					img[pos] = (uint8_t)tok; img[pos + 1u] = (uint8_t)(tok >> 8);
					atomicOr(&s_flagacc[t >> 3], 1u << (t & 7u));
				} else { img[pos] = s_data[w * 64u + lane]; }
			}
		}
		__syncthreads();
		for (uint32_t g = tid; g < ((T + 7u) >> 3); g += 256u) { img[s_flagpos[g]] = (uint8_t)s_flagacc[g]; }
		if (tid == 0) { img[0] = (uint8_t)(0xB000u | (csize - 1u)); img[1] = (uint8_t)((0xB000u | (csize - 1u)) >> 8); }
		total = 2u + csize;
	} else {
		const uint32_t hdr = 0x3000u | (n - 1u);
		for (uint32_t k = tid; k < (n + 2u + 3u) / 4u; k += 256u) {
			reinterpret_cast<uint32_t*>(img)[k] = (k == 0) ? (hdr | (ld16(s_data) << 16)) : ld32(s_data + 4u * k - 2u);
		}
		total = 2u + n;
	}
	if (tid == 0) { slot_size[c] = total; }
	LZ4_T(6)
#undef LZ4_WINDOW
}

static int g_lznt1_mode = 0;                                     // 0 = default, 1 = one wave per chunk, 2 = four waves per chunk (tests)
void set_lznt1_mode(int mode) { g_lznt1_mode = mode; }
void launch_lznt1_chunks(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint8_t* slots, uint32_t* slot_size)
{
	if (bt.n_chunks == 0) { return; }
	const int mode = g_lznt1_mode ? g_lznt1_mode : 2;                 // four waves per chunk: 1.43 vs 1.74 ms on the headline workload
	const bool serial = serial_atomics_on_current_device();
	if (mode == 1) {
		if (serial) { hipLaunchKernelGGL(lznt1_chunk_kernel<true>, dim3(bt.n_chunks), dim3(64), 0, st, d_in, bt, slots, slot_size); }
		else { hipLaunchKernelGGL(lznt1_chunk_kernel<false>, dim3(bt.n_chunks), dim3(64), 0, st, d_in, bt, slots, slot_size); }
	} else {
		if (serial) { hipLaunchKernelGGL(lznt1_chunk4_kernel<true>, dim3(bt.n_chunks), dim3(256), 0, st, d_in, bt, slots, slot_size); }
		else { hipLaunchKernelGGL(lznt1_chunk4_kernel<false>, dim3(bt.n_chunks), dim3(256), 0, st, d_in, bt, slots, slot_size); }
	}
}

} // namespace msc

// lznt1.hip -- LZNT1 one-shot compression for gfx950 (MI355X), bit-exact with the reference CPU encoder.
//
// Replaces: lznt1_compress / lznt1_compress_chunk (/root/reference/src/lznt1_compress.cpp:233-273, :49-94) and
// LZNT1Dictionary::Fill/Find (/root/reference/include/mscomp/LZNT1Dictionary.h:93-106, :114-143).
//
// One 256-thread block (4 wavefronts) = one 4 KiB chunk, everything staged in LDS (33 KiB -> 4 blocks / CU):
//   A. coalesced 16 B/thread load of the chunk into LDS;
//   B. dictionary = the reference's per-key position arrays, as ONE position-sorted bucket array in LDS built by a
//      stable counting sort on a 12-bit hash of the 3-byte key: rank[p] = #earlier positions with the same hash is
//      computed in 64 ascending batches (wave w owns hash class h&3==w, so the four waves never touch the same bin;
//      intra-batch conflicts are resolved with ballots), an exclusive scan turns counts into bucket starts, and a
//      scatter fills bucket[start+rank] = p;
//   C. Find for every position (a pure function of (chunk, position)): each lane scans the OLDEST 8 candidates of
//      its position itself (4 loads in flight, early exit at max_len exactly like the reference), positions with
//      more candidates are then finished cooperatively: 64 lanes take 64 candidates at a time, wave-max of
//      (len, -position) = "longest, oldest on ties";
//   D. wave 0 runs the greedy parse on the scalar unit over per-window ballot masks (one step per MATCH; literal
//      runs are skipped with s_ff1) and places tokens/flag bytes with the closed form
//      pos(t) = (t div 8 + 1) + sum size(u<t)  (mbcnt prefix popcounts); flag bits are OR-ed into LDS;
//   E. the chunk image (2-byte header + payload, or the raw chunk) goes to a 16 B aligned scratch slot with
//      16 B/thread stores; util.hip concatenates slots into the caller's buffer.
#include "common.h"
#include "kernels.h"

namespace msc {

#ifdef LZ_PROFILE   // dev-only phase timers (s_memtime cycles summed over blocks); not in the production build
__device__ unsigned long long g_lz_prof[16];
#define LZ_T(i) if (tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_lz_prof[i], t_ - t_prev); t_prev = t_; }
#define LZ_T0   unsigned long long t_prev = __builtin_readcyclecounter();
#else
#define LZ_T(i)
#define LZ_T0
#endif

#define LZ_TBL_BITS 12
#define LZ_TBL      (1u << LZ_TBL_BITS)
#define LZ_SELF     8u       // candidates each lane scans by itself before the wave cooperates

__device__ __forceinline__ uint32_t lz_hash(uint32_t key24) { return (key24 * 0x9E3779B1u) >> (32 - LZ_TBL_BITS); }

// position-dependent split of the 16-bit match token (lznt1_compress.cpp:51,66) -- pure function of pos
__device__ __forceinline__ uint32_t lz_shift(uint32_t pos)
{
	return pos <= 16u ? 12u : 12u - ((32u - (uint32_t)__builtin_clz(pos - 1u)) - 4u);
}

// racing LDS accesses between lanes of one wave: relaxed wavefront-scope atomics (plain ds_read/ds_write in the ISA;
// DS operations of a wave execute in program order) so the compiler neither forwards nor reorders them.
__device__ __forceinline__ void     wst16(uint16_t* p, uint32_t v) { __hip_atomic_store(p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ uint32_t wld16(uint16_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ void     wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// max over the 64 lanes (DPP: row_shr 1,2,4,8 then row_bcast 15/31), result broadcast from lane 63
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#define MSC_DPP_MAX(ctrl, rmask) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false); v = o > v ? o : v; }
	MSC_DPP_MAX(0x111, 0xf) MSC_DPP_MAX(0x112, 0xf) MSC_DPP_MAX(0x114, 0xf) MSC_DPP_MAX(0x118, 0xf)
	MSC_DPP_MAX(0x142, 0xa) MSC_DPP_MAX(0x143, 0xc)
#undef MSC_DPP_MAX
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// length of the common prefix of d[q..] and d[p..], limited to maxlen; x3 = (byte 3 of q) ^ (byte 3 of p)
__device__ __forceinline__ uint32_t lz_lcp(const uint8_t* d, uint32_t q, uint32_t p, uint32_t maxlen, uint32_t x3)
{
	if (x3) { return 3u; }
	uint32_t l = 4;
	while (l < maxlen) {
		const uint32_t x = ld32(d + q + l) ^ ld32(d + p + l);
		if (x) { l += (uint32_t)__builtin_ctz(x) >> 3; break; }
		l += 4;
	}
	return l < maxlen ? l : maxlen;
}

__global__ __launch_bounds__(256) void lznt1_chunk_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                         uint8_t* __restrict__ slots, uint32_t* __restrict__ slot_size)
{
	__shared__ __attribute__((aligned(16))) uint8_t  s_data[4096 + 32];
	__shared__ __attribute__((aligned(16))) uint16_t s_cnt[LZ_TBL];       // counts -> bucket starts
	__shared__ __attribute__((aligned(16))) uint16_t s_bucket[4096];      // positions sorted by (hash, position)
	__shared__ __attribute__((aligned(16))) uint16_t s_idx[4096];         // rank -> index in s_bucket -> match token
	__shared__ __attribute__((aligned(16))) uint8_t  s_out[LZNT1_SLOT];
	__shared__ uint32_t s_mbits[128];                                      // "position has a match" bits
	__shared__ uint32_t s_grpflag[16];
	__shared__ uint32_t s_wsum[4];
	__shared__ uint32_t s_total;

	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	const uint32_t c = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, c);
	const u64 coff = (u64)(c - bt.chunk_prefix[u]) * 4096u;
	const u64 left = bt.in_len[u] - coff;
	const uint32_t n = left < 4096u ? (uint32_t)left : 4096u;
	const uint8_t* __restrict__ src = d_in + bt.in_off[u] + coff;

	LZ_T0
	// ---- A. stage the chunk, clear the count table -----------------------------------------------------------
	{
		const uint32_t nvec = (((uintptr_t)src & 15u) == 0) ? (n & ~15u) : 0u;
		for (uint32_t i = tid * 16u; i < nvec; i += 4096u) {
			*reinterpret_cast<uint4*>(s_data + i) = *reinterpret_cast<const uint4*>(src + i);
		}
		for (uint32_t i = nvec + tid; i < n; i += 256u) { s_data[i] = src[i]; }
		for (uint32_t i = n + tid; i < 4096u + 32u; i += 256u) { s_data[i] = 0; }
		for (uint32_t i = tid * 8u; i < LZ_TBL; i += 2048u) { *reinterpret_cast<uint4*>(s_cnt + i) = make_uint4(0, 0, 0, 0); }
	}
	__syncthreads();
	LZ_T(0)

	// ---- B1. rank[p] = number of earlier positions with the same hash (wave w owns hash class w) ---------------
	const uint32_t nb = (n + 63u) >> 6;
	for (uint32_t b = 0; b < nb; ++b) {
		const uint32_t p = b * 64u + lane;
		const uint32_t h = lz_hash(ld32(s_data + p) & 0xFFFFFFu);
		const bool mine = (p + 2u < n) && ((h & 3u) == wv);
		uint32_t old = 0;
		if (mine) { old = wld16(&s_cnt[h]); wst16(&s_bucket[h], lane); }   // s_bucket doubles as conflict detector
		wave_fence();
		const bool loser = mine && wld16(&s_bucket[h]) != lane;
		uint32_t rank = old, newcnt = old + 1u;
		bool writer = mine;
		u64 lm = __ballot(loser);
		while (lm) {                                            // one iteration per hash value owned by >1 lane
			const uint32_t hh = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)ctz64(lm));
			const bool grp = mine && h == hh;
			const u64 g = __ballot(grp);
			if (grp) { rank = old + popc_below(g); newcnt = old + (uint32_t)__popcll(g); writer = (lane == ctz64(g)); }
			lm &= ~g;
		}
		if (mine) { s_idx[p] = (uint16_t)rank; if (writer) { wst16(&s_cnt[h], newcnt); } }
		wave_fence();
	}
	__syncthreads();
	LZ_T(1)

	// ---- B2. exclusive scan of the 4096 counts -> bucket starts (16 bins per thread) ---------------------------
	{
		uint4 a = reinterpret_cast<uint4*>(s_cnt)[tid * 2u], b = reinterpret_cast<uint4*>(s_cnt)[tid * 2u + 1u];
		uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
		uint32_t sum = 0;
		#pragma unroll
		for (int k = 0; k < 8; ++k) { sum += (w[k] & 0xFFFFu) + (w[k] >> 16); }
		uint32_t incl = sum;
		#pragma unroll
		for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) { incl += o; } }
		if (lane == 63) { s_wsum[wv] = incl; }
		__syncthreads();
		uint32_t run = incl - sum;
		for (uint32_t k = 0; k < wv; ++k) { run += s_wsum[k]; }
		#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const uint32_t lo = w[k] & 0xFFFFu, hi = w[k] >> 16;
			w[k] = run | ((run + lo) << 16);
			run += lo + hi;
		}
		reinterpret_cast<uint4*>(s_cnt)[tid * 2u] = make_uint4(w[0], w[1], w[2], w[3]);
		reinterpret_cast<uint4*>(s_cnt)[tid * 2u + 1u] = make_uint4(w[4], w[5], w[6], w[7]);
	}
	__syncthreads();

	// ---- B3. scatter: bucket[start[h] + rank[p]] = p ------------------------------------------------------------
	for (uint32_t p = tid; p + 2u < n; p += 256u) {
		const uint32_t h = lz_hash(ld32(s_data + p) & 0xFFFFFFu);
		const uint32_t a = (uint32_t)s_cnt[h] + (uint32_t)s_idx[p];
		s_bucket[a] = (uint16_t)p;
		s_idx[p] = (uint16_t)a;
	}
	__syncthreads();
	LZ_T(2)

	// ---- C. Find for every position: wave w takes windows w, w+4, ... -------------------------------------------
	const uint32_t nw = (n + 63u) >> 6;
	for (uint32_t w = wv; w < nw; w += 4u) {
		const uint32_t p = w * 64u + lane;
		const uint32_t own4 = ld32(s_data + p);
		const uint32_t shift = lz_shift(p);
		uint32_t maxlen = 0, s = 0, cnt = 0;
		if (p > 0 && p + 3u <= n) {
			const uint32_t mask3 = (1u << shift) + 2u;
			maxlen = (n - p < mask3) ? n - p : mask3;
			s = s_cnt[lz_hash(own4 & 0xFFFFFFu)];
			cnt = (uint32_t)s_idx[p] - s;
		}
		// phase 1: the oldest LZ_SELF candidates, in order, early exit at maxlen (LZNT1Dictionary.h:124-135)
		uint32_t key = 0;                                       // (len << 12) | (4095 - q): larger = longer, then older
		bool done = (cnt == 0);
		const uint32_t self_n = cnt < LZ_SELF ? cnt : LZ_SELF;
		for (uint32_t j = 0; j < LZ_SELF; j += 4u) {
			if (j >= self_n || done) { continue; }
			uint32_t q[4], x[4];
			#pragma unroll
			for (int k = 0; k < 4; ++k) { q[k] = (j + k < self_n) ? (uint32_t)s_bucket[s + j + k] : p; }
			#pragma unroll
			for (int k = 0; k < 4; ++k) { x[k] = ld32(s_data + q[k]) ^ own4; }
			#pragma unroll
			for (int k = 0; k < 4; ++k) {
				if (!done && j + k < self_n && (x[k] & 0xFFFFFFu) == 0) {
					const uint32_t l = lz_lcp(s_data, q[k], p, maxlen, x[k] >> 24);
					if (l > (key >> 12)) { key = (l << 12) | (4095u - q[k]); if (l == maxlen) { done = true; } }
				}
			}
		}
		// phase 2: positions with more candidates are finished by the whole wave, 64 candidates per step
		const uint32_t rem = done ? 0u : cnt - self_n;
		u64 hm = __ballot(rem != 0);
		while (hm) {
			const int L = (int)ctz64(hm);
			hm &= hm - 1;
			const uint32_t pL = w * 64u + (uint32_t)L;
			const uint32_t sL = (uint32_t)__builtin_amdgcn_readlane((int)s, L) + LZ_SELF;
			const uint32_t remL = (uint32_t)__builtin_amdgcn_readlane((int)rem, L);
			const uint32_t maxL = (uint32_t)__builtin_amdgcn_readlane((int)maxlen, L);
			const uint32_t ownL = (uint32_t)__builtin_amdgcn_readlane((int)own4, L);
			uint32_t kbest = (uint32_t)__builtin_amdgcn_readlane((int)key, L);
			for (uint32_t base = 0; base < remL; base += 64u) {
				uint32_t k2 = 0;
				if (base + lane < remL) {
					const uint32_t q = s_bucket[sL + base + lane];
					const uint32_t x = ld32(s_data + q) ^ ownL;
					if ((x & 0xFFFFFFu) == 0) { k2 = (lz_lcp(s_data, q, pL, maxL, x >> 24) << 12) | (4095u - q); }
				}
				const uint32_t m = wave_max_u32(k2);
				if ((m >> 12) > (kbest >> 12)) { kbest = m; }    // strictly longer only: older blocks win ties
				if ((kbest >> 12) == maxL) { break; }             // oldest candidate reaching max_len: stop
			}
			if ((int)lane == L) { key = kbest; }
		}
		// result -> token (lznt1_compress.cpp:72); match bit per position
		const uint32_t best = key >> 12;
		const bool is_match = best >= 3u;
		if (is_match) { s_idx[p] = (uint16_t)(((p - (4095u - (key & 0xFFFu)) - 1u) << shift) | (best - 3u)); }
		const u64 mb = __ballot(is_match);
		if (lane == 0) { s_mbits[2u * w] = (uint32_t)mb; s_mbits[2u * w + 1u] = (uint32_t)(mb >> 32); }
	}
	__syncthreads();
	LZ_T(3)

	// ---- D. greedy parse + emit (wave 0) -------------------------------------------------------------------------
	if (wv == 0) {
		uint32_t entry = 0, T = 0, S = 0;
		bool raw = false;
		for (uint32_t w = 0; w < nw; ++w) {
			const uint32_t wbase = w * 64u;
			const uint32_t wend = (wbase + 64u < n) ? wbase + 64u : n;
			if (entry >= wend) { continue; }                     // window wholly covered by a match
			const uint32_t p = wbase + lane;
			const uint32_t shift = lz_shift(p);
			const uint32_t tok = s_idx[p];
			const uint32_t mlen = (tok & ((1u << shift) - 1u)) + 3u;
			const u64 mm = ((u64)s_mbits[2u * w] | ((u64)s_mbits[2u * w + 1u] << 32));
			u64 tokmask = 0, matchmask = 0;
			uint32_t cur = entry;
			while (cur < wend) {
				const uint32_t rel = cur - wbase;
				const u64 rest = mm >> rel;
				if (rest == 0) { tokmask |= (~(u64)0) << rel; cur = wend; break; }
				const uint32_t j = ctz64(rest);
				const uint32_t mpos = rel + j;
				tokmask |= ((((u64)2) << j) - (u64)1) << rel;   // j literals + the match start
				matchmask |= ((u64)1) << mpos;
				cur = wbase + mpos + (uint32_t)__builtin_amdgcn_readlane((int)mlen, (int)mpos);
			}
			if (wend - wbase < 64u) { tokmask &= (((u64)1) << (wend - wbase)) - (u64)1; }
			entry = cur;

			// emit: pos(t) = 2 (header) + (t div 8 + 1) + sum size(u<t)
			const bool is_tok = (tokmask >> lane) & (u64)1;
			const bool is_m = (matchmask >> lane) & (u64)1;
			const uint32_t tb = popc_below(tokmask), mbl = popc_below(matchmask);
			const uint32_t t = T + tb;
			const uint32_t pos = 3u + (t >> 3) + S + tb + mbl;
			if (is_tok) {
				if ((t & 7u) == 0) { s_out[pos - 1u] = 0; s_grpflag[(t >> 3) & 15u] = pos - 1u; }
				if (is_m) { s_out[pos] = (uint8_t)tok; s_out[pos + 1u] = (uint8_t)(tok >> 8); }
				else { s_out[pos] = s_data[p]; }
			}
			wave_fence();
			if (is_m) {
				const uint32_t fp = __hip_atomic_load(&s_grpflag[(t >> 3) & 15u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				atomicOr(reinterpret_cast<uint32_t*>(s_out + (fp & ~3u)), (1u << (t & 7u)) << ((fp & 3u) * 8u));
			}
			wave_fence();
			const uint32_t nt = (uint32_t)__popcll(tokmask), nm = (uint32_t)__popcll(matchmask);
			T += nt; S += nt + nm;
			if (((T + 7u) >> 3) + S >= n) { raw = true; break; }  // running size reached n (:85-86) => store raw
		}
		const uint32_t csize = ((T + 7u) >> 3) + S;
		const bool compressed = !raw && csize < n;
		if (lane == 0) {
			if (compressed) { st16(s_out, 0xB000u | (csize - 1u)); }
			s_total = compressed ? 2u + csize : 0u;
		}
	}
	__syncthreads();
	LZ_T(4)

	// ---- E. chunk image -> scratch slot --------------------------------------------------------------------------
	uint32_t total = s_total;
	if (total == 0) {                                           // raw chunk: header 0x3000 | (n-1), then the bytes
		const uint32_t hdr = 0x3000u | (n - 1u);
		for (uint32_t k = tid; k < (n + 2u + 3u) / 4u; k += 256u) {
			reinterpret_cast<uint32_t*>(s_out)[k] = (k == 0) ? (hdr | (ld16(s_data) << 16)) : ld32(s_data + 4u * k - 2u);
		}
		total = 2u + n;
		__syncthreads();
	}
	uint8_t* __restrict__ slot = slots + (u64)c * LZNT1_SLOT;
	for (uint32_t i = tid * 16u; i < total; i += 4096u) {
		*reinterpret_cast<uint4*>(slot + i) = *reinterpret_cast<const uint4*>(s_out + i);
	}
	if (tid == 0) { slot_size[c] = total; }
	LZ_T(5)
}
#ifdef LZ_PROFILE
extern "C" void mscomp_amd_debug_lz_prof(unsigned long long* out, int reset)
{
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lz_prof), sizeof(unsigned long long) * 16);
	if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lz_prof), z, sizeof z); }
}
#endif

void launch_lznt1_chunks(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint8_t* slots, uint32_t* slot_size)
{
	if (bt.n_chunks == 0) { return; }
	hipLaunchKernelGGL(lznt1_chunk_kernel, dim3(bt.n_chunks), dim3(256), 0, st, d_in, bt, slots, slot_size);
}

} // namespace msc

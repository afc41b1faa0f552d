// lznt1_sa.hip -- LZNT1 chunk compression with the reference's OTHER dictionary flavour: the suffix-array dictionary
// (the reference built with MSCOMP_WITH_LZNT1_SA_DICT, /root/reference/include/mscomp/config.h:83-88).
//
// Replaces LZNT1Dictionary (SA version)::Fill / Find (/root/reference/include/mscomp/LZNT1Dictionary_SA.h:404-476) under the same
// chunk loop (/root/reference/src/lznt1_compress.cpp:49-94). Find takes the longest match from the two LEXICOGRAPHIC neighbours of the
// position's suffix that start earlier -- the nearest one before it in suffix-array order, the nearest one after it, the one before
// winning ties -- so the token lengths are those of the default dictionary ("optimal") but the offsets are not: other bytes, and a
// consumer that was built against that flavour can only be served by the same choice.
//
// One block of 256 threads per 4 KiB chunk, everything in LDS (76 KiB: two blocks per CU):
//   A. suffix array by prefix doubling: round 0 sorts the suffixes by their first 6 bytes, round k by (rank of the first k bytes, rank of
//      the next k bytes), k = 6, 12, 24 ..., with a bitonic network over 4096 packed 64-bit elements (key << 12 | suffix); ranks follow
//      from a block scan over "key differs from its left neighbour"; the rounds stop when all ranks differ (2.1 - 3.6 rounds on the corpus,
//      7.2 on nci). Each thread keeps 16 neighbouring slots in registers, so the network's stages with distance <= 8 need no LDS and the
//      others go two at a time: 29 barriers per round instead of 78. The suffix array of a string is unique (a suffix that is a prefix of
//      another sorts first: the reference's theoretical terminator, :360), so this gives SA-IS's array.
//   B. LCP of neighbours in suffix order, 16 bytes per step (the values of calc_lcp, :361-384).
//   C. Find for every position in closed form: "nearest earlier-starting suffix before / after me in suffix-array order" is the
//      previous / next SMALLER VALUE of the array sa[] at my index, and the match length is the minimum LCP over the indices in
//      between (the loops of :436-468 stop early only where that minimum cannot win any more). The first 4 neighbours on each side are
//      walked as the reference walks them (8 + 8 loads in flight at once; most walks end there), longer walks use two min-trees over the
//      suffix-array order (over sa[]: up from my leaf to the first sibling holding a smaller value, down to its nearest leaf; over lcp[]: a
//      range minimum), 12 levels instead of up to 4095 steps; then :431-470's choice: the side before wins ties.
//   D. the chunk loop of lznt1_compress.cpp:49-94: which positions start a token = the walk p -> p + max(len, 1) from 0. Per group of 64
//      positions, pointer doubling (6 x ds_bpermute) gives where a walk entering at each of them leaves the group; a chain of 64
//      look-ups gives every group's entry point; lane w then walks group w. Token and match counts before each position (popcounts +
//      one wave scan) place every token's bytes and flag bit, all lanes emit; a chunk whose image reaches its size is stored raw (:85).
//      util.hip places the images as for the default flavour.
// Cost per chunk (mozilla, cycles): sort 195 K, ranks 25 K, LCP 28 K, trees 7 K, Find 78 K, parse + emit 45 K; LDS-bandwidth and latency bound
// (8 waves per CU), one profiled run. The flavour exists for byte-compatibility with such builds, at roughly a tenth of the default kernel's speed.
#include "common.h"
#include "kernels.h"

namespace msc {

#define SA_NT 256u
#define SA_N  4096u
#define SA_PROBE 4u
struct SaLds {
	__attribute__((aligned(16))) uint8_t data[SA_N + 32];
	u64      el[SA_N + SA_N / 16u];                                           // sort elements of step A; then the two min-trees of step C (2 x 8192 x 16 bit); then the chunk image of step D
	uint16_t rank[SA_N];                                           // ranks of step A; then the match lengths per position
	uint16_t moff[SA_N];                                           // match offsets per position
	uint16_t sa[SA_N], inv[SA_N], lcp[SA_N];
	u64      tok[64], mat[64];                                     // per 64 positions: which start a token / a match token
	uint16_t tpre[64], mpre[64];                                   // tokens / match tokens before the window
	uint32_t flags[128];                                           // the flag bytes, one bit per token
	uint16_t flagpos[512];                                         // where each goes
	uint32_t wsum[4];
	uint32_t distinct;
};

__device__ __forceinline__ uint32_t sa_shift(uint32_t pos) { return pos <= 16u ? 12u : 12u - ((32u - (uint32_t)__builtin_clz(pos - 1u)) - 4u); }

#ifdef SA_PROFILE   // make EXTRA=-DSA_PROFILE: cycles per phase, summed over the chunks (tools/dev/gpu_saprof.py)
__device__ unsigned long long g_sa_prof[8];
extern "C" void mscomp_amd_debug_sa_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sa_prof), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sa_prof), z, 64); }
#define SA_TM(i) { const unsigned long long t_ = __builtin_readcyclecounter(); if (tid == 0) { atomicAdd(&g_sa_prof[i], t_ - sa_prev); } sa_prev = t_; }
#define SA_CN(i, v) { if (tid == 0) { atomicAdd(&g_sa_prof[i], (unsigned long long)(v)); } }
#else
#define SA_TM(i)
#define SA_CN(i, v)
#endif

#define SA_PH(e) ((e) + ((e) >> 4))                            // sort slot -> LDS slot: one slot of padding per 16, so that a thread reading ITS 16 slots does not meet its 63 colleagues in the same banks
__device__ __forceinline__ void sa_cx(u64& a, u64& b, bool up) { const bool sw = (a > b) == up; const u64 x = sw ? b : a, y = sw ? a : b; a = x; b = y; }

// min of m[l .. r) over the leaves of a min-tree with SA_N leaves (heap layout, leaves at SA_N + i)
__device__ __forceinline__ uint32_t sa_range_min(const uint16_t* m, uint32_t l, uint32_t r)
{
	uint32_t res = 0xFFFFu;
	for (l += SA_N, r += SA_N; l < r; l >>= 1, r >>= 1) {
		if (l & 1u) { const uint32_t v = m[l++]; res = v < res ? v : res; }
		if (r & 1u) { const uint32_t v = m[--r]; res = v < res ? v : res; }
	}
	return res;
}

__global__ __launch_bounds__(SA_NT) void lznt1_sa_chunk_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                              uint8_t* __restrict__ slots, uint32_t* __restrict__ slot_size)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t sa_smem[];
	SaLds& L = *reinterpret_cast<SaLds*>(sa_smem);
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	const uint32_t c = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, c);
	const u64 coff = (u64)(c - bt.chunk_prefix[u]) * 4096u;
	const u64 left = bt.in_len[u] - coff;
	const uint32_t n = left < 4096u ? (uint32_t)left : 4096u;
	const uint8_t* __restrict__ src = d_in + bt.in_off[u] + coff;
	uint8_t* __restrict__ img = slots + (u64)c * LZNT1_SLOT;

	uint16_t* const T    = reinterpret_cast<uint16_t*>(L.el);           // [2 x 4096] min-tree over sa[]
	uint16_t* const M    = T + 2u * SA_N;                               // [2 x 4096] min-tree over lcp[]
	uint16_t* const mlen = L.rank;
	uint16_t* const moff = L.moff;
	uint8_t*  const out  = reinterpret_cast<uint8_t*>(L.el);            // chunk payload (< 4096 bytes: a chunk that does not shrink is stored raw)

	for (uint32_t i = tid; i < SA_N + 32u; i += SA_NT) { L.data[i] = i < n ? src[i] : 0; }
	__syncthreads();

#ifdef SA_PROFILE
	unsigned long long sa_prev = __builtin_readcyclecounter();
#endif
	SA_CN(7, 1)
	if (n > 3u) {                                                  // Fill (:404-419): nothing is built for chunks of up to 3 bytes (and nothing is asked then)
		// ---- A. suffix array by prefix doubling ----
		u64 v[16];                                                     // thread t owns sorted slots 16 t .. 16 t + 15
		for (uint32_t k = 0;; k = k ? k << 1 : 6u) {                     // round 0 sorts by the first 6 bytes, round k by (rank, rank of the suffix k further): 6, 12, 24 ... bytes
			#pragma unroll
			for (uint32_t r = 0; r < 16u; ++r) {
				const uint32_t i = tid * 16u + r;
				u64 key = ~(u64)0 >> 12;                                    // (slots behind the chunk sort to the end)
				if (i < n && k == 0) {                                      // 6 bytes in reading order (zeros behind the end), then how many of them exist: the shorter suffix first
					const uint32_t lo = __builtin_bswap32(lds_ld32(L.data, i)), hi = __builtin_bswap32(lds_ld32(L.data, i + 4u)), left = n - i;
					key = ((((u64)lo << 16) | (hi >> 16)) << 3) | (left < 6u ? left : 6u);
				} else if (i < n) { key = ((u64)L.rank[i] << 13) | (i + k < n ? (u64)L.rank[i + k] : 0u); }
				v[r] = (key << 12) | i;
			}
			SA_TM(1) SA_CN(6, 1)
			// bitonic network, ascending. Runs of 16 slots live in one thread's registers: the stages with j <= 8 never touch LDS, the
			// others go through LDS two at a time (4 slots per step), batched so that the LDS round trips overlap.
			#pragma unroll
			for (uint32_t kk = 2; kk <= 16u; kk <<= 1) {
				#pragma unroll
				for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
					#pragma unroll
					for (uint32_t r = 0; r < 16u; ++r) { if (!(r & j)) { sa_cx(v[r], v[r | j], ((tid * 16u + r) & kk) == 0); } }
				}
			}
			#pragma unroll
			for (uint32_t r = 0; r < 16u; ++r) { L.el[tid * 17u + r] = v[r]; }
			__syncthreads();
			for (uint32_t kk = 32; kk <= SA_N; kk <<= 1) {
				uint32_t j = kk >> 1;
				for (; j >= 32u; j >>= 2) {                                 // stages j and j/2 on quads (i, i + j/2, i + j, i + j + j/2)
					const uint32_t h = j >> 1;
					u64 e[4][4];
					uint32_t at[4];
					#pragma unroll
					for (uint32_t q = 0; q < 4u; ++q) {
						const uint32_t t = tid + q * SA_NT;                    // quad number: its bits go around bits h and j of the slot number
						const uint32_t i = ((t & ~(h - 1u)) << 2) | (t & (h - 1u));
						at[q] = i;
						e[q][0] = L.el[SA_PH(i)]; e[q][1] = L.el[SA_PH(i | h)]; e[q][2] = L.el[SA_PH(i | j)]; e[q][3] = L.el[SA_PH(i | j | h)];
					}
					#pragma unroll
					for (uint32_t q = 0; q < 4u; ++q) {
						const uint32_t i = at[q];
						const bool up = (i & kk) == 0;
						sa_cx(e[q][0], e[q][2], up); sa_cx(e[q][1], e[q][3], up);
						sa_cx(e[q][0], e[q][1], up); sa_cx(e[q][2], e[q][3], up);
						L.el[SA_PH(i)] = e[q][0]; L.el[SA_PH(i | h)] = e[q][1]; L.el[SA_PH(i | j)] = e[q][2]; L.el[SA_PH(i | j | h)] = e[q][3];
					}
					__syncthreads();
				}
				if (j == 16u) {                                             // one stage left above the registers
					u64 ea[8], eb[8];
					#pragma unroll
					for (uint32_t q = 0; q < 8u; ++q) {
						const uint32_t t = tid + q * SA_NT, i = ((t & ~15u) << 1) | (t & 15u);
						ea[q] = L.el[SA_PH(i)]; eb[q] = L.el[SA_PH(i | 16u)];
					}
					#pragma unroll
					for (uint32_t q = 0; q < 8u; ++q) {
						const uint32_t t = tid + q * SA_NT, i = ((t & ~15u) << 1) | (t & 15u);
						sa_cx(ea[q], eb[q], (i & kk) == 0);
						L.el[SA_PH(i)] = ea[q]; L.el[SA_PH(i | 16u)] = eb[q];
					}
					__syncthreads();
				}
				#pragma unroll
				for (uint32_t r = 0; r < 16u; ++r) { v[r] = L.el[tid * 17u + r]; }
				const bool up = ((tid * 16u) & kk) == 0;
				#pragma unroll
				for (uint32_t jj = 8; jj > 0; jj >>= 1) {
					#pragma unroll
					for (uint32_t r = 0; r < 16u; ++r) { if (!(r & jj)) { sa_cx(v[r], v[r | jj], up); } }
				}
				#pragma unroll
				for (uint32_t r = 0; r < 16u; ++r) { L.el[tid * 17u + r] = v[r]; }
				__syncthreads();
			}
			SA_TM(0)
			// ranks: 1 + number of key changes before me
			uint32_t flg[16], sum = 0;
			const u64 before = tid ? L.el[tid * 17u - 2u] : (u64)0;      // slot 16 t - 1 (SA_PH)
			#pragma unroll
			for (uint32_t r = 0; r < 16u; ++r) {
				const uint32_t j = tid * 16u + r;
				flg[r] = (j > 0 && j < n && (v[r] >> 12) != ((r ? v[r ? r - 1u : 0u] : before) >> 12)) ? 1u : 0u;
				sum += flg[r];
			}
			const uint32_t incl = wave_incl_scan_add_u32(sum);
			if (lane == 63u) { L.wsum[wv] = incl; }
			__syncthreads();
			uint32_t run = 1u + incl - sum;
			for (uint32_t w = 0; w < wv; ++w) { run += L.wsum[w]; }
			#pragma unroll
			for (uint32_t r = 0; r < 16u; ++r) {
				const uint32_t j = tid * 16u + r;
				run += flg[r];
				if (j < n) { L.rank[(uint32_t)(v[r] & 0xFFFu)] = (uint16_t)run; }
				if (j + 1u == n) { L.distinct = (run == n) ? 1u : 0u; }
			}
			__syncthreads();
			SA_TM(1)
			if (L.distinct || k >= n) { break; }                          // (k >= n cannot happen without distinct ranks: a guard, not a rule)
		}
		#pragma unroll
		for (uint32_t r = 0; r < 16u; ++r) {
			const uint32_t j = tid * 16u + r, p = (uint32_t)(v[r] & 0xFFFu);
			if (j < n) { L.sa[j] = (uint16_t)p; L.inv[p] = (uint16_t)j; }
		}
		__syncthreads();
		// ---- B. LCP of suffix-array neighbours ----
		for (uint32_t j = tid; j < n; j += SA_NT) {
			uint32_t l = 0;
			if (j > 0) {
				const uint32_t a = L.sa[j - 1u], b = L.sa[j], m = n - (a > b ? a : b);
				while (l < m) {
					const uint4 x = lds_ld128(L.data, a + l), y = lds_ld128(L.data, b + l);
					const uint32_t f = first_nz_byte16(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
					l += f;
					if (f < 16u) { break; }
				}
				if (l > m) { l = m; }
			}
			L.lcp[j] = (uint16_t)l;
		}
		__syncthreads();
		SA_TM(2)
		// ---- C. two min-trees over the suffix-array order: T over sa[] (who starts earliest in a range), M over lcp[] ----
		for (uint32_t i = tid; i < SA_N; i += SA_NT) { T[SA_N + i] = i < n ? L.sa[i] : (uint16_t)0xFFFFu; M[SA_N + i] = i < n ? L.lcp[i] : (uint16_t)0xFFFFu; }
		__syncthreads();
		for (uint32_t w = SA_N >> 1; w >= 1u; w >>= 1) {
			for (uint32_t node = w + tid; node < 2u * w; node += SA_NT) {
				const uint32_t a0 = T[2u * node], a1 = T[2u * node + 1u], b0 = M[2u * node], b1 = M[2u * node + 1u];
				T[node] = (uint16_t)(a0 < a1 ? a0 : a1); M[node] = (uint16_t)(b0 < b1 ? b0 : b1);
			}
			__syncthreads();
		}
	}
	SA_TM(3)
	// per position: what Find returns (length 0 = no match), :424-476. "The nearest suffix before me in suffix-array order that starts earlier"
	// (the loop of :436-447) is the previous smaller value of sa[] at my index i, its match length the minimum LCP over the indices in between
	// -- the loop gives up exactly where that minimum is <= 2; the loop of :453-468 is the next smaller value, given up where the minimum cannot
	// beat the first candidate. Both by a walk up and down T, the minima by a range query on M.
	for (uint32_t pos = tid; pos < n; pos += SA_NT) {
		uint32_t len = 0, off = 0;
		if (n > 3u && pos > 0) {
			const uint32_t mask3 = (1u << sa_shift(pos)) + 2u, rem = n - pos, maxlen = rem < mask3 ? rem : mask3;
			if (maxlen >= 3u) {
				const uint32_t i = L.inv[pos];
				uint32_t best = 2, found = 0;
				// the first SA_PROBE neighbours on either side as the reference walks them (all their loads in flight together); the trees only for walks that go further
				uint32_t sb[SA_PROBE], lb[SA_PROBE], sn[SA_PROBE], ln[SA_PROBE];
				#pragma unroll
				for (uint32_t d = 0; d < SA_PROBE; ++d) {
					const bool okb = i >= d + 1u, okn = i + 1u + d < SA_N;
					sb[d] = okb ? (uint32_t)T[SA_N + i - 1u - d] : 0xFFFFu; lb[d] = okb ? (uint32_t)M[SA_N + i - d] : 0u;   // (before index 0: lcp[0] = 0 ends the walk)
					sn[d] = okn ? (uint32_t)T[SA_N + i + 1u + d] : 0xFFFFu; ln[d] = okn ? (uint32_t)M[SA_N + i + 1u + d] : 0u;
				}
				int state = 0;                                                 // 0 = walk on, 1 = found, 2 = given up
				uint32_t m = lb[0];
				if (m <= 2u) { state = 2; } else if (sb[0] < pos) { state = 1; }
				#pragma unroll
				for (uint32_t d = 1; d < SA_PROBE; ++d) {
					if (state == 0) {
						if (lb[d] < m) { m = lb[d]; if (m <= 2u) { state = 2; } }
						if (state == 0 && sb[d] < pos) { state = 1; }
						if (state == 1) { sb[0] = sb[d]; }
					}
				}
				if (state == 1) { best = m; found = sb[0]; }
				else if (state == 0) {
					uint32_t node = SA_N + i, j = SA_N;
					while (node > 1u) {
						if ((node & 1u) && T[node - 1u] < pos) {
							node -= 1u;
							while (node < SA_N) { node = 2u * node + 1u; if (!(T[node] < pos)) { node -= 1u; } }
							j = node - SA_N; break;
						}
						node >>= 1;
					}
					if (j < SA_N) {
						const uint32_t mm = sa_range_min(M, j + 1u, i + 1u);
						if (mm > 2u) { best = mm; found = T[SA_N + j]; }
					}
				}
				state = 0; m = ln[0];
				if (m <= best || i + 1u >= n) { state = 2; } else if (sn[0] < pos) { state = 1; }
				#pragma unroll
				for (uint32_t d = 1; d < SA_PROBE; ++d) {
					if (state == 0) {
						if (ln[d] < m) { m = ln[d]; if (m <= best) { state = 2; } }
						if (state == 0 && sn[d] < pos) { state = 1; }
						if (state == 1) { sn[0] = sn[d]; }
					}
				}
				if (state == 1) { best = m; found = sn[0]; }
				else if (state == 0) {
					uint32_t node = SA_N + i, j = SA_N;
					while (node > 1u) {
						if (!(node & 1u) && T[node + 1u] < pos) {
							node += 1u;
							while (node < SA_N) { node = 2u * node; if (!(T[node] < pos)) { node += 1u; } }
							j = node - SA_N; break;
						}
						node >>= 1;
					}
					if (j < SA_N) {
						const uint32_t mm = sa_range_min(M, i + 1u, j + 1u);
						if (mm > best) { best = mm; found = T[SA_N + j]; }
					}
				}
				if (best > 2u) { len = best > maxlen ? maxlen : best; off = pos - found; }
			}
		}
		mlen[pos] = (uint16_t)len; moff[pos] = (uint16_t)off;
	}
	for (uint32_t g = tid; g < 128u; g += SA_NT) { L.flags[g] = 0; }
	__syncthreads();
	SA_TM(4)
	// ---- D. the chunk loop (lznt1_compress.cpp:55-93). Which positions start a token: one wave, 64 positions at a time, lengths in registers ----
	// (a) where the walk that enters 64 positions at p leaves them: pointer doubling inside each group, one wave per group, no LDS
	{
		uint16_t* const X = L.sa;                                          // (the suffix array has served)
		for (uint32_t w = wv; w * 64u < n; w += SA_NT / 64u) {
			const uint32_t p = w * 64u + lane, wend = w * 64u + 64u;
			const uint32_t ml = p < n ? (uint32_t)mlen[p] : 64u;
			uint32_t nx = p + (ml ? ml : 1u);
			#pragma unroll
			for (uint32_t r = 0; r < 6u; ++r) {
				const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nx < wend ? nx & 63u : lane) << 2), (int)nx);
				if (nx < wend) { nx = o; }
			}
			X[p] = (uint16_t)nx;
		}
		__syncthreads();
		// (b) the entry point of every group: a chain of 64 look-ups
		if (tid == 0) {
			uint32_t e = 0;
			for (uint32_t w = 0; w * 64u < n; ++w) { L.tpre[w] = (uint16_t)e; if (e < w * 64u + 64u) { e = X[e]; } }
		}
		__syncthreads();
	}
	// (c) the tokens of each group: lane w walks group w from its entry point
	if (wv == 0) {
		u64 tok = 0, mat = 0;
		if (lane * 64u < n) {
			const uint32_t wend = lane * 64u + 64u < n ? lane * 64u + 64u : n;
			uint32_t q = L.tpre[lane];
			while (q < wend) {
				const uint32_t l = mlen[q], b = q & 63u;
				tok |= (u64)1 << b;
				if (l) { mat |= (u64)1 << b; q += l; } else { q += 1u; }
			}
			L.tok[lane] = tok; L.mat[lane] = mat;
		}
		// tokens / match tokens before each window, and the payload size: one flag byte per 8 tokens + 1 byte per literal + 2 per match
		const uint32_t tw = (uint32_t)__popcll(tok), mw = (uint32_t)__popcll(mat);
		const uint32_t ti = wave_incl_scan_add_u32(tw), mi = wave_incl_scan_add_u32(mw);
		L.tpre[lane] = (uint16_t)(ti - tw); L.mpre[lane] = (uint16_t)(mi - mw);
		if (lane == 63u) { L.wsum[0] = ti; L.wsum[1] = ((ti + 7u) >> 3) + ti + mi; }
	}
	__syncthreads();
	const uint32_t ntok = L.wsum[0], psize = L.wsum[1];
	const bool raw = psize >= n;                                          // :85: the running size only grows, so "some group ends at or past in_len" is "the last one does" (the capacity test of :85 is made per unit, by the placement kernels)
	if (!raw) {
		for (uint32_t pos = tid; pos < n; pos += SA_NT) {
			const uint32_t w = pos >> 6, b = pos & 63u;
			const u64 tok = L.tok[w], mat = L.mat[w], below = ((u64)1 << b) - 1u;
			if ((tok >> b) & 1u) {
				const uint32_t t = L.tpre[w] + (uint32_t)__popcll(tok & below), mb = L.mpre[w] + (uint32_t)__popcll(mat & below);
				const uint32_t at = (t >> 3) + 1u + t + mb;                    // flag bytes so far + literals + 2 x matches before me
				if ((t & 7u) == 0) { L.flagpos[t >> 3] = (uint16_t)(at - 1u); }
				if ((mat >> b) & 1u) {
					const uint32_t sym = (((uint32_t)moff[pos] - 1u) << sa_shift(pos)) | ((uint32_t)mlen[pos] - 3u);
					out[at] = (uint8_t)sym; out[at + 1u] = (uint8_t)(sym >> 8);
					atomicOr(&L.flags[t >> 5], 1u << (t & 31u));
				} else { out[at] = L.data[pos]; }
			}
		}
		__syncthreads();
		for (uint32_t g = tid; g * 8u < ntok; g += SA_NT) { out[L.flagpos[g]] = (uint8_t)(L.flags[g >> 2] >> ((g & 3u) * 8u)); }
		__syncthreads();
	}
	const uint32_t csize = raw ? 0u : psize;
	if (csize) {
		const uint32_t hdr = 0xB000u | (csize - 1u);
		if (tid == 0) { img[0] = (uint8_t)hdr; img[1] = (uint8_t)(hdr >> 8); }
		for (uint32_t i = tid; i < csize; i += SA_NT) { img[2u + i] = out[i]; }
	} else {
		const uint32_t hdr = 0x3000u | (n - 1u);
		if (tid == 0) { img[0] = (uint8_t)hdr; img[1] = (uint8_t)(hdr >> 8); }
		for (uint32_t i = tid; i < n; i += SA_NT) { img[2u + i] = L.data[i]; }
	}
	if (tid == 0) { slot_size[c] = 2u + (csize ? csize : n); }
	SA_TM(5)
}

void launch_lznt1_sa_chunks(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint8_t* slots, uint32_t* slot_size)
{
	if (bt.n_chunks == 0) { return; }
	static PerDeviceOnce attr;
	if (attr.needed()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lznt1_sa_chunk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SaLds)); attr.done(); }
	hipLaunchKernelGGL(lznt1_sa_chunk_kernel, dim3(bt.n_chunks), dim3(SA_NT), sizeof(SaLds), st, d_in, bt, slots, slot_size);
}

} // namespace msc

// lznt1_sa.hip -- LZNT1 chunk compression with the reference's OTHER dictionary flavour: the suffix-array dictionary
// (the reference built with MSCOMP_WITH_LZNT1_SA_DICT, /root/reference/include/mscomp/config.h:83-88).
//
// Replaces LZNT1Dictionary (SA version)::Fill / Find (/root/reference/include/mscomp/LZNT1Dictionary_SA.h:404-476) under the same
// chunk loop (/root/reference/src/lznt1_compress.cpp:49-94). Find takes the longest match from the two LEXICOGRAPHIC neighbours of the
// position's suffix that start earlier -- the nearest one before it in suffix-array order, the nearest one after it, the one before
// winning ties -- so the token lengths are those of the default dictionary ("optimal") but the offsets are not: other bytes, and a
// consumer that was built against that flavour can only be served by the same choice.
//
// One block of 256 threads per 4 KiB chunk, everything in LDS:
//   A. suffix array by prefix doubling: round k sorts the suffixes by (rank of the first k bytes, rank of the next k bytes) with a
//      bitonic network over 4096 packed 64-bit elements (key << 12 | suffix), ranks follow from a block scan over "key differs from
//      its left neighbour"; the rounds stop when all ranks differ (k >= the longest repeat). The suffix array of a string is unique
//      (a suffix that is a prefix of another sorts first: the reference's theoretical terminator, :360), so this gives SA-IS's array.
//   B. LCP of neighbours in suffix order, 16 bytes per step (the values of calc_lcp, :361-384).
//   C. Find for every position in closed form: "nearest earlier-starting suffix before / after me in suffix-array order" is the
//      previous / next SMALLER VALUE of the array sa[] at my index, and the match length is the minimum LCP over the indices in
//      between (the loops of :436-468 stop early only where that minimum cannot win any more). Both come from one stack scan in
//      each direction (one lane; O(n)), then every lane combines its positions' two candidates exactly as :431-470 does.
//   D. the chunk loop of lznt1_compress.cpp:49-94 by one lane into an LDS image (flag byte per 8 tokens, raw chunk when the running
//      size reaches the input size), copied out by all lanes. util.hip places the images as for the default flavour.
// This flavour exists for completeness, not for speed: 12 505 chunks take about 30 x the default kernel's time.
#include "common.h"
#include "kernels.h"

namespace msc {

#define SA_NT 256u
#define SA_N  4096u
struct SaLds {
	__attribute__((aligned(16))) uint8_t data[SA_N + 32];
	u64      el[SA_N];                                             // sort elements; later aliased: [0,16K) the two stacks of step C, then the match table; [16K,32K) fndb, lena; then the image
	uint16_t rank[SA_N];                                           // ranks of step A; later aliased: lenb
	uint16_t sa[SA_N], inv[SA_N], lcp[SA_N];
	uint16_t fnda[SA_N];                                           // (lenb, fndb, lena, fnda: per suffix-array index, the candidate before / after; length 0 = none)
	uint32_t wsum[4];
	uint32_t distinct;
};

__device__ __forceinline__ uint32_t sa_shift(uint32_t pos) { return pos <= 16u ? 12u : 12u - ((32u - (uint32_t)__builtin_clz(pos - 1u)) - 4u); }

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

	for (uint32_t i = tid; i < SA_N + 32u; i += SA_NT) { L.data[i] = i < n ? src[i] : 0; }
	__syncthreads();

	if (n > 3u) {                                                  // Fill (:404-419): nothing is built for chunks of up to 3 bytes (and nothing is asked then)
		// ---- A. suffix array by prefix doubling ----
		for (uint32_t i = tid; i < SA_N; i += SA_NT) { L.rank[i] = i < n ? (uint16_t)(L.data[i] + 1u) : 0; }
		__syncthreads();
		for (uint32_t k = 1;; k <<= 1) {
			for (uint32_t i = tid; i < SA_N; i += SA_NT) {
				const u64 key = i < n ? (((u64)L.rank[i] << 13) | (i + k < n ? (u64)L.rank[i + k] : 0u)) : ~(u64)0 >> 12;   // (slots behind the chunk sort to the end)
				L.el[i] = (key << 12) | i;
			}
			__syncthreads();
			for (uint32_t kk = 2; kk <= SA_N; kk <<= 1) {                // bitonic network, ascending
				for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
					for (uint32_t t = tid; t < SA_N / 2u; t += SA_NT) {
						const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), l = i | j;   // the pair (i, i ^ j) with bit j of i clear
						const u64 a = L.el[i], b = L.el[l];
						const bool up = (i & kk) == 0;
						if ((a > b) == up) { L.el[i] = b; L.el[l] = a; }
					}
					__syncthreads();
				}
			}
			// ranks: 1 + number of key changes before me (thread t owns sorted slots 16 t .. 16 t + 15)
			uint32_t flg[16], sum = 0;
			#pragma unroll
			for (uint32_t r = 0; r < 16u; ++r) {
				const uint32_t j = tid * 16u + r;
				flg[r] = (j > 0 && j < n && (L.el[j] >> 12) != (L.el[j - 1u] >> 12)) ? 1u : 0u;
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
				if (j < n) { L.rank[(uint32_t)(L.el[j] & 0xFFFu)] = (uint16_t)run; }
				if (j + 1u == n) { L.distinct = (run == n) ? 1u : 0u; }
			}
			__syncthreads();
			if (L.distinct || k >= n) { break; }
		}
		for (uint32_t j = tid; j < n; j += SA_NT) { const uint32_t p = (uint32_t)(L.el[j] & 0xFFFu); L.sa[j] = (uint16_t)p; L.inv[p] = (uint16_t)j; }
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
		// ---- C. previous / next smaller value of sa[] with the minimum LCP in between (one lane; the sort array is free: two stacks) ----
		uint16_t* const lenb = L.rank;
		uint16_t* const fndb = reinterpret_cast<uint16_t*>(L.el) + 2u * SA_N;
		uint16_t* const lena = fndb + SA_N;
		if (tid == 0) {
			uint16_t* const stk = reinterpret_cast<uint16_t*>(L.el);   // stack of suffix-array indices
			uint16_t* const smn = stk + SA_N;                          // minimum LCP between the element below and this one
			uint32_t sp = 0;
			for (uint32_t i = 0; i < n; ++i) {                         // before: :431-447
				const uint32_t v = L.sa[i];
				uint32_t cur = L.lcp[i];
				while (sp && L.sa[stk[sp - 1u]] > v) { const uint32_t m = smn[sp - 1u]; cur = cur < m ? cur : m; --sp; }
				if (sp) { lenb[i] = (uint16_t)cur; fndb[i] = L.sa[stk[sp - 1u]]; } else { lenb[i] = 0; }
				stk[sp] = (uint16_t)i; smn[sp] = (uint16_t)cur; ++sp;
			}
			sp = 0;
			for (uint32_t i = n; i-- > 0;) {                           // after: :450-468
				const uint32_t v = L.sa[i];
				uint32_t cur = i + 1u < n ? (uint32_t)L.lcp[i + 1u] : 0u;
				while (sp && L.sa[stk[sp - 1u]] > v) { const uint32_t m = smn[sp - 1u]; cur = cur < m ? cur : m; --sp; }
				if (sp) { lena[i] = (uint16_t)cur; L.fnda[i] = L.sa[stk[sp - 1u]]; } else { lena[i] = 0; }
				stk[sp] = (uint16_t)i; smn[sp] = (uint16_t)cur; ++sp;
			}
		}
		__syncthreads();
	}
	// per position: what Find returns (length 0 = no match), :424-476
	uint16_t* const mlen = reinterpret_cast<uint16_t*>(L.el);           // [4096]
	uint16_t* const moff = mlen + SA_N;                                 // [4096]
	uint8_t*  const out  = reinterpret_cast<uint8_t*>(moff + SA_N);     // chunk payload being assembled (<= 4096 + 17 bytes; over fndb / lena, which are done with by then)
	const uint16_t* const lenb = L.rank;
	const uint16_t* const fndb = moff + SA_N;
	const uint16_t* const lena = fndb + SA_N;
	for (uint32_t pos = tid; pos < n; pos += SA_NT) {
		uint32_t len = 0, off = 0;
		if (n > 3u && pos > 0) {
			const uint32_t mask3 = (1u << sa_shift(pos)) + 2u, rem = n - pos, maxlen = rem < mask3 ? rem : mask3;
			if (maxlen >= 3u) {
				const uint32_t i = L.inv[pos];
				uint32_t best = 2, found = 0;
				if (lenb[i] > 2u) { best = lenb[i]; found = fndb[i]; }
				if (lena[i] > best) { best = lena[i]; found = L.fnda[i]; }
				if (best > 2u) { len = best > maxlen ? maxlen : best; off = pos - found; }
			}
		}
		mlen[pos] = (uint16_t)len; moff[pos] = (uint16_t)off;
	}
	__syncthreads();
	// ---- D. the chunk loop (lznt1_compress.cpp:55-93), one lane ----
	if (tid == 0) {
		uint32_t in_pos = 0, out_pos = 0;
		bool raw = false;
		while (in_pos < n) {
			uint32_t i = 0, pos = 0, bits = 0;
			uint8_t* const grp = out + out_pos + 1u;
			for (; i < 8u && in_pos < n; ++i) {
				const uint32_t len = mlen[in_pos];
				if (len) {
					const uint32_t sym = ((moff[in_pos] - 1u) << sa_shift(in_pos)) | (len - 3u);
					grp[pos] = (uint8_t)sym; grp[pos + 1u] = (uint8_t)(sym >> 8);
					pos += 2u; bits |= 1u << i; in_pos += len;
				} else { grp[pos++] = L.data[in_pos++]; }
			}
			const uint32_t end = out_pos + 1u + pos;
			if (end >= n) { raw = true; break; }                        // :85 (the capacity test of :85 is made per unit, by the placement kernels)
			out[out_pos] = (uint8_t)bits;
			out_pos = end;
		}
		L.wsum[0] = raw ? 0u : out_pos;
	}
	__syncthreads();
	const uint32_t csize = L.wsum[0];
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
}

void launch_lznt1_sa_chunks(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint8_t* slots, uint32_t* slot_size)
{
	if (bt.n_chunks == 0) { return; }
	static bool attr_set = false;
	if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lznt1_sa_chunk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SaLds)); attr_set = true; }
	hipLaunchKernelGGL(lznt1_sa_chunk_kernel, dim3(bt.n_chunks), dim3(SA_NT), sizeof(SaLds), st, d_in, bt, slots, slot_size);
}

} // namespace msc

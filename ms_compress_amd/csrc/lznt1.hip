// lznt1.hip -- LZNT1 one-shot compression for gfx950 (MI355X), bit-exact with the reference CPU encoder.
//
// Replaces: lznt1_compress / lznt1_compress_chunk (/root/reference/src/lznt1_compress.cpp:233-273, :49-94) and
// LZNT1Dictionary::Fill/Find (/root/reference/include/mscomp/LZNT1Dictionary.h:93-106, :114-143).
//
// Design (one wavefront = one 4 KiB chunk, everything staged in LDS, no block barriers that cost anything):
//   1. coalesced 16 B/lane load of the chunk into LDS;
//   2. dictionary = ASCENDING per-key position lists in LDS (first[hash(3 bytes)] + next[pos]); built in 64
//      reverse batches of 64 positions, one LDS gather + scatter per batch, intra-batch hash conflicts are
//      resolved with wave ballots (so lists are exactly position-ordered: the reference scans candidates
//      oldest-first and the oldest wins ties);
//   3. per 64-position window: every lane runs the exhaustive Find for its own position (longest match,
//      oldest on ties, early exit at max_len) -- a pure function of (chunk, position) --, then the greedy
//      parse walks the window on the scalar unit over the ballot mask of match candidates (one step per
//      MATCH, literal runs are skipped with s_ff1), windows wholly covered by a long match are skipped;
//   4. token/flag-byte placement is the closed form  pos(t) = (t div 8 + 1) + sum size(u<t)  evaluated with
//      mbcnt prefix popcounts; flag bits are OR-ed into LDS; the chunk image (2-byte header + payload) is
//      written to a 16 B aligned scratch slot with 16 B/lane stores;
//   5. a second kernel concatenates the slots of each unit (exclusive scan of slot sizes) into the caller's
//      output, adds the End_of_buffer terminal and reports size/status per unit.
#include "common.h"
#include "kernels.h"

namespace msc {

#define LZ_TBL_BITS 12
#define LZ_TBL      (1u << LZ_TBL_BITS)
#define LZ_NONE     0xFFFFu

__device__ __forceinline__ uint32_t lz_hash(uint32_t key24) { return (key24 * 0x9E3779B1u) >> (32 - LZ_TBL_BITS); }

// position-dependent split of the 16-bit match token (lznt1_compress.cpp:51,66) -- pure function of pos
__device__ __forceinline__ uint32_t lz_shift(uint32_t pos)
{
	return pos <= 16u ? 12u : 12u - ((32u - (uint32_t)__builtin_clz(pos - 1u)) - 4u);
}

__global__ __launch_bounds__(64) void lznt1_chunk_kernel(const uint8_t* __restrict__ d_in, BatchTables bt,
                                                        uint8_t* __restrict__ slots, uint32_t* __restrict__ slot_size)
{
	__shared__ __attribute__((aligned(16))) uint8_t  s_data[4096 + 32];
	__shared__ __attribute__((aligned(16))) uint16_t s_first[LZ_TBL];
	__shared__ __attribute__((aligned(16))) uint16_t s_next[4096];
	__shared__ __attribute__((aligned(16))) uint8_t  s_out[LZNT1_SLOT];
	__shared__ uint32_t s_grpflag[16];

	const uint32_t lane = threadIdx.x;
	const uint32_t c = blockIdx.x;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, c);
	const u64 ubase = bt.in_off[u];
	const u64 coff = (u64)(c - bt.chunk_prefix[u]) * 4096u;
	const u64 left = bt.in_len[u] - coff;
	const uint32_t n = left < 4096u ? (uint32_t)left : 4096u;
	const uint8_t* __restrict__ src = d_in + ubase + coff;

	// ---- 1. stage the chunk in LDS -------------------------------------------------------------------------
	{
		const uint32_t nvec = (((uintptr_t)src & 15u) == 0) ? (n & ~15u) : 0u;
		for (uint32_t i = lane * 16u; i < nvec; i += 1024u) {
			*reinterpret_cast<uint4*>(s_data + i) = *reinterpret_cast<const uint4*>(src + i);
		}
		for (uint32_t i = nvec + lane; i < n; i += 64u) { s_data[i] = src[i]; }
		for (uint32_t i = n + lane; i < 4096u + 32u; i += 64u) { s_data[i] = 0; }
		for (uint32_t i = lane * 8u; i < LZ_TBL; i += 512u) {
			*reinterpret_cast<uint4*>(s_first + i) = make_uint4(~0u, ~0u, ~0u, ~0u);
		}
	}
	__syncthreads();

	// ---- 2. ascending per-key lists (reverse batches) ------------------------------------------------------
	const int nb = (int)((n + 63u) >> 6);
	for (int b = nb - 1; b >= 0; --b) {
		const uint32_t p = (uint32_t)b * 64u + lane;
		const bool valid = p + 2u < n;
		const uint32_t h = lz_hash(ld32(s_data + p) & 0xFFFFFFu);
		uint32_t nx = LZ_NONE;
		if (valid) { nx = s_first[h]; }
		__syncthreads();
		if (valid) { s_first[h] = (uint16_t)p; }
		__syncthreads();
		const bool loser = valid && s_first[h] != p;
		u64 lm = __ballot(loser);
		const uint32_t old = nx;
		while (lm) {                                            // one iteration per hash value with >1 lane
			const uint32_t l = ctz64(lm);
			const uint32_t hh = __builtin_amdgcn_readlane(h, l);
			const bool mine = valid && h == hh;
			const u64 g = __ballot(mine);
			if (mine) {
				const u64 above = (g >> lane) >> 1;                 // same-hash lanes above me
				nx = above ? p + 1u + ctz64(above) : old;
				if (lane == ctz64(g)) { s_first[h] = (uint16_t)p; }  // list head = lowest position
			}
			lm &= ~g;
		}
		if (valid) { s_next[p] = (uint16_t)nx; }
	}
	__syncthreads();

	// ---- 3./4. windows: Find (all lanes) -> scalar greedy walk over matches -> emit -------------------------
	uint32_t entry = 0;           // position where the next token starts
	uint32_t T = 0, S = 0;        // tokens so far, sum of token sizes so far
	bool raw = false;
	const uint32_t nw = (n + 63u) >> 6;
	for (uint32_t w = 0; w < nw; ++w) {
		const uint32_t wbase = w * 64u;
		const uint32_t wend = (wbase + 64u < n) ? wbase + 64u : n;
		if (entry >= wend) { continue; }                         // window wholly covered by a match
		const uint32_t p = wbase + lane;
		const uint32_t own4 = ld32(s_data + p);
		const uint32_t shift = lz_shift(p);
		uint32_t best = 0, boff = 0;
		if (p >= entry && p > 0 && p + 3u <= n) {
			const uint32_t mask3 = (1u << shift) + 2u;
			const uint32_t maxlen = (n - p < mask3) ? n - p : mask3;
			uint32_t q = s_first[lz_hash(own4 & 0xFFFFFFu)];
			while (q < p) {                                      // ascending: oldest first, ends at p itself
				const uint32_t x0 = ld32(s_data + q) ^ own4;
				const uint32_t qn = s_next[q];
				if ((x0 & 0xFFFFFFu) == 0) {
					uint32_t l = 3;
					if (x0 == 0) {
						l = 4;
						while (l < maxlen) {
							const uint32_t x = ld32(s_data + q + l) ^ ld32(s_data + p + l);
							if (x) { l += (uint32_t)(__builtin_ctz(x) >> 3); break; }
							l += 4;
						}
					}
					if (l > maxlen) { l = maxlen; }
					if (l > best) { best = l; boff = p - q; if (best == maxlen) { break; } }
				}
				q = qn;
			}
		}
		// greedy walk (wave-uniform, scalar unit)
		const u64 mm = __ballot(best >= 3u);
		u64 tokmask = 0, matchmask = 0;
		uint32_t cur = entry;
		while (cur < wend) {
			const uint32_t rel = cur - wbase;
			const u64 rest = mm >> rel;
			if (rest == 0) { tokmask |= (~(u64)0) << rel; cur = wend; break; }
			const uint32_t j = ctz64(rest);
			const uint32_t mpos = rel + j;
			tokmask |= ((((u64)2) << j) - ((u64)1)) << rel;              // j literals + the match start
			matchmask |= ((u64)1) << mpos;
			cur = wbase + mpos + (uint32_t)__builtin_amdgcn_readlane((int)best, (int)mpos);
		}
		if (wend - wbase < 64u) { tokmask &= (((u64)1) << (wend - wbase)) - ((u64)1); }
		entry = cur;

		// emit: pos(t) = 2 (header) + (t div 8 + 1) + sum size(u<t)
		const bool is_tok = (tokmask >> lane) & ((u64)1);
		const bool is_m = (matchmask >> lane) & ((u64)1);
		const uint32_t tb = popc_below(tokmask), mb = popc_below(matchmask);
		const uint32_t t = T + tb;
		const uint32_t pos = 3u + (t >> 3) + S + tb + mb;
		if (is_tok) {
			if ((t & 7u) == 0) { s_out[pos - 1u] = 0; s_grpflag[(t >> 3) & 15u] = pos - 1u; }
			if (is_m) {
				const uint32_t tok = ((boff - 1u) << shift) | (best - 3u);
				s_out[pos] = (uint8_t)tok; s_out[pos + 1u] = (uint8_t)(tok >> 8);
			} else { s_out[pos] = (uint8_t)own4; }
		}
		__syncthreads();
		if (is_m) {
			const uint32_t fp = s_grpflag[(t >> 3) & 15u];
			atomicOr(reinterpret_cast<uint32_t*>(s_out + (fp & ~3u)), (1u << (t & 7u)) << ((fp & 3u) * 8u));
		}
		__syncthreads();
		const uint32_t nt = (uint32_t)__popcll(tokmask), nm = (uint32_t)__popcll(matchmask);
		T += nt; S += nt + nm;
		if (((T + 7u) >> 3) + S >= n) { raw = true; break; }    // running size reached n (:85-86) => store raw
	}

	// ---- chunk image: header + payload into the scratch slot ----------------------------------------------
	const uint32_t csize = ((T + 7u) >> 3) + S;
	uint32_t total;
	if (!raw && csize < n) {
		if (lane == 0) { st16(s_out, 0xB000u | (csize - 1u)); }
		total = 2u + csize;
	} else {
		__syncthreads();
		const uint32_t hdr = 0x3000u | (n - 1u);
		for (uint32_t k = lane; k < (n + 2u + 3u) / 4u; k += 64u) {
			const uint32_t v = (k == 0) ? (hdr | (ld16(s_data) << 16)) : ld32(s_data + 4u * k - 2u);
			reinterpret_cast<uint32_t*>(s_out)[k] = v;
		}
		total = 2u + n;
	}
	__syncthreads();
	uint8_t* __restrict__ slot = slots + (u64)c * LZNT1_SLOT;
	for (uint32_t i = lane * 16u; i < total; i += 1024u) {
		*reinterpret_cast<uint4*>(slot + i) = *reinterpret_cast<const uint4*>(s_out + i);
	}
	if (lane == 0) { slot_size[c] = total; }
}

void launch_lznt1_chunks(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, uint8_t* slots, uint32_t* slot_size)
{
	if (bt.n_chunks == 0) { return; }
	hipLaunchKernelGGL(lznt1_chunk_kernel, dim3(bt.n_chunks), dim3(64), 0, st, d_in, bt, slots, slot_size);
}

} // namespace msc

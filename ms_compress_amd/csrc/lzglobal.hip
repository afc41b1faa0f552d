// lzglobal.hip -- tokens -> bytes for LARGE units, by all CUs (the last stage of Xpress / Xpress+Huffman decompression).
//
// The copy loop of the reference decoders (/root/reference/src/xpress_decompress.cpp:442-452, xpress_huff_decompress.cpp:120-127) produces a
// byte from a literal or from an earlier byte of the output: a chain through the output. lz_copy_block_kernel (decompress.hip) walks it 8 KiB
// at a time with ONE block per unit: 30 000 cycles per tile, 78 ms for the 51 MB of mozilla, while 255 CUs look on. Here the unit is cut into
// tiles of 8192 output bytes that are worked on by different blocks at the same time:
//   lzg_sums_kernel / lzg_scan_kernel / lzg_dir_kernel   where every token starts (prefix sum of the token lengths over the unit, 8192 tokens
//       per block, two levels) -> for every tile: the first token that starts in it or behind it, and where (the "directory").
//   lzg_expand_kernel   one block per tile, independent of all other tiles: the tile's tokens are placed, every byte finds its token (as in
//       lz_copy_block_kernel) and becomes a VALUE (a literal, or a copy whose source chain ends at a literal inside the tile: pointer jumping in
//       LDS) or a POINTER to a byte of an earlier tile; values go to the output at once, both go to a 32-bit word per output byte in HBM.
//   lzg_jump_kernel     passes over those words: a pointer reads the word it points to, up to 8 hops; a value ends the chain (the byte is
//       written), another pointer replaces it (so chains halve or better from pass to pass). Sources always lie before their byte, so every
//       chain ends; 32 passes cover chains of 2^32, a pass that finds the one before it left nothing open returns at once. A word is
//       written with one 32-bit store, so a racing reader sees the old pointer or the new word: both lead to the same byte.
// Scratch: 4 bytes per output byte of the units taken (capacity >= LZG_MIN_CAP; the plan gives up the path when that exceeds its budget).
#include "common.h"
#include "kernels.h"

namespace msc {

#define LZG_T   (1u << LZG_TILE_SHIFT)                             // output bytes per tile
#define LZG_TB  8192u                                              // tokens per token block
#define LZG_NT  1024u
#define LZG_VAL 0xFFFFFF00u                                        // word >= this: the byte is word & 0xFF; below: the index of the source byte in the unit

__device__ __forceinline__ uint32_t lzg_seg(const u64* __restrict__ prefix, uint32_t n_seg, u64 c)
{
	uint32_t lo = 0, hi = n_seg;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (prefix[mid] <= c) { lo = mid; } else { hi = mid; } }
	return lo;
}
__device__ __forceinline__ uint32_t lzg_len(uint32_t w) { return (w & 0x80000000u) ? 1u : (w >> 16) & 0x7FFFu; }

// block sum of `v` (u64) over LZG_NT threads; s_w: 16 u64
__device__ __forceinline__ u64 lzg_block_excl(u64 v, u64* s_w, uint32_t tid, u64* total)
{
	const uint32_t lane = tid & 63u, wv = tid >> 6;
	u64 x = v;
	#pragma unroll
	for (uint32_t d = 1; d < 64u; d <<= 1) { const u64 y = __shfl_up(x, d, 64); if (lane >= d) { x += y; } }
	if (lane == 63u) { s_w[wv] = x; }
	__syncthreads();
	u64 base = 0, tot = 0;
	for (uint32_t k = 0; k < LZG_NT / 64u; ++k) { const u64 s = s_w[k]; if (k < wv) { base += s; } tot += s; }
	__syncthreads();
	*total = tot;
	return base + x - v;
}

// ---- directory ----------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LZG_NT) void lzg_sums_kernel(LzgTables g, const u64* __restrict__ tok_prefix, const uint32_t* __restrict__ tok, const u64* __restrict__ ntok,
                                                         const int32_t* __restrict__ d_status)
{
	__shared__ u64 s_w[16];
	const uint32_t tid = threadIdx.x;
	const uint32_t b = lzg_seg(g.tb_prefix, g.n_big, blockIdx.x), u = g.unit[b];
	const u64 j = blockIdx.x - g.tb_prefix[b];
	if (d_status[u] != 0) { return; }
	const u64 nt = ntok[u];
	if (j * LZG_TB >= nt) { return; }
	const uint32_t* __restrict__ mytok = tok + tok_prefix[u];
	u64 sum = 0;
	#pragma unroll
	for (uint32_t r = 0; r < LZG_TB / LZG_NT; ++r) { const u64 ti = j * LZG_TB + (u64)tid * (LZG_TB / LZG_NT) + r; if (ti < nt) { sum += lzg_len(mytok[ti]); } }
	u64 tot;
	(void)lzg_block_excl(sum, s_w, tid, &tot);
	if (tid == 0) { g.bsum[blockIdx.x] = tot; }
}

__global__ __launch_bounds__(LZG_NT) void lzg_scan_kernel(LzgTables g, const u64* __restrict__ ntok, const int32_t* __restrict__ d_status)
{
	__shared__ u64 s_w[16];
	const uint32_t tid = threadIdx.x, b = blockIdx.x, u = g.unit[b];
	if (d_status[u] != 0) { return; }
	const u64 nb = (ntok[u] + LZG_TB - 1u) / LZG_TB;
	u64* __restrict__ my = g.bsum + g.tb_prefix[b];
	u64 carry = 0;
	for (u64 i0 = 0; i0 < nb; i0 += LZG_NT) {
		const u64 i = i0 + tid;
		const u64 v = i < nb ? my[i] : 0;
		u64 tot;
		const u64 ex = lzg_block_excl(v, s_w, tid, &tot);
		if (i < nb) { my[i] = carry + ex; }
		carry += tot;
	}
}

__global__ __launch_bounds__(LZG_NT) void lzg_dir_kernel(LzgTables g, const u64* __restrict__ tok_prefix, const uint32_t* __restrict__ tok, const u64* __restrict__ ntok,
                                                        const u64* __restrict__ d_out_len, const int32_t* __restrict__ d_status)
{
	__shared__ u64 s_w[16];
	const uint32_t tid = threadIdx.x;
	const uint32_t b = lzg_seg(g.tb_prefix, g.n_big, blockIdx.x), u = g.unit[b];
	const u64 j = blockIdx.x - g.tb_prefix[b];
	if (d_status[u] != 0) { return; }
	const u64 nt = ntok[u], total = d_out_len[u];
	uint32_t* __restrict__ dt = g.dir_tok + g.tile_prefix[b];
	uint32_t* __restrict__ dp = g.dir_pos + g.tile_prefix[b];
	if (j == 0 && tid == 0) { dt[0] = 0; dp[0] = 0; }
	if (j * LZG_TB >= nt) { return; }
	const uint32_t* __restrict__ mytok = tok + tok_prefix[u];
	constexpr uint32_t TPT = LZG_TB / LZG_NT;
	uint32_t len[TPT]; u64 sum = 0;
	#pragma unroll
	for (uint32_t r = 0; r < TPT; ++r) { const u64 ti = j * LZG_TB + (u64)tid * TPT + r; len[r] = ti < nt ? lzg_len(mytok[ti]) : 0u; sum += len[r]; }
	u64 tot;
	u64 p = g.bsum[blockIdx.x] + lzg_block_excl(sum, s_w, tid, &tot);
	#pragma unroll
	for (uint32_t r = 0; r < TPT; ++r) {
		const u64 e = p + len[r];                                       // where the next token starts
		for (u64 k = (p >> LZG_TILE_SHIFT) + 1u; k <= (e >> LZG_TILE_SHIFT) && k * LZG_T < total; ++k) {   // tile boundaries in (p, e]: the next token is the first that starts at or behind them
			dt[k] = (uint32_t)(j * LZG_TB + (u64)tid * TPT + r + 1u); dp[k] = (uint32_t)e;
		}
		p = e;
	}
}

// ---- one tile ------------------------------------------------------------------------------------------------------------------------------
struct LzgLds {
	uint32_t info[LZG_T];                                          // token word at its start position; later one word per byte: value, source in the tile, or source before it
	uint32_t bm[LZG_T / 32u];
	uint32_t last[LZG_T / 32u];
	uint32_t wsum[16];
};
#define LZG_W_VAL 0x80000000u                                      // | byte
#define LZG_W_EXT 0x40000000u                                      // | how far before the tile start (1 .. 65535)

__global__ __launch_bounds__(LZG_NT) void lzg_expand_kernel(LzgTables g, BatchTables bt, const u64* __restrict__ tok_prefix, const uint32_t* __restrict__ tok,
                                                           const u64* __restrict__ ntok, const u64* __restrict__ d_out_len, const int32_t* __restrict__ d_status,
                                                           uint8_t* __restrict__ d_out)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lzg_smem[];
	LzgLds& L = *reinterpret_cast<LzgLds*>(lzg_smem);
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	const uint32_t b = lzg_seg(g.tile_prefix, g.n_big, blockIdx.x), u = g.unit[b];
	const u64 k = blockIdx.x - g.tile_prefix[b];
	if (d_status[u] != 0) { return; }
	const u64 total = d_out_len[u], nt = ntok[u], w0 = k * LZG_T;
	if (w0 >= total) { return; }
	const uint32_t wlen = total - w0 < LZG_T ? (uint32_t)(total - w0) : LZG_T;
	const uint32_t* __restrict__ mytok = tok + tok_prefix[u];
	uint8_t* __restrict__ dst = d_out + bt.out_off[u];
	uint32_t* __restrict__ P = g.words + g.word_prefix[b];
	const u64 t = g.dir_tok[g.tile_prefix[b] + k];
	const u64 tpos = g.dir_pos[g.tile_prefix[b] + k];                    // >= w0 (units are below 4 GiB)
	const uint32_t run_w = (tpos > w0 && t > 0) ? mytok[t - 1u] : LZG_W_VAL;   // the token running at the tile start
	constexpr uint32_t TPT = LZG_T / LZG_NT;
	if (tid < LZG_T / 32u) { L.bm[tid] = 0; }
	__syncthreads();
	// ---- the tokens that start in this tile (at most LZG_T of them): thread j looks at tokens t + 8 j .. t + 8 j + 7 ----
	if (t < nt && tpos < w0 + wlen) {
		uint32_t w[TPT], len[TPT], sum = 0;
		#pragma unroll
		for (uint32_t r = 0; r < TPT; ++r) {
			const u64 ti = t + (u64)tid * TPT + r;
			w[r] = ti < nt ? mytok[ti] : LZG_W_VAL;
			len[r] = ti < nt ? lzg_len(w[r]) : 0u;
			sum += len[r];
		}
		const uint32_t incl = wave_incl_scan_add_u32(sum);
		if (lane == 63u) { L.wsum[wv] = incl; }
		__syncthreads();
		u64 base = 0;
		for (uint32_t q = 0; q < wv; ++q) { base += L.wsum[q]; }
		u64 p = tpos + base + incl - sum;
		uint32_t accw = 0xFFFFFFFFu, accb = 0;
		#pragma unroll
		for (uint32_t r = 0; r < TPT; ++r) {
			if (t + (u64)tid * TPT + r < nt && p < w0 + wlen) {
				const uint32_t q = (uint32_t)(p - w0);
				L.info[q] = w[r];
				if ((q >> 5) != accw) { if (accb) { atomicOr(&L.bm[accw], accb); } accw = q >> 5; accb = 0; }
				accb |= 1u << (q & 31u);
			}
			p += len[r];
		}
		if (accb) { atomicOr(&L.bm[accw], accb); }
	}
	__syncthreads();
	// ---- per 32-position word: the highest token start before it (block max-scan over 256 words), biased by 1 ----
	if (tid < LZG_T / 32u) {
		const uint32_t wd = L.bm[tid];
		const uint32_t hi = wd ? tid * 32u + 31u - (uint32_t)__builtin_clz(wd) + 1u : 0u;
		const uint32_t incl = wave_incl_scan_max(hi);
		if (lane == 63u) { L.wsum[wv] = incl; }
		L.last[tid] = incl;
	}
	__syncthreads();
	if (tid < LZG_T / 32u) {
		uint32_t before = 0;
		for (uint32_t q = 0; q < wv; ++q) { before = before > L.wsum[q] ? before : L.wsum[q]; }
		const uint32_t incl = L.last[tid] > before ? L.last[tid] : before;
		const uint32_t prev_lane = (uint32_t)__shfl_up((int)incl, 1, 64);
		L.last[tid] = lane ? prev_lane : before;                         // exclusive: the highest start in the words before this one (0 = none in this tile)
	}
	__syncthreads();
	// ---- bytes: value, source in the tile, or source before the tile ----
	uint32_t myw[TPT];
	#pragma unroll
	for (uint32_t r = 0; r < TPT; ++r) {
		const uint32_t q = r * LZG_NT + tid;
		uint32_t word = LZG_W_VAL;
		if (q < wlen) {
			const uint32_t bits = L.bm[q >> 5] & (0xFFFFFFFFu >> (31u - (q & 31u)));
			const uint32_t sb = bits ? (q & ~31u) + 32u - (uint32_t)__builtin_clz(bits) : L.last[q >> 5];
			const uint32_t inf = sb ? L.info[sb - 1u] : run_w;
			if (inf & 0x80000000u) { word = LZG_W_VAL | (inf & 0xFFu); }
			else {
				const int32_t rel = (int32_t)q - (int32_t)(inf & 0xFFFFu);        // one offset back: the same byte
				word = rel >= 0 ? (uint32_t)rel : (LZG_W_EXT | (uint32_t)(-rel));
			}
		}
		myw[r] = word;
	}
	__syncthreads();
	#pragma unroll
	for (uint32_t r = 0; r < TPT; ++r) { L.info[r * LZG_NT + tid] = myw[r]; }
	__syncthreads();
	for (;;) {
		bool open = false;
		#pragma unroll
		for (uint32_t r = 0; r < TPT; ++r) {
			const uint32_t q = r * LZG_NT + tid;
			uint32_t mine = myw[r];
			if (!(mine & (LZG_W_VAL | LZG_W_EXT))) {
				mine = __hip_atomic_load(&L.info[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // its value, its source before the tile, or where IT looks
				if (!(mine & (LZG_W_VAL | LZG_W_EXT))) { mine = __hip_atomic_load(&L.info[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // (two hops per round: half the barriers)
				myw[r] = mine;
				__hip_atomic_store(&L.info[q], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				open |= !(mine & (LZG_W_VAL | LZG_W_EXT));
			}
		}
		if (!__syncthreads_or(open ? 1 : 0)) { break; }
	}
	bool ext = false;
	#pragma unroll
	for (uint32_t r = 0; r < TPT; ++r) {
		const uint32_t q = r * LZG_NT + tid;
		if (q < wlen) {
			const uint32_t mine = myw[r];
			if (mine & LZG_W_VAL) { dst[w0 + q] = (uint8_t)mine; P[w0 + q] = LZG_VAL | (mine & 0xFFu); }
			else { P[w0 + q] = (uint32_t)(w0 - (mine & 0xFFFFu)); ext = true; }      // (the parsers have checked that no match reaches in front of the unit)
		}
	}
	const int any = __syncthreads_or(ext ? 1 : 0);
	if (tid == 0) { g.tile_pass[blockIdx.x] = any ? 0u : 0xFFu; }       // the pass that has to look at this tile next (0xFF: none)
}

// ---- pointer passes ------------------------------------------------------------------------------------------------------------------------
#define LZG_HOPS 8u
__global__ __launch_bounds__(256) void lzg_jump_kernel(LzgTables g, BatchTables bt, const u64* __restrict__ d_out_len, const int32_t* __restrict__ d_status,
                                                      uint8_t* __restrict__ d_out, uint32_t pass, uint32_t hops)
{
	if (pass > 0 && g.open[pass - 1u] == 0) { return; }                  // the pass before left nothing open
	uint32_t still_all = 0;
	for (uint32_t tile = blockIdx.x; tile < g.n_tiles; tile += gridDim.x) {
		if (g.tile_pass[tile] != pass) { continue; }                     // all its words are values already (or it lies behind the unit's end)
		const uint32_t b = lzg_seg(g.tile_prefix, g.n_big, tile), u = g.unit[b];
		const u64 total = d_out_len[u], w0 = (u64)(tile - g.tile_prefix[b]) * LZG_T;
		if (d_status[u] != 0 || w0 >= total) { continue; }                 // (a tile the expansion did not visit: its flag is stale)
		uint32_t* __restrict__ P = g.words + g.word_prefix[b];
		uint8_t* __restrict__ dst = d_out + bt.out_off[u];
		const u64 end = total - w0 < LZG_T ? total : w0 + LZG_T;
		uint32_t still = 0;
		for (u64 i = w0 + threadIdx.x; i < end; i += 256u) {
			uint32_t v = __hip_atomic_load(&P[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (v >= LZG_VAL) { continue; }
			bool done = false;
			#pragma unroll 1
			for (uint32_t h = 0; h < hops; ++h) {
				const uint32_t w = __hip_atomic_load(&P[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				v = w;
				if (w >= LZG_VAL) { done = true; break; }
			}
			__hip_atomic_store(&P[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (done) { dst[i] = (uint8_t)v; } else { ++still; }
		}
		const int any = __syncthreads_or(still ? 1 : 0);
		if (any && threadIdx.x == 0) { g.tile_pass[tile] = (uint8_t)(pass + 1u); }
		still_all += still;
	}
	if (__ballot(still_all != 0)) {
		uint32_t s = still_all;
		#pragma unroll
		for (uint32_t d = 32; d > 0; d >>= 1) { s += (uint32_t)__shfl_down((int)s, d, 64); }
		if ((threadIdx.x & 63u) == 0 && s) { atomicAdd(&g.open[pass], s); }
	}
}

void launch_lz_copy_global(hipStream_t st, const LzgTables& g, const BatchTables& bt, const u64* tok_prefix, const uint32_t* tok, const u64* ntok,
                           const u64* d_out_len, const int32_t* d_status, uint8_t* d_out, int phase)
{
	if (g.n_big == 0) { return; }
	switch (phase) {
	case 0:
		(void)hipMemsetAsync(g.open, 0, LZG_PASSES * sizeof(uint32_t), st);
		hipLaunchKernelGGL(lzg_sums_kernel, dim3(g.n_tb), dim3(LZG_NT), 0, st, g, tok_prefix, tok, ntok, d_status);
		hipLaunchKernelGGL(lzg_scan_kernel, dim3(g.n_big), dim3(LZG_NT), 0, st, g, ntok, d_status);
		hipLaunchKernelGGL(lzg_dir_kernel, dim3(g.n_tb), dim3(LZG_NT), 0, st, g, tok_prefix, tok, ntok, d_out_len, d_status);
		break;
	case 1:
		{ static PerDeviceOnce attr; if (attr.needed()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lzg_expand_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LzgLds)); attr.done(); } }
		hipLaunchKernelGGL(lzg_expand_kernel, dim3(g.n_tiles), dim3(LZG_NT), sizeof(LzgLds), st, g, bt, tok_prefix, tok, ntok, d_out_len, d_status, d_out);
		break;
	default: {
		static const uint32_t hops = [] { const char* e = getenv("MSCOMP_AMD_LZG_HOPS"); const long v = e ? atol(e) : 0; return (uint32_t)(v > 0 && v < 1000 ? v : LZG_HOPS); }();
		for (uint32_t pass = 0; pass < LZG_PASSES; ++pass) { hipLaunchKernelGGL(lzg_jump_kernel, dim3(g.n_tiles < 4096u ? g.n_tiles : 4096u), dim3(256), 0, st, g, bt, d_out_len, d_status, d_out, pass, hops); }
		break;
	}
	}
}

} // namespace msc

// decompress.hip -- gfx950 decompressors, batch form: n independent units resident in HBM.
//
// LZNT1 follows the one-shot semantics of the reference, which is its streaming inflate driven once over the whole
// buffer (/root/reference/src/lznt1_decompress.cpp:122-290 through ALL_AT_ONCE_WRAPPER_DECOMPRESS,
// /root/reference/include/mscomp/internal.h:616-630):
//   * chunk headers are walked while output room AND >= 2 input bytes remain (:252-259); header 0 ends the stream and is
//     DATA_ERROR unless it is the last two bytes (:128-134); a chunk longer than the remaining input is kept as partial
//     state, so the one-shot call ends in BUF_ERROR (:136-143, wrapper :627); the signature is checked after that (:151);
//   * every compressed chunk is decoded against a 4096-byte limit (:158,:179), any chunk error becomes DATA_ERROR;
//   * a chunk may decode to fewer than 4096 bytes anywhere in the stream - the next chunk continues right behind it;
//   * output that does not fit (or input left when the output is full) gives BUF_ERROR; a single trailing byte is accepted
//     when it is 0 (:223-227).
// The chunks of a unit are independent once their headers are known, so the work is: (1) one block per unit walks the
// header chain through an LDS window, (2) one wave per chunk decodes in LDS and writes the chunk where it lands if every
// earlier chunk holds 4096 bytes, (3) one wave per unit adds the sizes up, decides the status and notices units with short
// chunks in the middle, (4) whose chunks are decoded again to their exact places.
#include "kernels.h"

namespace msc {

// ===================================================================================================================
// (1) header chain
// ===================================================================================================================
#define LZD_WIN 65536u                  // bytes per LDS window (two windows: the next one loads while the chain walks this one)
#define LZD_SCAN_THREADS 1024u
#define LZD_SCAN_LDS (2u * (LZD_WIN + 16u))
enum { LZD_EOI0 = 0, LZD_EOI1 = 1, LZD_ZERO_OK = 2, LZD_ZERO_BAD = 3, LZD_TRUNC = 4, LZD_BADSIG = 5 };

// cin[chunk_prefix[u] + j] = offset of the header of chunk j in unit u; cnt[u] chunks; stop[2u] = what ended the walk | last byte << 8, stop[2u+1] = where
__global__ __launch_bounds__(LZD_SCAN_THREADS) void lzd_scan_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, uint32_t* __restrict__ cin,
                                                                 uint32_t* __restrict__ cnt, uint32_t* __restrict__ stop)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t s_win[];      // [2][LZD_WIN + 16]
	__shared__ uint32_t s_done;
	const uint32_t tid = threadIdx.x, u = blockIdx.x;
	const u64 n = bt.in_len[u];
	const uint8_t* base = d_in + bt.in_off[u];
	const uint32_t a0 = (uint32_t)((uintptr_t)base & 15u);
	const uint8_t* ab = base - a0;                                       // 16-byte aligned; stream coordinate q = unit offset + a0
	const u64 end = n + a0;
	uint32_t* __restrict__ my_cin = cin + bt.chunk_prefix[u];
	if (tid == 0) { s_done = 0; }
	uint4 r[4]; uint4 rt = make_uint4(0, 0, 0, 0);
	// window k: q in [k*W, (k+1)*W + 16)
	#define LZD_LOAD(k) { const u64 w0_ = (u64)(k) * LZD_WIN; \
		_Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { const u64 q_ = w0_ + ((u64)i_ * LZD_SCAN_THREADS + tid) * 16u; \
			r[i_] = q_ < end ? *reinterpret_cast<const uint4*>(ab + q_) : make_uint4(0, 0, 0, 0); } \
		if (tid == 0) { const u64 q_ = w0_ + LZD_WIN; rt = q_ < end ? *reinterpret_cast<const uint4*>(ab + q_) : make_uint4(0, 0, 0, 0); } }
	#define LZD_STORE(k) { uint8_t* b_ = s_win + ((k) & 1u) * (LZD_WIN + 16u); \
		_Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { *reinterpret_cast<uint4*>(b_ + ((uint32_t)i_ * LZD_SCAN_THREADS + tid) * 16u) = r[i_]; } \
		if (tid == 0) { *reinterpret_cast<uint4*>(b_ + LZD_WIN) = rt; } }
	LZD_LOAD(0) LZD_STORE(0)
	__syncthreads();
	u64 pos = a0; uint32_t count = 0;                                    // lane 0 of wave 0
	for (u64 k = 0; ; ++k) {
		const bool more = (k + 1) * LZD_WIN < end;
		if (more) { LZD_LOAD(k + 1) }
		if (tid == 0) {
			const uint8_t* b = s_win + (k & 1u) * (LZD_WIN + 16u);
			const u64 wend = (k + 1) * LZD_WIN;
			uint32_t done = 0, kind = 0;
			while (pos < wend) {
				if (pos + 2 > end) { kind = (pos < end) ? (LZD_EOI1 | ((uint32_t)b[pos - k * LZD_WIN] << 8)) : LZD_EOI0; done = 1; break; }
				const uint32_t o = (uint32_t)(pos - k * LZD_WIN);
				const uint32_t hdr = (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8);
				if (hdr == 0) { kind = (end - pos == 2) ? LZD_ZERO_OK : LZD_ZERO_BAD; done = 1; break; }   // :128-134
				const uint32_t sz = (hdr & 0xFFFu) + 3u;
				if (sz > end - pos) { kind = LZD_TRUNC; done = 1; break; }                                  // :136-143
				if ((hdr & 0x7000u) != 0x3000u) { kind = LZD_BADSIG; done = 1; break; }                     // :151
				my_cin[count++] = (uint32_t)(pos - a0);
				pos += sz;
			}
			if (!done && !more) { kind = LZD_EOI0; done = 1; }           // the chain left the last window: pos == end
			if (done) { cnt[u] = count; stop[2u * u] = kind; stop[2u * u + 1u] = (uint32_t)(pos - a0); s_done = 1; }
		}
		if (more) { LZD_STORE(k + 1) }
		__syncthreads();
		if (s_done) { break; }
	}
	#undef LZD_LOAD
	#undef LZD_STORE
}

// ===================================================================================================================
// (2)/(4) chunk decode: one wave per chunk
// ===================================================================================================================
#define LZD_ERR 0x8000u
struct LzdLds {
	__attribute__((aligned(16))) uint8_t in[4128];     // the chunk (header + data) at its 16-byte phase in global memory
	__attribute__((aligned(16))) uint8_t out[4096 + 64];
	uint16_t info[4096];                               // at token starts: literal byte, or 0x8000 | (offset - 1)
	u64      bm[64];                                   // token-start bitmap over the output positions
	uint16_t gs[464];                                  // start (data offset) of every flag group
};

// size (<= 4096) or LZD_ERR. The decoded bytes are left in L.out.
__device__ __forceinline__ uint32_t lzd_decode_chunk(LzdLds& L, const uint8_t* __restrict__ src, uint32_t in_size, uint32_t lane)
{
	// ---- load: 16-byte words that hold at least one byte of the chunk ----
	const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
	const uint8_t* ab = src - a0;
	const uint32_t nw = (a0 + in_size + 15u) >> 4;                       // <= 258
	for (uint32_t i = lane; i < nw; i += 64u) { *reinterpret_cast<uint4*>(L.in + i * 16u) = *reinterpret_cast<const uint4*>(ab + i * 16u); }
	L.bm[lane] = 0;
	__syncthreads();
	const uint8_t* d = L.in + a0 + 2u;                                  // chunk data
	const uint32_t n = in_size - 2u;                                     // 1..4096
	// ---- flag groups: p -> p + 9 + popcount(flags) ----
	uint32_t G = 0;
	{
		uint32_t p = 0;
		while (p < n) {
			if (lane == 0) { L.gs[G] = (uint16_t)p; }
			++G;
			p += 9u + (uint32_t)__builtin_popcount(d[p]);
		}
	}
	__syncthreads();
	// ---- tokens, 64 at a time: input position, length (the offset/length split depends on the output position), output position ----
	uint32_t base_pos = 0, sh = 12, err = 0;                             // sh: per-lane copy of the (monotone) split
	const uint32_t nb = (G + 7u) >> 3;
	for (uint32_t tb = 0; tb < nb; ++tb) {
		const uint32_t t = tb * 64u + lane, g = t >> 3, k = t & 7u;
		uint32_t ipos = 0, flags = 0;
		if (g < G) { const uint32_t gp = L.gs[g]; flags = d[gp]; ipos = gp + 1u + k + (uint32_t)__builtin_popcount(flags & ((1u << k) - 1u)); }
		const bool valid = g < G && ipos < n;
		const bool is_match = valid && ((flags >> k) & 1u);
		if (is_match && ipos + 2u > n) { err = 1; }                     // :99 (two bytes needed)
		const uint32_t raw = valid ? ((uint32_t)d[ipos] | (is_match ? (uint32_t)d[ipos + 1u] << 8 : 0u)) : 0u;
		uint32_t len, pos;
		for (;;) {
			len = valid ? (is_match ? (raw & ((1u << sh) - 1u)) + 3u : 1u) : 0u;
			pos = base_pos + wave_incl_scan_add_u32(len) - len;
			const u64 m = __ballot(is_match && sh > 4u && pos > (16u << (12u - sh)));       // :100 (the split follows the position)
			if (!m) { break; }
			const uint32_t f = ctz64(m);
			const uint32_t pf = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)f);
			uint32_t ns = (uint32_t)__builtin_amdgcn_readlane((int)sh, (int)f);
			while (ns > 4u && pf > (16u << (12u - ns))) { --ns; }
			if (lane >= f) { sh = ns; }
		}
		const uint32_t off = (raw >> sh) + 1u;
		if (valid && (is_match ? (off > pos || pos + len > 4096u) : pos >= 4096u)) { err = 1; }   // :104-105; a literal beyond the chunk is our DATA_ERROR
		if (__ballot(err)) { return LZD_ERR; }
		if (valid) {
			L.info[pos] = (uint16_t)(is_match ? (0x8000u | (off - 1u)) : raw);
			atomicOr(reinterpret_cast<uint32_t*>(L.bm) + (pos >> 5), 1u << (pos & 31u));
		}
		base_pos = (uint32_t)__builtin_amdgcn_readlane((int)(pos + len), 63);
		sh = (uint32_t)__builtin_amdgcn_readlane((int)sh, 63);
	}
	const uint32_t total = base_pos;
	__syncthreads();
	// ---- bytes, 64 at a time: every byte finds its token and its source; sources inside the row are chased with bpermute ----
	uint32_t carry = 0;
	for (uint32_t rowbase = 0; rowbase < total; rowbase += 64u) {
		const u64 word = L.bm[rowbase >> 6];
		const uint32_t i = rowbase + lane;
		const u64 mine = word & ((2ull << lane) - 1ull);
		const uint32_t s = mine ? rowbase + 63u - (uint32_t)__builtin_clzll(mine) : carry;
		if (word) { carry = rowbase + 63u - (uint32_t)__builtin_clzll(word); }
		const uint32_t inf = L.info[s];
		const bool lit = !(inf & 0x8000u);
		uint32_t ptr = i, val = inf & 0xFFu;
		bool resolved = lit || i >= total;
		if (!resolved) {
			const uint32_t off = (inf & 0xFFFu) + 1u, dd = i - s;
			uint32_t rem = dd;
			if (dd >= off) {
				const uint32_t q = (uint32_t)((float)dd * __builtin_amdgcn_rcpf((float)off));
				int32_t rr = (int32_t)dd - (int32_t)(q * off);
				if (rr < 0) { rr += (int32_t)off; } else if (rr >= (int32_t)off) { rr -= (int32_t)off; }
				rem = (uint32_t)rr;
			}
			ptr = s - off + rem;
			if (ptr < rowbase) { val = L.out[ptr]; resolved = true; }
		}
		while (__ballot(!resolved)) {
			const uint32_t tl = resolved ? lane : ptr - rowbase;
			const uint32_t tv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tl << 2), (int)(val | (resolved ? 0x100u : 0u)));
			const uint32_t tp = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tl << 2), (int)ptr);
			if (!resolved) { if (tv & 0x100u) { val = tv & 0xFFu; resolved = true; } else { ptr = tp; if (ptr < rowbase) { val = L.out[ptr]; resolved = true; } } }
		}
		L.out[i] = (uint8_t)val;
		__syncthreads();
	}
	return total;
}

// LDS bytes [0, n) -> global dst (any alignment), one wave
__device__ __forceinline__ void lzd_store(uint8_t* __restrict__ dst, const uint8_t* lds, uint32_t n, uint32_t lane)
{
	uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u);
	if (head > n) { head = n; }
	if (lane < head) { dst[lane] = lds[lane]; }
	const uint32_t body = (n - head) >> 2;
	uint32_t* __restrict__ d32 = reinterpret_cast<uint32_t*>(dst + head);
	for (uint32_t i = lane; i < body; i += 64u) { d32[i] = lds_ld32(lds, head + i * 4u); }
	for (uint32_t i = head + body * 4u + lane; i < n; i += 64u) { dst[i] = lds[i]; }
}

// largest u with prefix[u] <= c (prefix: u64 exclusive scan of the chunk counts; empty units share an entry with their successor)
__device__ __forceinline__ uint32_t unit_of_flat(const u64* __restrict__ prefix, uint32_t n_units, u64 c)
{
	uint32_t lo = 0, hi = n_units;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (prefix[mid] <= c) { lo = mid; } else { hi = mid; } }
	return lo;
}

// EXACT = false: every chunk, written where it lands if all earlier chunks hold 4096 bytes; records the size.
// EXACT = true: only units flagged irregular; chunks whose place differs are decoded again to the exact place.
template <bool EXACT>
__global__ __launch_bounds__(64) void lzd_chunk_kernel(const uint8_t* __restrict__ d_in, BatchTables bt, const uint32_t* __restrict__ cin,
                                                      const u64* __restrict__ start, uint16_t* __restrict__ csize,
                                                      const uint32_t* __restrict__ irregular, uint8_t* __restrict__ d_out)
{
	__shared__ LzdLds L;
	const uint32_t lane = threadIdx.x;
	const u64 total_chunks = start[bt.n_units];
	if (EXACT && irregular[bt.n_units] == 0) { return; }
	for (u64 c = blockIdx.x; c < total_chunks; c += gridDim.x) {
		const uint32_t u = unit_of_flat(start, bt.n_units, c);
		if (EXACT && !irregular[u]) { continue; }
		const uint32_t j = (uint32_t)(c - start[u]);
		const uint32_t slot = bt.chunk_prefix[u] + j;
		u64 pos = (u64)j * 4096u;
		if (EXACT) {
			u64 acc = 0;                                                 // sum of the sizes of the chunks before j
			for (uint32_t i = lane; i < j; i += 64u) { acc += csize[bt.chunk_prefix[u] + i] & 0x1FFFu; }
			for (int o = 32; o; o >>= 1) { acc += __shfl_xor(acc, o, 64); }
			if (acc == pos) { continue; }
			pos = acc;
		}
		const uint8_t* src = d_in + bt.in_off[u] + cin[slot];
		const uint32_t hdr = (uint32_t)src[0] | ((uint32_t)src[1] << 8);
		const uint32_t in_size = (hdr & 0xFFFu) + 3u;
		uint32_t size;
		if (hdr & 0x8000u) {
			size = lzd_decode_chunk(L, src, in_size, lane);
		} else {                                                         // stored chunk (:192-209)
			size = in_size - 2u;
			for (uint32_t i = lane; i < size; i += 64u) { L.out[i] = src[2u + i]; }
			__syncthreads();
		}
		if (!EXACT && lane == 0) { csize[slot] = (uint16_t)size; }
		const u64 cap = bt.out_cap[u];
		if (size != LZD_ERR && pos < cap) {
			const u64 room = cap - pos;
			lzd_store(d_out + bt.out_off[u] + pos, L.out, room < size ? (uint32_t)room : size, lane);
		}
		__syncthreads();
	}
}

// ===================================================================================================================
// (3) per unit: positions, status, out_len
// ===================================================================================================================
__global__ __launch_bounds__(64) void lzd_finalize_kernel(BatchTables bt, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ stop,
                                                         const uint16_t* __restrict__ csize, uint32_t* __restrict__ irregular,
                                                         u64* __restrict__ d_out_len, int32_t* __restrict__ d_status)
{
	const uint32_t lane = threadIdx.x, u = blockIdx.x;
	const uint32_t count = cnt[u], kind = stop[2u * u] & 0xFFu, last = stop[2u * u] >> 8;
	const u64 cap = bt.out_cap[u], n = bt.in_len[u];
	const uint16_t* __restrict__ sz = csize + bt.chunk_prefix[u];
	u64 pos = 0; int32_t status = 1; bool irr = false;                   // status 1 = undecided
	for (uint32_t j0 = 0; j0 < count && status == 1; j0 += 64u) {
		const uint32_t j = j0 + lane;
		const uint32_t raw = j < count ? sz[j] : 0u;
		const bool bad = raw == LZD_ERR;
		const uint32_t size = bad ? 0u : raw;
		uint32_t incl = size;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if ((int)lane >= o) { incl += t; } }
		const u64 p = pos + incl - size;
		// the walk of :252-259 at chunk j: no room left -> the caller sees BUF_ERROR; chunk error -> DATA_ERROR; chunk does not fit -> BUF_ERROR
		const uint32_t ev = j < count ? (p >= cap ? 5u : bad ? 3u : p + size > cap ? 5u : 0u) : 0u;
		const u64 m = __ballot(ev != 0);
		irr |= __ballot(j + 1u < count && raw != 4096u) != 0;
		if (m) { status = -(int32_t)__builtin_amdgcn_readlane((int)ev, (int)ctz64(m)); break; }
		pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
	}
	if (status == 1) {
		const bool room = pos < cap;
		switch (kind) {
		case LZD_EOI0:     status = 0; break;
		case LZD_EOI1:     status = last == 0 ? 0 : -5; break;                       // :223-227
		case LZD_ZERO_OK:  status = room ? 0 : -5; break;                            // the header is only read while there is room (:252)
		case LZD_ZERO_BAD: status = room ? -3 : -5; break;
		case LZD_TRUNC:    status = -5; break;
		default:           status = room ? -3 : -5; break;                           // LZD_BADSIG
		}
		(void)n;
	}
	if (lane == 0) {
		d_status[u] = status; d_out_len[u] = status == 0 ? pos : 0;
		irregular[u] = irr ? 1u : 0u;
		if (irr) { atomicOr(&irregular[bt.n_units], 1u); }
	}
}

__global__ void lzd_clear_kernel(uint32_t* p) { *p = 0; }

void launch_lzd_scan(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const LzdBufs& b)
{
	if (bt.n_units == 0) { return; }
	static bool attr_set = false;
	if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lzd_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LZD_SCAN_LDS); attr_set = true; }
	hipLaunchKernelGGL(lzd_clear_kernel, dim3(1), dim3(1), 0, st, b.irregular + bt.n_units);
	hipLaunchKernelGGL(lzd_scan_kernel, dim3(bt.n_units), dim3(LZD_SCAN_THREADS), LZD_SCAN_LDS, st, d_in, bt, b.cin, b.cnt, b.stop);
}
void launch_lzd_chunks(hipStream_t st, const uint8_t* d_in, const BatchTables& bt, const LzdBufs& b, uint8_t* d_out, int exact)
{
	if (bt.n_units == 0) { return; }
	const uint32_t grid = bt.n_chunks < 16384u ? (bt.n_chunks ? bt.n_chunks : 1u) : 16384u;
	if (exact) { hipLaunchKernelGGL(lzd_chunk_kernel<true>, dim3(grid), dim3(64), 0, st, d_in, bt, b.cin, b.start, b.csize, b.irregular, d_out); }
	else       { hipLaunchKernelGGL(lzd_chunk_kernel<false>, dim3(grid), dim3(64), 0, st, d_in, bt, b.cin, b.start, b.csize, b.irregular, d_out); }
}
void launch_lzd_finalize(hipStream_t st, const BatchTables& bt, const LzdBufs& b, u64* d_out_len, int32_t* d_status)
{
	if (bt.n_units == 0) { return; }
	hipLaunchKernelGGL(lzd_finalize_kernel, dim3(bt.n_units), dim3(64), 0, st, bt, b.cnt, b.stop, b.csize, b.irregular, d_out_len, d_status);
}

} // namespace msc

// common.h -- device helpers shared by the gfx950 kernels (wave64 idioms, LDS byte access, batch tables).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msc {

typedef uint64_t u64;

// ---- wave64 helpers (gfx950: wavefront = 64 lanes, hard-coded) -----------------------------------------
__device__ __forceinline__ uint32_t lane_id()
{
	return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of m strictly below this lane
__device__ __forceinline__ uint32_t popc_below(u64 m)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t ctz64(u64 m) { return (uint32_t)__builtin_ctzll(m); }
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64 sgpr64(u64 v)   // tell the compiler a wave-uniform 64-bit value lives in SGPRs
{
	return ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// DPP scans over the 64 lanes: row_shr 1,2,4,8 then row_bcast 15/31, the max/add folded into the DPP instruction itself
// (the builtin form costs v_mov + v_mov_dpp + op per step). A lane whose DPP source is out of range or whose row is
// masked off keeps its value. Two wait states are required between a VALU write and a DPP read of the same VGPR.
#define MSC_DPP_SCAN(op) \
	"s_nop 1\n\t" op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
	"s_nop 1\n\t" op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
	"s_nop 1\n\t" op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
	"s_nop 1\n\t" op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
	"s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
	"s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
	"s_nop 1"
__device__ __forceinline__ uint32_t wave_incl_scan_max(uint32_t v) { asm volatile(MSC_DPP_SCAN("v_max_u32_dpp") : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t wave_incl_scan_add_u32(uint32_t v) { asm volatile(MSC_DPP_SCAN("v_add_u32_dpp") : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_max(v), 63); }

// ---- LDS / global byte-granular access ----------------------------------------------------------------
// gfx950 has unaligned DS access enabled: a misaligned 4-byte LDS read is ONE ds_read_b32.
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ void     st16(uint8_t* p, uint32_t v) { uint16_t w = (uint16_t)v; __builtin_memcpy(p, &w, 2); }
__device__ __forceinline__ void     st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// Index of the first non-zero byte of the 16-byte value x (little endian), 16 when x == 0. Branch-free, 10 VALU:
// v_ffbl_b32 returns 0xFFFFFFFF for a zero dword (the builtin ctz is undefined there, so the instruction is named
// directly); OR-ing 32 / 64 / 96 into a bit index below 32 ADDS the dword's offset and leaves the all-ones "none" as it
// is, and the unsigned minimum picks the first hit. (Round 5: v_or_b32 where rounds 1-4 had v_add_u32 ... clamp. A
// 3-operand / modifier-carrying VOP3 instruction issues at HALF the rate of a VOP1 / VOP2 one on this GPU -- 1 against 2
// wave-instructions per CU and cycle, tools/dev/issue_peak.hip -- and the clamp made the three adds VOP3.)
__device__ __forceinline__ uint32_t ffbl_raw(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) { const uint32_t t = a < b ? a : b; return t < c ? t : c; }
__device__ __forceinline__ uint32_t first_nz_byte16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
	const uint32_t a = ffbl_raw(x1) | 32u, b = ffbl_raw(x2) | 64u, c = ffbl_raw(x3) | 96u;
	return min3u(min3u(ffbl_raw(x0), a, b), c, 128u) >> 3;
}

// 16 bytes at ANY byte offset of an LDS array whose base is 4-byte aligned. A byte-misaligned ds_read_b128 is replayed
// by the LDS pipe (about 64 cycles; SQ_LDS_UNALIGNED_STALL was 77 % of the match finder's LDS cycles), so read 5 ALIGNED
// dwords and funnel-shift (v_alignbyte).
__device__ __forceinline__ uint4 lds_ld128(const uint8_t* base, uint32_t off)
{
	const uint32_t* a = reinterpret_cast<const uint32_t*>(base + (off & ~3u));
	const uint32_t sh = off & 3u;
	const uint32_t w0 = a[0], w1 = a[1], w2 = a[2], w3 = a[3], w4 = a[4];
	return make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
	                  __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh));
}

// 4 bytes at any byte offset of such an array, same reason (two aligned dwords + v_alignbyte)
__device__ __forceinline__ uint32_t lds_ld32(const uint8_t* base, uint32_t off)
{
	const uint32_t* a = reinterpret_cast<const uint32_t*>(base + (off & ~3u));
	return __builtin_amdgcn_alignbyte(a[1], a[0], off & 3u);
}

// ---- per-position match results of the Xpress-family finders --------------------------------------------
// ONE 32-bit word per position: len - 3 in the low half, the offset in the high half (0 = no match). The kernels see the two halves
// as two arrays of uint16_t with a stride of two (`mlen3[i]`, `moff[i]`): a finder that stores a position's match touches ONE 4-byte
// word (the lazy finder's stores are scattered: two separate arrays cost two 32-byte sectors of HBM writes per match, 40 GB per pass of
// BASELINE configs[4]), and the parse kernels find both halves in the same cache line.
struct S16 {
	uint16_t* p;
	__host__ __device__ S16(const uint16_t* q) : p(const_cast<uint16_t*>(q)) {}
	__device__ __forceinline__ uint16_t& operator[](u64 i) const { return p[2u * i]; }
	__device__ __forceinline__ S16 operator+(u64 k) const { return S16(p + 2u * k); }
	__device__ __forceinline__ uint32_t word(u64 i) const { return *reinterpret_cast<const uint32_t*>(p + 2u * i); }   // both halves of position i (called on the LENGTH view: p is word aligned)
};

// ---- batch tables (uploaded once per plan) ------------------------------------------------------------
// unit u owns chunks [chunk_prefix[u], chunk_prefix[u+1]); input = in_off/in_len, output = out_off/out_cap.
struct BatchTables {
	const u64*      in_off;        // n_units
	const u64*      in_len;        // n_units
	const u64*      out_off;       // n_units
	const u64*      out_cap;       // n_units
	const uint32_t* chunk_prefix;  // n_units+1
	uint32_t        n_units;
	uint32_t        n_chunks;
};

// largest u with chunk_prefix[u] <= c   (uniform per block: scalar loads)
__device__ __forceinline__ uint32_t unit_of_chunk(const uint32_t* __restrict__ prefix, uint32_t n_units, uint32_t c)
{
	uint32_t lo = 0, hi = n_units;      // invariant: prefix[lo] <= c < prefix[hi]
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (prefix[mid] <= c) { lo = mid; } else { hi = mid; }
	}
	return lo;
}

// Cooperative byte copy global->global: src is 4-byte aligned (a scratch slot), dst has any alignment.
// Body moves aligned dwords on the destination side, funnel-shifting two source dwords.
__device__ __forceinline__ void copy_from_aligned(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src,
                                                  uint32_t n, uint32_t tid, uint32_t nthr)
{
	uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u);
	if (head > n) { head = n; }
	if (tid < head) { dst[tid] = src[tid]; }
	const uint32_t body = (n - head) >> 2;
	const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(src);
	uint32_t* __restrict__ d32 = reinterpret_cast<uint32_t*>(dst + head);
	// four destination dwords per thread and step (round 5: one dword per step left ~10 dependent load -> shift -> store rounds per 2.4 KiB chunk
	// image in flight one at a time; the destination is only dword-aligned, the 16 bytes go out as the compiler sees fit for that alignment)
	const uint32_t body4 = body & ~3u;
	for (uint32_t i = tid * 4u; i < body4; i += nthr * 4u) {
		const uint32_t a0 = s32[i], a1 = s32[i + 1u], a2 = s32[i + 2u], a3 = s32[i + 3u], a4 = head ? s32[i + 4u] : 0u;   // (index <= body, like the one-dword form's s32[i + 1]: at most 3 bytes behind the image, inside the slot's slack)
		uint32_t v[4];
		if (head == 0) { v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3; }
		else { v[0] = __builtin_amdgcn_alignbyte(a1, a0, head); v[1] = __builtin_amdgcn_alignbyte(a2, a1, head); v[2] = __builtin_amdgcn_alignbyte(a3, a2, head); v[3] = __builtin_amdgcn_alignbyte(a4, a3, head); }
		__builtin_memcpy(__builtin_assume_aligned(d32 + i, 4), v, 16);
	}
	if (head == 0) {
		for (uint32_t i = body4 + tid; i < body; i += nthr) { d32[i] = s32[i]; }
	} else {
		for (uint32_t i = body4 + tid; i < body; i += nthr) { d32[i] = __builtin_amdgcn_alignbyte(s32[i + 1], s32[i], head); }
	}
	for (uint32_t i = head + body * 4u + tid; i < n; i += nthr) { dst[i] = src[i]; }
}

} // namespace msc

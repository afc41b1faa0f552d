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

// ---- LDS / global byte-granular access ----------------------------------------------------------------
// gfx950 has unaligned DS access enabled: a misaligned 4-byte LDS read is ONE ds_read_b32.
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ void     st16(uint8_t* p, uint32_t v) { uint16_t w = (uint16_t)v; __builtin_memcpy(p, &w, 2); }
__device__ __forceinline__ void     st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

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
	if (head == 0) {
		for (uint32_t i = tid; i < body; i += nthr) { d32[i] = s32[i]; }
	} else {
		for (uint32_t i = tid; i < body; i += nthr) { d32[i] = __builtin_amdgcn_alignbyte(s32[i + 1], s32[i], head); }
	}
	for (uint32_t i = head + body * 4u + tid; i < n; i += nthr) { dst[i] = src[i]; }
}

} // namespace msc

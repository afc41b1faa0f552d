// util.hip -- output placement: size scan, slot concatenation, per-unit size/status.
//
// The reference concatenates per-chunk outputs while it walks the buffer (lznt1_compress.cpp:262,
// xpress_huff_compress.cpp:287). On the GPU chunk sizes only exist after the chunk kernels ran, so placement is
// "sizes -> exclusive scan -> copy/encode at the final offset".
#include "common.h"
#include "kernels.h"
#include <atomic>

namespace msc {

#define SCAN_TILE 1024u

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane)
{
	#pragma unroll
	for (uint32_t d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(v, d, 64);
		if (lane >= d) { v += o; }
	}
	return v;
}

// phase 1: per 1024-element tile: local exclusive prefix (u64 store) + tile total
__global__ __launch_bounds__(256) void scan_tiles_kernel(const uint32_t* __restrict__ sizes, u64* __restrict__ prefix,
                                                        uint32_t n, u64* __restrict__ tile_sums)
{
	__shared__ uint32_t s_wave[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	const uint32_t base = blockIdx.x * SCAN_TILE + tid * 4u;
	uint32_t v[4], sum = 0;
	#pragma unroll
	for (int k = 0; k < 4; ++k) { v[k] = (base + k < n) ? sizes[base + k] : 0u; sum += v[k]; }
	const uint32_t incl = wave_incl_scan(sum, lane);
	if (lane == 63) { s_wave[wv] = incl; }
	__syncthreads();
	uint32_t woff = 0;
	for (uint32_t k = 0; k < wv; ++k) { woff += s_wave[k]; }
	uint32_t run = woff + incl - sum;
	#pragma unroll
	for (int k = 0; k < 4; ++k) { if (base + k < n) { prefix[base + k] = run; } run += v[k]; }
	if (tid == 255) { tile_sums[blockIdx.x] = run; }
}
// phase 2: one block turns tile_sums[0..nt) into exclusive offsets in place; tile_sums[nt] = grand total
__global__ __launch_bounds__(256) void scan_tile_sums_kernel(u64* __restrict__ tile_sums, uint32_t nt)
{
	__shared__ u64 s_part[256];
	const uint32_t tid = threadIdx.x;
	const uint32_t per = (nt + 255u) / 256u;
	const uint32_t lo = tid * per, hi = (lo + per < nt) ? lo + per : nt;
	u64 sum = 0;
	for (uint32_t i = lo; i < hi; ++i) { sum += tile_sums[i]; }
	s_part[tid] = sum;
	__syncthreads();
	if (tid == 0) { u64 run = 0; for (uint32_t k = 0; k < 256; ++k) { const u64 t = s_part[k]; s_part[k] = run; run += t; } tile_sums[nt] = run; }
	__syncthreads();
	u64 run = s_part[tid];
	for (uint32_t i = lo; i < hi; ++i) { const u64 t = tile_sums[i]; tile_sums[i] = run; run += t; }
}
// phase 3: add tile offsets; prefix[n] = total
__global__ __launch_bounds__(256) void scan_add_kernel(u64* __restrict__ prefix, uint32_t n, const u64* __restrict__ tile_sums, uint32_t nt)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) { prefix[i] += tile_sums[i / SCAN_TILE]; }
	if (i == 0) { prefix[n] = tile_sums[nt]; }
}

void launch_scan_sizes(hipStream_t st, const uint32_t* sizes, u64* prefix, uint32_t n, u64* tile_sums)
{
	const uint32_t nt = (n + SCAN_TILE - 1u) / SCAN_TILE;
	if (nt) { hipLaunchKernelGGL(scan_tiles_kernel, dim3(nt), dim3(256), 0, st, sizes, prefix, n, tile_sums); }
	hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(1), dim3(256), 0, st, tile_sums, nt);
	hipLaunchKernelGGL(scan_add_kernel, dim3((n + 256u) / 256u), dim3(256), 0, st, prefix, n, tile_sums, nt);
}

// one wave per chunk image
__global__ __launch_bounds__(256) void concat_slots_kernel(const uint8_t* __restrict__ slots, uint32_t slot_stride,
                                                          const uint32_t* __restrict__ slot_size, const u64* __restrict__ prefix,
                                                          BatchTables bt, uint8_t* __restrict__ d_out)
{
	const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6);
	if (c >= bt.n_chunks) { return; }
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t u = unit_of_chunk(bt.chunk_prefix, bt.n_units, c);
	const u64 ustart = prefix[bt.chunk_prefix[u]];
	const u64 utotal = prefix[bt.chunk_prefix[u + 1]] - ustart;
	const u64 cap = bt.out_cap[u];
	if (utotal > cap) { return; }                               // BUF_ERROR unit: output unspecified
	copy_from_aligned(d_out + bt.out_off[u] + (prefix[c] - ustart), slots + (u64)c * slot_stride, slot_size[c], lane, 64u);
}

void launch_concat_slots(hipStream_t st, const uint8_t* slots, uint32_t slot_stride, const uint32_t* slot_size,
                         const u64* prefix, const BatchTables& bt, uint8_t* d_out)
{
	if (bt.n_chunks == 0) { return; }
	hipLaunchKernelGGL(concat_slots_kernel, dim3((bt.n_chunks + 3u) / 4u), dim3(256), 0, st, slots, slot_stride, slot_size, prefix, bt, d_out);
}

__global__ __launch_bounds__(256) void finalize_units_kernel(const u64* __restrict__ prefix, BatchTables bt, uint8_t* __restrict__ d_out,
                                                            u64* __restrict__ d_out_len, int32_t* __restrict__ d_status, int lznt1_eob)
{
	const uint32_t u = blockIdx.x * 256u + threadIdx.x;
	if (u >= bt.n_units) { return; }
	const u64 total = prefix[bt.chunk_prefix[u + 1]] - prefix[bt.chunk_prefix[u]];
	const u64 cap = bt.out_cap[u];
	const bool ok = total <= cap;
	d_out_len[u] = ok ? total : 0;
	d_status[u] = ok ? 0 : -5;                                   // MSCOMP_OK / MSCOMP_BUF_ERROR
	if (ok && lznt1_eob && cap - total >= 2) {                  // End_of_buffer, not counted (lznt1_compress.cpp:270-271)
		uint8_t* e = d_out + bt.out_off[u] + total;
		e[0] = 0; e[1] = 0;
	}
}

void launch_finalize_units(hipStream_t st, const u64* prefix, const BatchTables& bt, uint8_t* d_out,
                           u64* d_out_len, int32_t* d_status, int lznt1_eob)
{
	if (bt.n_units == 0) { return; }
	hipLaunchKernelGGL(finalize_units_kernel, dim3((bt.n_units + 255u) / 256u), dim3(256), 0, st, prefix, bt, d_out, d_out_len, d_status, lznt1_eob);
}

// ---- device-side compaction: the outputs of a batch, packed back to back ----
// packed_off[0..n] = exclusive scan of out_len (one block; n is the number of units of a batch)
__global__ __launch_bounds__(1024) void compact_offsets_kernel(const u64* __restrict__ out_len, u64* __restrict__ packed_off, uint32_t n)
{
	__shared__ u64 s_part[1024];
	const uint32_t tid = threadIdx.x;
	const uint32_t per = (n + 1023u) / 1024u;
	const uint32_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
	u64 sum = 0;
	for (uint32_t i = lo; i < hi; ++i) { sum += out_len[i]; }
	s_part[tid] = sum;
	__syncthreads();
	if (tid == 0) { u64 run = 0; for (uint32_t k = 0; k < 1024u; ++k) { const u64 t = s_part[k]; s_part[k] = run; run += t; } packed_off[n] = run; }
	__syncthreads();
	u64 run = s_part[tid];
	for (uint32_t i = lo; i < hi; ++i) { packed_off[i] = run; run += out_len[i]; }
}
// one block per 64 KiB tile of a unit's capacity (tile_prefix[u] = first tile of unit u); tiles behind the unit's length do nothing
__global__ __launch_bounds__(256) void compact_copy_kernel(const uint8_t* __restrict__ d_out, const u64* __restrict__ out_off, const uint32_t* __restrict__ tile_prefix,
                                                          uint32_t n_units, const u64* __restrict__ out_len, const u64* __restrict__ packed_off, uint8_t* __restrict__ d_packed)
{
	const uint32_t tile = blockIdx.x, tid = threadIdx.x;
	const uint32_t u = unit_of_chunk(tile_prefix, n_units, tile);
	const u64 at = (u64)(tile - tile_prefix[u]) * 65536u, len = out_len[u];
	if (at >= len) { return; }
	const uint32_t n = len - at < 65536u ? (uint32_t)(len - at) : 65536u;
	const uint8_t* src = d_out + out_off[u] + at;
	uint8_t* dst = d_packed + packed_off[u] + at;
	if (((uintptr_t)src & 3u) == 0) { copy_from_aligned(dst, src, n, tid, 256u); }
	else { for (uint32_t i = tid; i < n; i += 256u) { dst[i] = src[i]; } }
}
void launch_compact(hipStream_t st, const uint8_t* d_out, const u64* d_out_off, const uint32_t* d_tile_prefix, uint32_t n_units, uint32_t n_tiles,
                    const u64* d_out_len, u64* d_packed_off, uint8_t* d_packed)
{
	if (n_units == 0) { return; }
	hipLaunchKernelGGL(compact_offsets_kernel, dim3(1), dim3(1024), 0, st, d_out_len, d_packed_off, n_units);
	if (n_tiles) { hipLaunchKernelGGL(compact_copy_kernel, dim3(n_tiles), dim3(256), 0, st, d_out, d_out_off, d_tile_prefix, n_units, d_out_len, d_packed_off, d_packed); }
}

// ---- hardware self-check ---------------------------------------------------------------------------------------
// lznt1_chunk_kernel (bucket ranks) and xp_links_kernel (chain links) rely on one gfx950 behaviour: the returning
// same-address LDS atomics issued by ONE wave instruction are served in LANE ORDER. Every lane adds to the 16-bit
// counter of a pseudo-random key and must read back the number of lower lanes with the same key; exchanges must hand
// the value of the nearest lower lane with the same key (or the initial value) to every lane.
__global__ __launch_bounds__(64) void lds_lane_order_kernel(uint32_t seed, uint32_t rounds, uint32_t nkeys, uint32_t* __restrict__ bad)
{
	__shared__ uint32_t s_cnt[1024];
	__shared__ uint32_t s_head[2048];
	const uint32_t lane = threadIdx.x;
	uint32_t x = seed ^ (blockIdx.x * 0x9E3779B9u) ^ (lane * 0x85EBCA6Bu), nbad = 0;
	for (uint32_t r = 0; r < rounds; ++r) {
		for (uint32_t i = lane; i < 1024u; i += 64u) { s_cnt[i] = 0; }
		for (uint32_t i = lane; i < 2048u; i += 64u) { s_head[i] = 0xFFFFFFFFu; }
		__syncthreads();
		x = x * 1664525u + 1013904223u;
		const uint32_t h = (x >> 8) % nkeys;                          // nkeys <= 2048
		const uint32_t old = atomicAdd(&s_cnt[h >> 1], (h & 1u) ? 0x10000u : 1u);
		const uint32_t rank = (h & 1u) ? old >> 16 : old & 0xFFFFu;
		const uint32_t prev = __hip_atomic_exchange(&s_head[h], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		uint32_t exp_rank = 0, exp_prev = 0xFFFFFFFFu;
		for (uint32_t l = 0; l < 64u; ++l) {
			const uint32_t hl = (uint32_t)__shfl((int)h, (int)l, 64);
			if (l < lane && hl == h) { ++exp_rank; exp_prev = l; }
		}
		nbad += (rank != exp_rank) + (prev != exp_prev);
		__syncthreads();
	}
	if (nbad) { atomicAdd(bad, nbad); }
}

// Two sources, kept apart: the device's own self-check (per device, set once by api.hip lane_order_verdict, never cleared) and
// the test hook (process-wide, mscomp_amd_debug_set_serial_atomics). The hook's "off" only withdraws the hook: a device that failed its
// check keeps the order-independent kernels -- wrong bytes must never leave silently.
static std::atomic<int> g_serial_verdict[64];
static std::atomic<int> g_serial_forced;
bool serial_atomics_on_current_device()
{
	if (g_serial_forced.load(std::memory_order_relaxed) != 0) { return true; }
	int d = 0;
	if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return false; }
	return g_serial_verdict[d < 64 ? d : 63].load(std::memory_order_relaxed) != 0;
}
void set_serial_atomics(int device, int on)            // device >= 0: that device's verdict; device < 0: the process-wide hook
{
	if (device < 0) { g_serial_forced.store(on ? 1 : 0, std::memory_order_relaxed); }
	else { g_serial_verdict[device < 64 ? device : 63].store(on ? 1 : 0, std::memory_order_relaxed); }
}

uint32_t run_lds_lane_order_check(hipStream_t st, uint32_t seed, uint32_t blocks, uint32_t rounds, uint32_t nkeys, uint32_t* d_bad)
{
	if (hipMemsetAsync(d_bad, 0, 4, st) != hipSuccess) { return 0xFFFFFFFFu; }
	hipLaunchKernelGGL(lds_lane_order_kernel, dim3(blocks), dim3(64), 0, st, seed, rounds, nkeys, d_bad);
	uint32_t h = 0xFFFFFFFFu;
	if (hipMemcpyAsync(&h, d_bad, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { return 0xFFFFFFFFu; }
	return h;
}

} // namespace msc

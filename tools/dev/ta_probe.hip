// Dev microbenchmark (MI355X): what a fully DIVERGENT global access costs per wave instruction when every CU does it at once.
// One block of 1024 threads per CU-slot, each block owns a 256 KiB region (L2-resident); every lane touches a pseudo-random
// element of the region. Modes: 0 = u16 store, 1 = u32 store, 2 = u16 load, 3 = 8-byte load, 4 = 16-byte load,
// 5 = u16 store to CONSECUTIVE addresses (coalesced reference), 6 = two u16 stores to two arrays (what the finder does now).
// Prints cycles per wave instruction per CU (block time / instructions issued by the block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REGION (128u * 1024u)   // bytes every access falls into (a chunk's u16 array)
__global__ __launch_bounds__(1024) void k(uint8_t* buf, unsigned long long* out, int mode, int iters, int per_xcd)
{
	// per_xcd > 0: the blocks of an XCD (block id mod 8) share per_xcd regions, so every access is an L2 hit
	const unsigned reg = per_xcd > 0 ? (blockIdx.x & 7u) * per_xcd + ((blockIdx.x >> 3) % per_xcd) : blockIdx.x;
	uint8_t* r = buf + (size_t)reg * REGION * 2u;
	unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	unsigned acc = 0;
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; ++it) {
		#pragma unroll
		for (int j = 0; j < 8; ++j) {
			x = x * 1664525u + 1013904223u;
			const unsigned b = (x >> 8) & (REGION - 32u) ;                          // byte offset, 16-byte aligned
			const unsigned e = b >> 4; (void)e;
			switch (mode) {
			case 0: *reinterpret_cast<uint16_t*>(r + b + (x & 14u)) = (uint16_t)x; break;
			case 1: *reinterpret_cast<uint32_t*>(r + b + (x & 12u)) = x; break;
			case 2: acc += *reinterpret_cast<const uint16_t*>(r + b + (x & 14u)); break;
			case 3: { const uint2 v = *reinterpret_cast<const uint2*>(r + b + (x & 8u)); acc += v.x + v.y; break; }
			case 4: { const uint4 v = *reinterpret_cast<const uint4*>(r + b); acc += v.x + v.w; break; }
			case 5: *reinterpret_cast<uint16_t*>(r + ((it * 8 + j) & 127) * 2048u + threadIdx.x * 2u) = (uint16_t)x; break;
			case 6: *reinterpret_cast<uint16_t*>(r + b + (x & 14u)) = (uint16_t)x; *reinterpret_cast<uint16_t*>(r + REGION + b + (x & 14u)) = (uint16_t)(x >> 16); break;
			case 7: { const uint4 v = *reinterpret_cast<const uint4*>(r + b + (x & 12u)); const uint2 w = *reinterpret_cast<const uint2*>(r + b + (x & 12u) + 16u); acc += v.x + v.w + w.y; break; }
			default: acc += *reinterpret_cast<const uint32_t*>(r + b + (x & 12u)); break;
			}
		}
	}
	__syncthreads();
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
	if (acc == 0x12345678u) { out[0] = acc; }
}
int main()
{
	int ncu = 256; hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); ncu = p.multiProcessorCount;
	uint8_t* buf; unsigned long long* d; hipMalloc(&buf, (size_t)ncu * REGION * 2u); hipMalloc(&d, ncu * 8); hipMemset(buf, 0, (size_t)ncu * REGION * 2u);
	const char* names[] = {"u16 store random", "u32 store random", "u16 load random", "8B load random", "16B load random", "u16 store coalesced", "2 x u16 store random", "24B load random (16+8, 4B aligned)", "u32 load random"};
	const int iters = 256;
	for (int per_xcd : {0, 4}) for (int blocks : {1, ncu}) for (int mode = 0; mode < 9; ++mode) {
		if (per_xcd && blocks == 1) continue;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, buf, d, mode, iters, per_xcd);
		hipDeviceSynchronize();
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, buf, d, mode, iters, per_xcd);
		hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		unsigned long long h[1024]; hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
		double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i]; s /= blocks;
		const double instr = 16.0 * iters * 8 * (mode == 6 || mode == 7 ? 2 : 1);            // wave instructions issued by one block (16 waves)
		printf("share %d blocks %3d %-22s: %.1f counter ticks, %.1f ns per wave instruction per CU (kernel %.3f ms)\n", per_xcd, blocks, names[mode], s / instr, ms * 1e6 / instr, ms);
	}
	return 0;
}

// Dev microbenchmark: cycles per DS instruction (one wave, 8 independent ops in flight) for returning exchange,
// returning add, plain read, plain write on random LDS addresses, with 64 / 32 / 16 / 1 active lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int mode, int active)
{
	__shared__ unsigned heads[32768];
	const unsigned lane = threadIdx.x & 63u;
	for (unsigned i = threadIdx.x; i < 32768; i += blockDim.x) heads[i] = i;
	__syncthreads();
	unsigned x = lane * 2654435761u + threadIdx.x;
	unsigned acc = 0;
	const unsigned long long t0 = __builtin_readcyclecounter();
	if (lane < (unsigned)active) {
		for (int it = 0; it < 1024; ++it) {
			unsigned r[8];
			#pragma unroll
			for (int j = 0; j < 8; ++j) {
				x = x * 1664525u + 1013904223u;
				const unsigned a = (x >> 9) & 32767u;
				if (mode == 0) r[j] = __hip_atomic_exchange(&heads[a], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				else if (mode == 1) r[j] = __hip_atomic_fetch_add(&heads[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				else if (mode == 2) r[j] = __hip_atomic_load(&heads[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				else { __hip_atomic_store(&heads[a], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); r[j] = 0; }
			}
			#pragma unroll
			for (int j = 0; j < 8; ++j) acc += r[j];
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
	if (acc == 0x12345678u) out[2] = acc;
}
int main()
{
	unsigned long long* d; hipMalloc(&d, 64);
	const char* names[] = {"xchg_rtn", "add_rtn", "read", "write"};
	for (int waves = 1; waves <= 2; ++waves)
	for (int mode = 0; mode < 4; ++mode)
	for (int active : {64, 32, 16, 1}) {
		hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, mode, active);
		unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
		printf("waves %d %-9s active %2d: %.1f cycles per DS instruction (wave 0)\n", waves, names[mode], active, (double)h[0] / 8192.0);
	}
	return 0;
}

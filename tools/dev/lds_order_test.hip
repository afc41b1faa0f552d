// Dev experiment: do same-address LDS atomics issued by ONE wave instruction return their old values in lane order?
// Each lane adds to counter[key[lane]] (two u16 counters per dword, like the LZNT1 histogram); the returned old value
// must equal the number of lower lanes with the same key.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const unsigned* keys, unsigned* bad, int rounds, int nkeys)
{
	__shared__ unsigned cnt[1024];
	const unsigned lane = threadIdx.x;
	unsigned nbad = 0;
	for (int r = 0; r < rounds; ++r) {
		for (unsigned i = lane; i < 1024; i += 64) cnt[i] = 0;
		__syncthreads();
		const unsigned h = keys[(size_t)(blockIdx.x * rounds + r) * 64 + lane] % nkeys;
		const unsigned old = atomicAdd(&cnt[h >> 1], (h & 1u) ? 0x10000u : 1u);
		const unsigned rank = (h & 1u) ? old >> 16 : old & 0xFFFFu;
		// expected: lower lanes with the same key
		unsigned exp = 0;
		for (unsigned l = 0; l < 64; ++l) { const unsigned hl = __shfl(h, l, 64); if (l < lane && hl == h) ++exp; }
		if (rank != exp) ++nbad;
		__syncthreads();
	}
	if (nbad) atomicAdd(bad, nbad);
}
int main()
{
	const int blocks = 2048, rounds = 64;
	std::vector<unsigned> h((size_t)blocks * rounds * 64);
	unsigned *dk, *db;
	hipMalloc(&dk, h.size() * 4); hipMalloc(&db, 4);
	const int nk[] = {1, 2, 3, 4, 7, 16, 33, 64, 257, 2048};
	for (int t = 0; t < 10; ++t) {
		srand(1234 + t);
		for (auto& v : h) v = (unsigned)rand();
		hipMemcpy(dk, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(db, 0, 4);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, dk, db, rounds, nk[t]);
		unsigned bad = 0; hipMemcpy(&bad, db, 4, hipMemcpyDeviceToHost);
		printf("nkeys %4d: lanes out of lane order: %u of %zu\n", nk[t], bad, h.size());
	}
	return 0;
}

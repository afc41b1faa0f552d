// dev (GPU box): what a CU of this GPU can ISSUE per cycle -- vector (VALU) and scalar (SALU) wave-instructions, alone and together, with 1..8 waves
// per SIMD. Settles how to read SQ_INSTS_VALU / SQ_INSTS_SALU against a kernel's duration ("is 1.0 VALU instruction per CU cycle a full pipe?").
//   hipcc --offload-arch=gfx950 -O2 -o tools/dev/issue_peak tools/dev/issue_peak.hip && tools/dev/issue_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(x) x x x x x x x x x x x x x x x x
// 64 independent-enough vector adds per iteration (8 accumulators), no memory
__global__ void valu_kernel(uint32_t* out, int iters)
{
	uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
	for (int i = 0; i < iters; ++i) {
		asm volatile(REP16("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n")
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
		asm volatile(REP16("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0\n")
		             : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
	}
	if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) { out[0] = 1; }
}
// the same with v_alignbyte_b32 (a VOP3 instruction, what the 16-byte compares are made of)
__global__ void valu3_kernel(uint32_t* out, int iters)
{
	uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
	for (int i = 0; i < iters; ++i) {
		asm volatile(REP16("v_alignbyte_b32 %0, %0, %1, 1\n v_alignbyte_b32 %1, %1, %2, 2\n v_alignbyte_b32 %2, %2, %3, 3\n v_alignbyte_b32 %3, %3, %0, 1\n")
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
		asm volatile(REP16("v_min3_u32 %0, %0, %1, %2\n v_min3_u32 %1, %1, %2, %3\n v_min3_u32 %2, %2, %3, %0\n v_min3_u32 %3, %3, %0, %1\n")
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
	}
	if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345u) { out[0] = 1; }
}
// 128 scalar adds per iteration (4 accumulators)
__global__ void salu_kernel(uint32_t* out, int iters)
{
	uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
	for (int i = 0; i < iters; ++i) {
		asm volatile(REP16("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n")
		             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
		asm volatile(REP16("s_xor_b32 %0, %0, %1\n s_xor_b32 %1, %1, %2\n s_xor_b32 %2, %2, %3\n s_xor_b32 %3, %3, %0\n")
		             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
	}
	if ((s0 ^ s1 ^ s2 ^ s3) == 0x12345u) { out[0] = 1; }
}
// both: 64 vector + 64 scalar per iteration, interleaved
__global__ void mixed_kernel(uint32_t* out, int iters)
{
	uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
	uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
	for (int i = 0; i < iters; ++i) {
		asm volatile(REP16("v_add_u32 %0, %0, %0\n s_add_u32 %4, %4, %5\n v_add_u32 %1, %1, %1\n s_add_u32 %5, %5, %6\n v_add_u32 %2, %2, %2\n s_add_u32 %6, %6, %7\n v_add_u32 %3, %3, %3\n s_add_u32 %7, %7, %4\n")
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
	}
	if ((a0 ^ a1 ^ a2 ^ a3 ^ s0 ^ s1 ^ s2 ^ s3) == 0x12345u) { out[0] = 1; }
}
// VOP3 + scalar interleaved in one wave
__global__ void mixed3_kernel(uint32_t* out, int iters)
{
	uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
	uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
	for (int i = 0; i < iters; ++i) {
		asm volatile(REP16("v_alignbyte_b32 %0, %0, %1, 1\n s_add_u32 %4, %4, %5\n v_alignbyte_b32 %1, %1, %2, 2\n s_add_u32 %5, %5, %6\n v_alignbyte_b32 %2, %2, %3, 3\n s_add_u32 %6, %6, %7\n v_alignbyte_b32 %3, %3, %0, 1\n s_add_u32 %7, %7, %4\n")
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
	}
	if ((a0 ^ a1 ^ a2 ^ a3 ^ s0 ^ s1 ^ s2 ^ s3) == 0x12345u) { out[0] = 1; }
}
// vector work and scalar work in DIFFERENT waves of a block: even waves run the vector loop, odd waves the scalar loop (can a CU issue both at their own peaks?)
__global__ void split_kernel(uint32_t* out, int iters)
{
	uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
	uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
	if ((threadIdx.x >> 6) & 1u) {
		for (int i = 0; i < iters; ++i) {
			asm volatile(REP16("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
			asm volatile(REP16("s_xor_b32 %0, %0, %1\n s_xor_b32 %1, %1, %2\n s_xor_b32 %2, %2, %3\n s_xor_b32 %3, %3, %0\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
		}
	} else {
		for (int i = 0; i < iters; ++i) {
			asm volatile(REP16("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
			asm volatile(REP16("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
		}
	}
	if ((a0 ^ a1 ^ a2 ^ a3 ^ s0 ^ s1 ^ s2 ^ s3) == 0x12345u) { out[0] = 1; }
}
template <class K> static double run(K k, int blocks, int threads, int iters, uint32_t* d)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 10);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	return ms;
}
int main()
{
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	uint32_t* d; hipMalloc(&d, 64);
	printf("%s: %d CUs, clock (property) %.0f MHz\n", p.name, cus, p.clockRate / 1000.0);
	const int iters = 20000;
	for (int wps = 1; wps <= 8; wps *= 2) {                       // waves per SIMD (4 SIMDs per CU): blocks of 256 threads = one wave per SIMD
		const int blocks = cus * wps;
		const double waves = (double)blocks * 4;
		const double tv = run(valu_kernel, blocks, 256, iters, d), t3 = run(valu3_kernel, blocks, 256, iters, d);
		const double ts = run(salu_kernel, blocks, 256, iters, d), tm = run(mixed_kernel, blocks, 256, iters, d);
		// wave-instructions per CU per ns
		const double v = waves * iters * 128.0 / cus / (tv * 1e6), v3 = waves * iters * 128.0 / cus / (t3 * 1e6);
		const double s = waves * iters * 128.0 / cus / (ts * 1e6);
		const double mv = waves * iters * 64.0 / cus / (tm * 1e6);
		const double tm3 = run(mixed3_kernel, blocks, 256, iters, d), tsp = run(split_kernel, blocks, 256, iters, d);
		const double m3 = waves * iters * 64.0 / cus / (tm3 * 1e6);
		const double sp = waves / 2 * iters * 128.0 / cus / (tsp * 1e6);    // per kind: half the waves each (both finish together only if they issue at the same rate: a lower bound for the faster kind)
		printf("waves/SIMD %d | VALU (v_add/v_xor) %.3f  VOP3 (alignbyte/min3) %.3f  SALU %.3f | interleaved in one wave: VOP2 %.3f + SALU %.3f, VOP3 %.3f + SALU %.3f | vector and scalar in different waves (half each, until the slower half ends): %.3f + %.3f   wave-instructions per CU per ns (divide by the clock in GHz for per cycle)\n", wps, v, v3, s, mv, mv, m3, m3, sp, sp);
	}
	return 0;
}

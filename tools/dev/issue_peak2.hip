// dev (GPU box): issue rate of single VALU instruction forms (wave-instructions per CU per cycle at 8 waves per SIMD, four independent chains,
// no memory) -- which forms run at the ~1.9 / cycle of a plain 4-byte VOP1/VOP2 and which at ~1 / cycle. Follows tools/dev/issue_peak.hip.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o tools/dev/issue_peak2 tools/dev/issue_peak2.hip && tools/dev/issue_peak2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, body)                                                                                   \
__global__ void name(uint32_t* out, int iters)                                                                \
{                                                                                                             \
	uint32_t a0 = threadIdx.x, a1 = threadIdx.x * 3u + 1u, a2 = threadIdx.x ^ 5u, a3 = 3u + threadIdx.x;       \
	uint32_t s0 = 0x12345u + blockIdx.x;                                                                        \
	for (int i = 0; i < iters; ++i) {                                                                           \
		asm volatile(REP16(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0) : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc"); \
	}                                                                                                           \
	if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345u) { out[0] = 1; }                                                        \
}
#define Q(a) a "\n"
// each body = 4 instructions (8 for the pairs)
KERNEL(k_add_e32,     Q("v_add_u32 %0, %0, %0") Q("v_add_u32 %1, %1, %1") Q("v_add_u32 %2, %2, %2") Q("v_add_u32 %3, %3, %3"))
KERNEL(k_add_lit,     Q("v_add_u32 %0, 0x12345, %0") Q("v_add_u32 %1, 0x12345, %1") Q("v_add_u32 %2, 0x12345, %2") Q("v_add_u32 %3, 0x12345, %3"))
KERNEL(k_add_sgpr,    Q("v_add_u32 %0, %4, %0") Q("v_add_u32 %1, %4, %1") Q("v_add_u32 %2, %4, %2") Q("v_add_u32 %3, %4, %3"))
KERNEL(k_add_inline,  Q("v_add_u32 %0, 17, %0") Q("v_add_u32 %1, 17, %1") Q("v_add_u32 %2, 17, %2") Q("v_add_u32 %3, 17, %3"))
KERNEL(k_add_e64,     Q("v_add_u32_e64 %0, %0, %0") Q("v_add_u32_e64 %1, %1, %1") Q("v_add_u32_e64 %2, %2, %2") Q("v_add_u32_e64 %3, %3, %3"))
KERNEL(k_add_dpp,     Q("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") Q("v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf") Q("v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf") Q("v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_add_sdwa,    Q("v_add_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD") Q("v_add_u32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD") Q("v_add_u32_sdwa %2, %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD") Q("v_add_u32_sdwa %3, %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD"))
KERNEL(k_and_lit,     Q("v_and_b32 %0, 0xffffff, %0") Q("v_and_b32 %1, 0xffffff, %1") Q("v_and_b32 %2, 0xffffff, %2") Q("v_and_b32 %3, 0xffffff, %3"))
KERNEL(k_logic_e32,   Q("v_and_b32 %0, %0, %1") Q("v_or_b32 %1, %1, %2") Q("v_lshlrev_b32 %2, 1, %2") Q("v_lshrrev_b32 %3, 1, %3"))
KERNEL(k_ffbl_mov,    Q("v_ffbl_b32 %0, %0") Q("v_mov_b32 %1, %2") Q("v_ffbl_b32 %2, %2") Q("v_mov_b32 %3, %0"))
KERNEL(k_cmp_cnd_e32, Q("v_cmp_lt_u32 vcc, %0, %1") Q("v_cndmask_b32 %0, %0, %1, vcc") Q("v_cmp_lt_u32 vcc, %2, %3") Q("v_cndmask_b32 %2, %2, %3, vcc"))
KERNEL(k_cmp_cnd_e64, Q("v_cmp_lt_u32 s[40:41], %0, %1") Q("v_cndmask_b32 %0, %0, %1, s[40:41]") Q("v_cmp_lt_u32 s[42:43], %2, %3") Q("v_cndmask_b32 %2, %2, %3, s[42:43]"))
KERNEL(k_cmp_e32,     Q("v_cmp_lt_u32 vcc, %0, %1") Q("v_cmp_lt_u32 vcc, %1, %2") Q("v_cmp_lt_u32 vcc, %2, %3") Q("v_cmp_lt_u32 vcc, %3, %0"))
KERNEL(k_cmp_e64,     Q("v_cmp_lt_u32 s[40:41], %0, %1") Q("v_cmp_lt_u32 s[42:43], %1, %2") Q("v_cmp_lt_u32 s[44:45], %2, %3") Q("v_cmp_lt_u32 s[46:47], %3, %0"))
KERNEL(k_readlane,    Q("v_readlane_b32 s40, %0, 3") Q("v_readlane_b32 s41, %1, 5") Q("v_readlane_b32 s42, %2, 7") Q("v_readlane_b32 s43, %3, 9"))
KERNEL(k_readfirst,   Q("v_readfirstlane_b32 s40, %0") Q("v_readfirstlane_b32 s41, %1") Q("v_readfirstlane_b32 s42, %2") Q("v_readfirstlane_b32 s43, %3"))
KERNEL(k_lshl_add,    Q("v_lshl_add_u32 %0, %0, 1, %1") Q("v_lshl_add_u32 %1, %1, 1, %2") Q("v_lshl_add_u32 %2, %2, 1, %3") Q("v_lshl_add_u32 %3, %3, 1, %0"))
KERNEL(k_add3,        Q("v_add3_u32 %0, %0, %1, %2") Q("v_add3_u32 %1, %1, %2, %3") Q("v_add3_u32 %2, %2, %3, %0") Q("v_add3_u32 %3, %3, %0, %1"))
KERNEL(k_bfe_perm,    Q("v_bfe_u32 %0, %0, 3, 20") Q("v_perm_b32 %1, %1, %2, %3") Q("v_bfe_u32 %2, %2, 3, 20") Q("v_perm_b32 %3, %3, %0, %1"))
KERNEL(k_mbcnt,       Q("v_mbcnt_lo_u32_b32 %0, %0, %1") Q("v_mbcnt_hi_u32_b32 %1, %1, %2") Q("v_mbcnt_lo_u32_b32 %2, %2, %3") Q("v_mbcnt_hi_u32_b32 %3, %3, %0"))
KERNEL(k_mul_lo,      Q("v_mul_lo_u32 %0, %0, %1") Q("v_mul_lo_u32 %1, %1, %2") Q("v_mul_lo_u32 %2, %2, %3") Q("v_mul_lo_u32 %3, %3, %0"))
KERNEL(k_mul_u24,     Q("v_mul_u32_u24 %0, %0, %1") Q("v_mul_u32_u24 %1, %1, %2") Q("v_mul_u32_u24 %2, %2, %3") Q("v_mul_u32_u24 %3, %3, %0"))
KERNEL(k_alignbyte,   Q("v_alignbyte_b32 %0, %0, %1, 1") Q("v_alignbyte_b32 %1, %1, %2, 2") Q("v_alignbyte_b32 %2, %2, %3, 3") Q("v_alignbyte_b32 %3, %3, %0, 1"))
KERNEL(k_mix_1to1,    Q("v_add_u32 %0, %0, %0") Q("v_alignbyte_b32 %1, %1, %2, 2") Q("v_add_u32 %2, %2, %2") Q("v_alignbyte_b32 %3, %3, %0, 1"))
KERNEL(k_mix_3to1,    Q("v_add_u32 %0, %0, %0") Q("v_add_u32 %1, %1, %1") Q("v_add_u32 %2, %2, %2") Q("v_alignbyte_b32 %3, %3, %0, 1"))
template <class K> static double run(K k, int blocks, int iters, uint32_t* d)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 10);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	return ms;
}
int main()
{
	hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	uint32_t* d; (void)hipMalloc(&d, 64);
	const int iters = 20000;
	const double ref_ms = run(k_add_e32, cus * 8, iters, d);
	printf("%s: %d CUs; wave-instructions per CU per ns and relative to the plain 4-byte v_add_u32 (8 waves per SIMD | 2 waves per SIMD)\n", p.name, cus);
#define ROW(k, label) { const double t8 = run(k, cus * 8, iters, d), t2 = run(k, cus * 2, iters, d); \
	printf("%-44s %.3f (%.2f) | %.3f\n", label, 32.0 * iters * 64.0 / (t8 * 1e6), ref_ms / t8, 8.0 * iters * 64.0 / (t2 * 1e6)); }
	ROW(k_add_e32,     "v_add_u32 v,v,v (VOP2, 4 bytes)")
	ROW(k_add_inline,  "v_add_u32 v,17,v (inline constant)")
	ROW(k_add_sgpr,    "v_add_u32 v,s,v (SGPR operand)")
	ROW(k_add_lit,     "v_add_u32 v,0x12345,v (32-bit literal, 8 B)")
	ROW(k_and_lit,     "v_and_b32 v,0xffffff,v (literal, 8 B)")
	ROW(k_add_e64,     "v_add_u32_e64 (VOP3 encoding, 8 B)")
	ROW(k_add_dpp,     "v_add_u32_dpp row_shr:1 (8 B)")
	ROW(k_add_sdwa,    "v_add_u32_sdwa (8 B)")
	ROW(k_logic_e32,   "v_and / v_or / v_lshlrev / v_lshrrev (4 B)")
	ROW(k_ffbl_mov,    "v_ffbl_b32 / v_mov_b32 (VOP1)")
	ROW(k_cmp_e32,     "v_cmp_lt_u32 vcc (VOPC, 4 B)")
	ROW(k_cmp_e64,     "v_cmp_lt_u32 s[n:n+1] (VOP3, 8 B)")
	ROW(k_cmp_cnd_e32, "v_cmp vcc + v_cndmask vcc (4 B each)")
	ROW(k_cmp_cnd_e64, "v_cmp sgpr + v_cndmask sgpr (8 B each)")
	ROW(k_readlane,    "v_readlane_b32")
	ROW(k_readfirst,   "v_readfirstlane_b32")
	ROW(k_lshl_add,    "v_lshl_add_u32 (VOP3)")
	ROW(k_add3,        "v_add3_u32 (VOP3)")
	ROW(k_bfe_perm,    "v_bfe_u32 / v_perm_b32 (VOP3)")
	ROW(k_mbcnt,       "v_mbcnt_lo / hi (VOP3)")
	ROW(k_mul_lo,      "v_mul_lo_u32 (VOP3)")
	ROW(k_mul_u24,     "v_mul_u32_u24 (VOP2)")
	ROW(k_alignbyte,   "v_alignbyte_b32 (VOP3)")
	ROW(k_mix_1to1,    "v_add_u32 : v_alignbyte 1 : 1")
	ROW(k_mix_3to1,    "v_add_u32 : v_alignbyte 3 : 1")
	return 0;
}

// study (round 5): the LAZY Xpress+Huffman finder with the chunk's links in LDS and the data from L2 -- lockstep simulation of the
// claimed walks of one 1024-lane block per 64 KiB chunk (lane = 64-byte segment, one token per lane and wave step), counting what the
// design would pay: wave steps, lane occupancy, chain-walk iterations (max over the wave's lanes), candidate gathers, links that lie
// further back in the previous chunk than the PT positions kept in LDS, extension round trips.
//   cc -O2 -o /tmp/xh_lazy_study tools/dev/xh_lazy_study.c && /tmp/xh_lazy_study /tmp/corpus/<file> [max_bytes]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static uint32_t hash3(const uint8_t* d) { return (((d[0] & 0x1Fu) << 10) ^ ((uint32_t)d[1] << 5) ^ d[2]) & 0x7FFF; }
#define SEG 64u
#define NL (65536u / SEG)
int main(int argc, char** argv)
{
	FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t N = ftell(f); fseek(f, 0, SEEK_SET);
	if (argc > 2 && (size_t)atol(argv[2]) < N) { N = atol(argv[2]); }
	uint8_t* d = malloc(N + 64); memset(d + N, 0, 64); if (fread(d, 1, N, f) != N) return 1; fclose(f);
	int32_t* link = malloc(N * 4); int32_t* head = malloc(32768 * 4);
	for (int i = 0; i < 32768; ++i) head[i] = -1;
	for (size_t p = 0; p + 2 < N; ++p) { uint32_t h = hash3(d + p); link[p] = head[h]; head[h] = (int32_t)p; }
	for (size_t p = (N >= 2 ? N - 2 : 0); p < N; ++p) link[p] = -1;
	uint8_t* claimed = calloc(N + 1, 1);
	const uint32_t PT[3] = {4096, 8192, 12288};
	double claims = 0, truetok = 0, wsteps = 0, crit = 0, lanes_active = 0, walk_iters = 0, cands = 0, far[3] = {0}, ext_iters = 0, coop = 0, chunks = 0, nolink = 0, cmp16 = 0;
	// the true parse (for the token count)
	for (size_t p = 0; p < N;) {
		uint32_t best = 2; truetok++;
		uint32_t inch = 65536 - (p & 65535);
		if (p + 2 < N && inch >= 3) {
			uint32_t cap = N - p - 1; int32_t x = link[p]; uint32_t chain = 11;
			while (chain && x >= 0 && p - x <= 0xFFFF) { uint32_t l = 0; while (l < cap && d[x + l] == d[p + l]) l++; if (l > best) { best = l; if (best >= 48) break; } x = link[x]; chain--; }
		}
		if (best >= 3) { if (best > inch) best = inch; p += best; } else p++;
	}
	for (size_t cs = 0; cs < N; cs += 65536) {
		const uint32_t cn = N - cs < 65536 ? (uint32_t)(N - cs) : 65536u;
		chunks++;
		for (uint32_t o = 0; o < cn; ++o) { if (cs + o + 2 < N && (link[cs + o] < 0 || (size_t)link[cs + o] < cs)) nolink++; }
		uint32_t q[NL]; uint8_t alive[NL];
		for (uint32_t l = 0; l < NL; ++l) { q[l] = l * SEG; alive[l] = q[l] < cn; }
		uint32_t steps_w[16] = {0};
		int any = 1;
		while (any) {
			any = 0;
			for (uint32_t w = 0; w < 16; ++w) {
				uint32_t act = 0, maxchain = 0, maxext = 0;
				for (uint32_t l = w * 64; l < w * 64 + 64; ++l) {
					if (!alive[l]) continue;
					const size_t p = cs + q[l];
					if (claimed[p]) { alive[l] = 0; continue; }
					claimed[p] = 1; claims++; act++;
					uint32_t best = 2, chainlen = 0;
					const uint32_t inch = cn - q[l];
					if (p + 2 < N && inch >= 3) {
						uint32_t cap = N - p - 1; int32_t x = link[p]; uint32_t chain = 11;
						size_t src = p;           // the position whose link we just followed
						while (chain && x >= 0 && p - x <= 0xFFFF) {
							// the link that led here was stored at src: in LDS if src >= cs - PT
							for (int k = 0; k < 3; ++k) { if (src + PT[k] < cs) far[k]++; }
							chainlen++; cands++;
							uint32_t l2 = 0; while (l2 < cap && d[x + l2] == d[p + l2]) l2++;
							if (l2 >= 16) cmp16++;
							if (l2 > best) { best = l2; }
							src = (size_t)x; x = link[x]; chain--;   /* (no early break: all <= 11 are gathered) */
						}
					}
					if (chainlen > maxchain) maxchain = chainlen;
					uint32_t len = 1;
					if (best >= 3) { len = best > inch ? inch : best; }
					if (best >= 16) { uint32_t e = (best > 112 ? 112 : best); uint32_t it = (e - 16) / 16 + 1; if (it > maxext) maxext = it; if (best > 112) coop++; }
					q[l] += len;
					if (q[l] >= cn) alive[l] = 0;
				}
				if (act) { any = 1; steps_w[w]++; wsteps++; lanes_active += act; walk_iters += maxchain; ext_iters += maxext; }
			}
		}
		uint32_t mx = 0; for (int w = 0; w < 16; ++w) if (steps_w[w] > mx) mx = steps_w[w];
		crit += mx;
	}
	const char* nm = strrchr(argv[1], '/'); nm = nm ? nm + 1 : argv[1];
	printf("%-8s true tok/B %.3f claims/B %.3f (x%.2f) | per chunk: wave steps %.0f, slowest wave %.0f | per step: lanes %.1f, walk iters %.1f, ext iters %.2f | cand/claim %.2f, >=16B %.3f of cands, coop/claim %.4f | far links / cand: PT4K %.3f 8K %.3f 12K %.3f | no-in-chunk-link positions / chunk %.0f\n",
	       nm, truetok / N, claims / N, claims / truetok, wsteps / chunks, crit / chunks, lanes_active / wsteps, walk_iters / wsteps, ext_iters / wsteps, cands / claims, cmp16 / cands, coop / claims,
	       far[0] / cands, far[1] / cands, far[2] / cands, nolink / chunks);
	return 0;
}

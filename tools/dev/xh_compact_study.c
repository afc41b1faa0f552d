// study: wave-steps of the XH all-positions chain walk with and without compaction of long-chain positions
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static uint32_t hash3(const uint8_t* d) { return (((d[0] & 0x1Fu) << 10) ^ ((uint32_t)d[1] << 5) ^ d[2]) & 0x7FFF; }
int main(int argc, char** argv)
{
	FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t N = ftell(f); fseek(f, 0, SEEK_SET);
	uint8_t* d = malloc(N + 64); memset(d + N, 0, 64); if (fread(d, 1, N, f) != N) return 1; fclose(f);
	const uint32_t maxoff = 0xFFFF;
	int32_t* link = malloc(N * 4); int32_t* head = malloc(32768 * 4);
	uint8_t* ncand = malloc(N);
	for (int i = 0; i < 32768; ++i) head[i] = -1;
	for (size_t p = 0; p + 2 < N; ++p) { uint32_t h = hash3(d + p); link[p] = head[h]; head[h] = (int32_t)p; }
	for (size_t p = (N >= 2 ? N - 2 : 0); p < N; ++p) link[p] = -1;
	double hist[13] = {0};
	for (size_t p = 0; p < N; ++p) {
		uint32_t best = 2, cnt = 0;
		if (p + 2 < N) {
			uint32_t cap = N - p - 1 < 48 ? N - p - 1 : 48;
			int32_t x = link[p]; uint32_t chain = 11;
			while (cap && chain && x >= 0 && p - x <= maxoff) {
				cnt++;
				uint32_t l = 0; while (l < cap && d[x + l] == d[p + l]) l++;
				if (l > best) { best = l; if (best >= 48) break; }
				x = link[x]; chain--;
			}
		}
		ncand[p] = cnt; hist[cnt]++;
	}
	double now = 0, useful = 0;
	for (size_t w = 0; w < N; w += 64) { uint32_t m = 0; for (size_t p = w; p < w + 64 && p < N; ++p) { if (ncand[p] > m) m = ncand[p]; useful += ncand[p]; } now += m; }
	printf("%-10s N %zu  steps/pos %.2f  wave-steps now %.0f (eff %.2f) | hist:", argv[1] + 12, N, useful / N, now, useful / (now * 64));
	for (int i = 0; i <= 11; ++i) printf(" %.3f", hist[i] / N);
	printf("\n");
	for (uint32_t K = 1; K <= 6; ++K) {
		// tile = 8192 positions, 16 waves: wave v handles rounds of 64; survivors compact per WAVE in order
		double p1 = 0, p2 = 0, nb = 0, surv = 0;
		for (size_t t = 0; t < N; t += 8192) {
			for (uint32_t wv = 0; wv < 16; ++wv) {
				uint32_t q[128]; uint32_t nq = 0;
				for (uint32_t r = 0; r < 8; ++r) {
					size_t w = t + (size_t)(r * 16 + wv) * 64; uint32_t m = 0;
					for (size_t p = w; p < w + 64 && p < N; ++p) {
						uint32_t c = ncand[p]; uint32_t c1 = c < K ? c : K; if (c1 > m) m = c1;
						if (c > K) { q[nq++] = c - K; surv++; }
						if (nq == 64) { uint32_t mm = 0; for (int i = 0; i < 64; ++i) if (q[i] > mm) mm = q[i]; p2 += mm; nb++; nq = 0; }
					}
					p1 += m;
				}
				if (nq) { uint32_t mm = 0; for (uint32_t i = 0; i < nq; ++i) if (q[i] > mm) mm = q[i]; p2 += mm; nb++; }
			}
		}
		printf("   K=%u: pass1 %.0f + pass2 %.0f (batches %.0f, survivors %.3f/pos) = %.0f  ratio to now %.3f; with overhead (1.5 steps/batch, 0.5/round) %.3f\n", K, p1, p2, nb, surv / N, p1 + p2, (p1 + p2) / now,
		       (p1 + p2 + 1.5 * nb + 0.5 * (N / 64.0)) / now);
	}
	return 0;
}

// study: quad finishing with CHAINED speculation: the targets of a step are the landed position and the next three landings of a walk that assumes
// every unfinished position keeps its eager result (the best of the first SELF candidates)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static int HB = 11;
static uint32_t hash(const uint8_t* d) { uint32_t k = d[0] | (d[1] << 8) | (d[2] << 16); return (k * 0x9E3779B1u) >> (32 - HB); }
static uint32_t shift_of(uint32_t pos) { if (pos <= 16) return 12; uint32_t b = 32 - __builtin_clz(pos - 1); return 12 - (b - 4); }
int main(int argc, char** argv)
{
	FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t N = ftell(f); fseek(f, 0, SEEK_SET);
	uint8_t* d = malloc(N + 64); memset(d + N, 0, 64); if (fread(d, 1, N, f) != N) return 1; fclose(f);
	const uint32_t SELF = 4, G = argc > 2 ? atoi(argv[2]) : 4, GL = argc > 3 ? atoi(argv[3]) : 16; const int inwin = argc > 4 ? atoi(argv[4]) : 1;
	double chunks = 0, events = 0, quads = 0, hits = 0, same = 0, spec = 0, wide = 0, wide0 = 0, hops = 0, olds = 0;
	static uint16_t bucket[4096]; static uint32_t bstart[65537]; static uint16_t rnk[4096];
	for (size_t cb = 0; cb < N; cb += 4096) {
		const uint8_t* c = d + cb; uint32_t n = N - cb < 4096 ? N - cb : 4096; chunks++;
		static uint32_t cnt[65536]; memset(cnt, 0, sizeof(uint32_t) << HB);
		for (uint32_t p = 0; p + 2 < n; ++p) { rnk[p] = cnt[hash(c + p)]++; }
		bstart[0] = 0; for (int h = 0; h < (1 << HB); ++h) bstart[h + 1] = bstart[h] + cnt[h];
		for (uint32_t p = 0; p + 2 < n; ++p) bucket[bstart[hash(c + p)] + rnk[p]] = p;
		static uint16_t blen[4096], elen[4096], ncand[4096]; static uint8_t res4[4096];
		for (uint32_t p = 0; p < n; ++p) {
			blen[p] = 0; elen[p] = 0; ncand[p] = 0; res4[p] = 1;
			if (p == 0 || p + 3 > n) continue;
			uint32_t mask3 = (1u << shift_of(p)) + 2, maxlen = n - p < mask3 ? n - p : mask3;
			uint32_t h = hash(c + p), nc = rnk[p], best = 0; int done = 0;
			ncand[p] = nc;
			for (uint32_t j = 0; j < nc; ++j) {
				uint32_t q = bucket[bstart[h] + j], l = 0; while (l < maxlen && c[q + l] == c[p + l]) l++;
				if (l >= 3 && l > best) { best = l; if (l == maxlen) done = 1; }
				if (j + 1 == SELF) { elen[p] = best; if (!done && nc > SELF) res4[p] = 0; }
				if (done) break;
			}
			if (nc <= SELF) elen[p] = best;
			blen[p] = best;
		}
		uint8_t st[4096];                       // 1 resolved, 0 unresolved, 2 pre-finished
		for (uint32_t p = 0; p < n; ++p) st[p] = res4[p];
		uint32_t p = 0;
		while (p < n) {
			if (st[p] == 0) {
				events++;
				same += blen[p] == elen[p];
				uint32_t rem = ncand[p] - SELF; wide0 += (rem + 63) / 64 - 1;
				st[p] = 1;
				if (rem > GL) { olds++; wide += (rem + 63) / 64 - 1; }
				else {
				quads++;
				// chained speculation
				uint32_t q = p, g = 1; const uint32_t wend = ((p >> 6) + 1) * 64 < n ? ((p >> 6) + 1) * 64 : n;
				uint32_t steplen = elen[q] >= 3 ? elen[q] : 1;
				q += steplen;
				while (g < G && q < (inwin ? wend : n)) {
					if (st[q] == 0) { hops++; if (ncand[q] - SELF <= GL) { st[q] = 2; spec++; } g++; steplen = elen[q] >= 3 ? elen[q] : 1; q += steplen; }
					else if (st[q] == 2 || blen[q] >= 3) { hops++; steplen = blen[q] >= 3 ? blen[q] : 1; q += steplen; }   /* a stop: one hop */
					else { q += 1; }                                   /* a resolved literal is no stop: J jumps over it */
				}
				}
			} else if (st[p] == 2) { events++; hits++; st[p] = 1; }
			p += blen[p] >= 3 ? blen[p] : 1;
		}
	}
	const double c_old = (events * 115 + wide0 * 70) / chunks, c_new = (quads * 125 + hops * 4 + hits * 20 + olds * 118 + wide * 70) / chunks;
	printf("%-9s events/chunk %.1f: quad steps %.1f (hops %.1f) + pre-finished landings %.1f (speculated %.1f, hit rate %.2f) + old-style events %.1f | wide steps %.1f (today %.1f) | modelled finishing instructions/chunk %.0f -> %.0f (%.2f)\n",
	       strrchr(argv[1], '/') + 1, events / chunks, quads / chunks, hops / chunks, hits / chunks, spec / chunks, hits / (spec > 0 ? spec : 1), olds / chunks, wide / chunks, wide0 / chunks, c_old, c_new, c_new / c_old);
	return 0;
}

// lz_finish_study.c -- CPU study (dev tool): how often lznt1_chunk4_kernel's lazy parse has to "finish" a position with the whole wave, how many
// candidates such a position has left, and how many of these events a scheme that finishes the next unresolved positions of the window in the
// same step (16 lanes each) would save.   usage: lz_finish_study <file> [self=4] [groups=4] [glanes=16]
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
	if (argc > 5) HB = atoi(argv[5]);
	const uint32_t SELF = argc > 2 ? atoi(argv[2]) : 4, G = argc > 3 ? atoi(argv[3]) : 4, GL = argc > 4 ? atoi(argv[4]) : 16;
	double chunks = 0, windows = 0, events = 0, iters = 0, events2 = 0, iters2 = 0, tok = 0, unres_total = 0, spec = 0, spec_used = 0;
	double remh[8] = {0};
	static uint16_t bucket[4096]; static uint32_t bstart[65537]; static uint16_t rnk[4096]; double coll = 0, cands = 0;
	for (size_t cb = 0; cb < N; cb += 4096) {
		const uint8_t* c = d + cb; uint32_t n = N - cb < 4096 ? N - cb : 4096; chunks++;
		static uint32_t cnt[65536]; memset(cnt, 0, sizeof(uint32_t) << HB);
		for (uint32_t p = 0; p + 2 < n; ++p) { rnk[p] = cnt[hash(c + p)]++; }
		bstart[0] = 0; for (int h = 0; h < (1 << HB); ++h) bstart[h + 1] = bstart[h] + cnt[h];
		for (uint32_t p = 0; p + 2 < n; ++p) bucket[bstart[hash(c + p)] + rnk[p]] = p;
		// per position: number of candidates (older entries), best match (exact Find: oldest-first, strictly longer, stop at maxlen), and
		// whether the first SELF candidates resolve it
		static uint16_t blen[4096], ncand[4096]; static uint8_t res4[4096];
		for (uint32_t p = 0; p < n; ++p) {
			blen[p] = 0; ncand[p] = 0; res4[p] = 1;
			if (p == 0 || p + 3 > n) continue;
			uint32_t mask3 = (1u << shift_of(p)) + 2, maxlen = n - p < mask3 ? n - p : mask3;
			uint32_t h = hash(c + p), nc = rnk[p], best = 0; int done = 0;
			ncand[p] = nc;
			for (uint32_t j = 0; j < nc; ++j) {
				uint32_t q = bucket[bstart[h] + j], l = 0; cands++; coll += memcmp(c + q, c + p, 3) != 0; while (l < maxlen && c[q + l] == c[p + l]) l++;
				if (l >= 3 && l > best) { best = l; if (l == maxlen) { done = 1; if (j >= SELF) {} } }
				if (j + 1 == SELF && !done && nc > SELF) res4[p] = 0;
				if (done) break;
			}
			blen[p] = best;
		}
		// lazy walk window by window: events = positions the walk lands on that are unresolved (res4 == 0)
		uint8_t resolved[4096], resolved2[4096];
		for (uint32_t p = 0; p < n; ++p) { resolved[p] = res4[p]; resolved2[p] = res4[p]; unres_total += !res4[p]; }
		uint32_t p = 0;
		while (p < n) {
			tok++;
			uint32_t w = p >> 6, wend = (w + 1) * 64 < n ? (w + 1) * 64 : n;
			if (!resolved[p]) { events++; uint32_t rem = ncand[p] - SELF; iters += (rem + 63) / 64; remh[rem <= 4 ? 0 : rem <= 8 ? 1 : rem <= 16 ? 2 : rem <= 32 ? 3 : rem <= 64 ? 4 : rem <= 128 ? 5 : rem <= 256 ? 6 : 7]++; resolved[p] = 1; }
			if (!resolved2[p]) {
				events2++;
				uint32_t rem = ncand[p] - SELF;
				if (rem <= GL * 2) {
					iters2 += (rem + GL - 1) / GL; resolved2[p] = 1;
					uint32_t g = 1;
					for (uint32_t q = p + 1; q < wend && g < G; ++q) if (!resolved2[q]) { if (ncand[q] - SELF <= GL * ((rem + GL - 1) / GL)) { resolved2[q] = 2; spec++; } g++; }
				} else { iters2 += (rem + 63) / 64; resolved2[p] = 1; }
			} else if (resolved2[p] == 2) { spec_used++; }
			p += blen[p] >= 3 ? blen[p] : 1;
		}
		windows += (n + 63) / 64;
	}
	printf("%-9s self %u: chunks %.0f tok/chunk %.0f unresolved/chunk %.0f | finishing events/chunk %.1f (%.2f per window) steps/chunk %.1f | groups of %u x %u lanes: events/chunk %.1f steps/chunk %.1f (speculated %.1f, used %.1f)\n",
	       strrchr(argv[1], '/') + 1, SELF, chunks, tok / chunks, unres_total / chunks, events / chunks, events / windows, iters / chunks, G, GL, events2 / chunks, iters2 / chunks, spec / chunks, spec_used / chunks);
	printf("   hash bits %d: candidates examined per chunk %.0f, of them hash collisions %.3f\n", HB, cands / chunks, coll / cands);
	printf("   candidates left at an event: <=4 %.2f  <=8 %.2f  <=16 %.2f  <=32 %.2f  <=64 %.2f  <=128 %.2f  <=256 %.2f  more %.2f\n",
	       remh[0] / events, remh[1] / events, remh[2] / events, remh[3] / events, remh[4] / events, remh[5] / events, remh[6] / events, remh[7] / events);
	return 0;
}

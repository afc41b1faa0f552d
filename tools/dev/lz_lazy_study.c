// lz_lazy_study.c -- CPU study (round 4): LZNT1 with per-lane CLAIMED WALKS (the Xpress lazy finder's scheme) instead of the eager 4-candidate
// scan + wave-cooperative finishing: 256 lanes per 4 KiB chunk, lane i starts at byte 16 i, claims a position, runs Find over its bucket's
// candidates (oldest first, stop at max_len), steps to the end of the token, claims that ... until it meets a claimed position. One wave step =
// up to FREP candidate compares per lane + DONE + NEW. Counts wave steps per chunk (4 waves), claims, compares, lane occupancy.
// usage: lz_lazy_study <file> [FREP]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static uint32_t shiftof(uint32_t pos) { return pos <= 16 ? 12 : 12 - ((32 - __builtin_clz(pos - 1)) - 4); }
int main(int argc, char** argv)
{
	FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t N = ftell(f); fseek(f, 0, SEEK_SET);
	uint8_t* d = malloc(N + 64); memset(d + N, 0, 64); if (fread(d, 1, N, f) != N) return 1; fclose(f);
	const int FREP = argc > 2 ? atoi(argv[2]) : 3;
	if (N > (4u << 20)) N = 4u << 20;
	double chunks = 0, wsteps = 0, claims = 0, compares = 0, lanesteps = 0, tokens_true = 0, maxw = 0;
	for (size_t cb = 0; cb + 4096 <= N; cb += 4096) {
		const uint8_t* c = d + cb; chunks++;
		static uint8_t claimed[4097]; memset(claimed, 0, sizeof claimed);
		// per position: candidates examined and token length (Find is a function of the position)
		static uint16_t ncand[4096], tlen[4096];
		for (uint32_t p = 0; p < 4096; ++p) {
			uint32_t best = 0, cnt = 0;
			if (p > 0 && p + 3 <= 4096) {
				uint32_t mask3 = (1u << shiftof(p)) + 2, maxlen = 4096 - p < mask3 ? 4096 - p : mask3;
				for (uint32_t q = 0; q < p; ++q) {
					if (c[q] != c[p] || c[q + 1] != c[p + 1] || c[q + 2] != c[p + 2]) continue;   // (exact keys: hash collisions add a little)
					cnt++;
					uint32_t l = 3; while (l < maxlen && c[q + l] == c[p + l]) l++;
					if (l > best) { best = l; if (best == maxlen) break; }
				}
			}
			ncand[p] = cnt; tlen[p] = best >= 3 ? best : 1;
		}
		for (uint32_t p = 0; p < 4096; p += tlen[p]) tokens_true++;
		// lockstep simulation: lane state = (position being worked on, steps left for it)
		int32_t pos[256], left[256]; uint32_t nactive = 256;
		for (int i = 0; i < 256; ++i) { pos[i] = 16 * i; left[i] = -1; }       // left = -1: has to claim pos first
		uint32_t steps[4] = {0, 0, 0, 0};
		while (nactive) {
			for (int w = 0; w < 4; ++w) {
				int any = 0;
				for (int l = 0; l < 64; ++l) { if (pos[w * 64 + l] >= 0) { any = 1; lanesteps++; } }
				if (any) steps[w]++;
			}
			// one step for every lane (waves in lockstep with each other: an approximation)
			for (int i = 0; i < 256; ++i) {
				if (pos[i] < 0) continue;
				if (left[i] < 0) {                                   // NEW: claim
					if (pos[i] >= 4096 || claimed[pos[i]]) { pos[i] = -1; nactive--; continue; }
					claimed[pos[i]] = 1; claims++;
					compares += ncand[pos[i]];
					left[i] = (ncand[pos[i]] + FREP - 1) / FREP;      // FIND steps (0: DONE + NEW of the next position happen in this very step)
					if (left[i] == 0) { pos[i] += tlen[pos[i]]; left[i] = -1; }
					continue;
				}
				if (--left[i] == 0) { pos[i] += tlen[pos[i]]; left[i] = -1; }   // last FIND step: DONE, and NEW is tried in the next step
			}
		}
		for (int w = 0; w < 4; ++w) { wsteps += steps[w]; if (steps[w] > maxw) maxw = steps[w]; }
	}
	printf("%-9s FREP %d: tokens/chunk %.0f claims/chunk %.0f (x%.2f) compares/chunk %.0f | wave steps per chunk (4 waves) %.0f = %.1f per wave, lane occupancy %.2f, worst wave %.0f\n",
	       argv[1] + 12, FREP, tokens_true / chunks, claims / chunks, claims / tokens_true, compares / chunks, wsteps / chunks, wsteps / chunks / 4, lanesteps / (wsteps * 64), maxw);
	return 0;
}

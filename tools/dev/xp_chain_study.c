// xp_chain_study.c -- CPU study of the work the Xpress-family match finder does per position on the bench corpus (dev tool, not shipped).
// usage: xp_chain_study <file> <mode: 0 = Xpress 64 KiB units / 8 KiB window, 1 = Xpress+Huffman whole file / 65535 window>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static uint32_t hash3(const uint8_t* d) { return (((d[0] & 0x1Fu) << 10) ^ ((uint32_t)d[1] << 5) ^ d[2]) & 0x7FFF; }
int main(int argc, char** argv)
{
	FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t N = ftell(f); fseek(f, 0, SEEK_SET);
	uint8_t* d = malloc(N + 64); memset(d + N, 0, 64); if (fread(d, 1, N, f) != N) return 1; fclose(f);
	int mode = atoi(argv[2]);
	const uint32_t maxoff = mode ? 0xFFFF : 0x2000;
	const size_t unit = mode ? N : 65536;
	int32_t* link = malloc(N * 4); int32_t* head = malloc(32768 * 4);
	uint8_t* blen = malloc(N); uint32_t* boff = malloc(N * 4); uint8_t* ncand = malloc(N);
	double pairs = 0, rounds = 0, improv = 0, nice = 0, witers = 0, pos = 0, has1 = 0, filt4 = 0, wit_first2 = 0, cand_hist[13] = {0};
	double compares_to_best = 0, lenh[6] = {0};
	for (size_t ub = 0; ub < N; ub += unit) {
		size_t n = N - ub < unit ? N - ub : unit; const uint8_t* u = d + ub;
		for (int i = 0; i < 32768; ++i) head[i] = -1;
		for (size_t p = 0; p + 2 < n; ++p) { uint32_t h = hash3(u + p); link[ub + p] = head[h]; head[h] = (int32_t)p; }
		for (size_t p = (n >= 2 ? n - 2 : 0); p < n; ++p) link[ub + p] = -1;
		for (size_t w = 0; w < n; w += 64) {
			uint32_t wmax = 0;
			for (size_t p = w; p < w + 64 && p < n; ++p) {
				uint32_t best = 2, bo = 0, cnt = 0; pos++;
				if (p + 2 < n) {
					uint32_t cap = n - p - 1 < 48 ? n - p - 1 : 48;
					if (mode) { uint32_t inch = 65536 - (p & 65535); if (inch < 3) { cap = 0; } }
					int32_t x = link[ub + p]; uint32_t chain = 11;
					while (cap && chain && x >= 0 && p - x <= maxoff) {
						cnt++; pairs++;
						// filter: 4 bytes ending at index best (or bytes 0..3 when best < 3)
						uint32_t fo = best >= 3 ? best - 3 : 0;
						if (memcmp(u + x + fo, u + p + fo, 4) == 0) filt4++;
						uint32_t l = 0; while (l < cap && u[x + l] == u[p + l]) l++;
						rounds += 1 + (l >= 16 && cap > 16) + (l >= 32 && cap > 32);
						lenh[l < 3 ? 0 : l < 8 ? 1 : l < 16 ? 2 : l < 32 ? 3 : l < 48 ? 4 : 5]++;
						if (l > best) { best = l; bo = p - x; improv++; compares_to_best = compares_to_best; if (best >= 48) { nice++; break; } }
						x = link[ub + x]; chain--;
					}
				}
				if (cnt) has1++;
				cand_hist[cnt]++;
				ncand[ub + p] = cnt; blen[ub + p] = best; boff[ub + p] = bo;
				if (cnt > wmax) wmax = cnt;
			}
			witers += wmax;
		}
	}
	// greedy parse (no lagging-fill subtleties): token starts, and windows of 64 positions with no token start
	double tok = 0, lit = 0, emptyw = 0, tokpairs = 0, wcnt = 0;
	for (size_t ub = 0; ub < N; ub += unit) {
		size_t n = N - ub < unit ? N - ub : unit;
		uint8_t* isstart = calloc(n + 1, 1);
		for (size_t p = 0; p < n;) {
			isstart[p] = 1; tok++; tokpairs += ncand[ub + p];
			if (blen[ub + p] >= 3) {
				uint32_t l = blen[ub + p];
				if (l >= 48) { const uint8_t* a = d + ub + p - boff[ub + p]; const uint8_t* b = d + ub + p; size_t lim = n - p - 1; if (mode) { size_t inch = 65536 - (p & 65535); if (inch < lim + 1) lim = inch; } while (l < lim && a[l] == b[l]) l++; }
				p += l;
			} else { lit++; p++; }
		}
		for (size_t w = 0; w < n; w += 64) { int any = 0; for (size_t p = w; p < w + 64 && p < n; ++p) any |= isstart[p]; wcnt++; if (!any) emptyw++; }
		free(isstart);
	}
	printf("%-9s mode %d: pos %.0f pairs/pos %.2f (has>=1: %.2f) rounds/pair %.2f improv/pos %.2f nice/pos %.3f filt4pass/pair %.2f lane_eff %.2f witers/wave %.2f | tok/pos %.3f lit/tok %.2f pairs/tok %.2f emptywin %.3f\n",
	       strrchr(argv[1], '/') + 1, mode, pos, pairs / pos, has1 / pos, rounds / pairs, improv / pos, nice / pos, filt4 / pairs, pairs / (64 * witers), witers / (pos / 64), tok / pos, lit / tok, tokpairs / tok, emptyw / wcnt);
	printf("   pair length shares: <3 %.3f  3-7 %.3f  8-15 %.3f  16-31 %.3f  32-47 %.3f  48 %.3f\n", lenh[0] / pairs, lenh[1] / pairs, lenh[2] / pairs, lenh[3] / pairs, lenh[4] / pairs, lenh[5] / pairs);
	printf("   cand hist:"); for (int i = 0; i <= 11; ++i) printf(" %d:%.3f", i, cand_hist[i] / pos); printf("\n");
	return 0;
}

// LZNT1: how much of the candidate work is hash collisions (11-bit hash of the 3-byte key) -- per visited token start of the greedy parse
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static uint32_t lz_hash(uint32_t k, int bits) { return (k * 0x9E3779B1u) >> (32 - bits); }
static uint32_t shiftof(uint32_t pos) { return pos <= 16 ? 12 : 12 - ((32 - __builtin_clz(pos - 1)) - 4); }
int main(int argc, char** argv)
{
	FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t N = ftell(f); fseek(f, 0, SEEK_SET);
	uint8_t* d = malloc(N + 64); memset(d + N, 0, 64); if (fread(d, 1, N, f) != N) return 1; fclose(f);
	if (N > (4u << 20)) N = 4u << 20;
	for (int bits = 11; bits <= 13; ++bits) {
		double tokens = 0, cand_hash = 0, cand_true = 0, fin_hash = 0, fin_true = 0, finpos_hash = 0, finpos_true = 0, chunks = 0, eager_coll = 0;
		for (size_t cb = 0; cb + 4096 <= N; cb += 4096) {
			const uint8_t* c = d + cb; chunks++;
			for (uint32_t p = 0; p < 4096;) {
				uint32_t best = 0; tokens++;
				if (p > 0 && p + 3 <= 4096) {
					uint32_t mask3 = (1u << shiftof(p)) + 2, maxlen = 4096 - p < mask3 ? 4096 - p : mask3;
					uint32_t key = c[p] | c[p + 1] << 8 | c[p + 2] << 16, h = lz_hash(key, bits);
					uint32_t nh = 0, nt = 0; int done_h = 0, done_t = 0;
					for (uint32_t q = 0; q < p; ++q) {
						uint32_t kq = c[q] | c[q + 1] << 8 | c[q + 2] << 16;
						if (lz_hash(kq, bits) != h) continue;
						// hash-bucket scan order: oldest first; stops when maxlen reached
						uint32_t l = 0; if (kq == key) { while (l < maxlen && c[q + l] == c[p + l]) l++; }
						if (!done_h) { nh++; if (nh <= 4 && kq != key) eager_coll++; }
						if (kq == key && !done_t) nt++;
						if (l > best) { best = l; if (best == maxlen) { done_h = 1; done_t = 1; } }
					}
					cand_hash += nh; cand_true += nt;
					if (nh > 4) { finpos_hash++; fin_hash += (nh - 4 + 63) / 64; }
					if (nt > 4) { finpos_true++; fin_true += (nt - 4 + 63) / 64; }
				}
				p += best >= 3 ? best : 1;
			}
		}
		printf("%-9s bits %d: tokens/chunk %.0f | cand/token hash %.2f exact %.2f | eager slots wasted on collisions/token %.2f | finished positions/chunk hash %.0f exact %.0f | finish steps/chunk hash %.0f exact %.0f\n",
		       argv[1] + 12, bits, tokens / chunks, cand_hash / tokens, cand_true / tokens, eager_coll / tokens, finpos_hash / chunks, finpos_true / chunks, fin_hash / chunks, fin_true / chunks);
	}
	return 0;
}

// study: XH greedy parse (whole file, 64K window approx): candidates examined at token starts by distance class
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static uint32_t hash3(const uint8_t* d) { return (((d[0] & 0x1Fu) << 10) ^ ((uint32_t)d[1] << 5) ^ d[2]) & 0x7FFF; }
int main(int argc, char** argv)
{
	FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t N = ftell(f); fseek(f, 0, SEEK_SET);
	uint8_t* d = malloc(N + 64); memset(d + N, 0, 64); if (fread(d, 1, N, f) != N) return 1; fclose(f);
	int32_t* link = malloc(N * 4); int32_t* head = malloc(32768 * 4);
	for (int i = 0; i < 32768; ++i) head[i] = -1;
	for (size_t p = 0; p + 2 < N; ++p) { uint32_t h = hash3(d + p); link[p] = head[h]; head[h] = (int32_t)p; }
	for (size_t p = (N >= 2 ? N - 2 : 0); p < N; ++p) link[p] = -1;
	double cls[6] = {0}, tok = 0, cand = 0, tokfar[6] = {0};
	const uint32_t lim[5] = {4096, 8192, 16384, 24576, 32768};
	for (size_t p = 0; p < N;) {
		uint32_t best = 2; tok++;
		uint32_t far = 0;
		if (p + 2 < N) {
			uint32_t cap = N - p - 1; uint32_t inch = 65536 - (p & 65535); if (inch < 3) cap = 0;
			int32_t x = link[p]; uint32_t chain = 11;
			while (cap && chain && x >= 0 && p - x <= 0xFFFF) {
				cand++;
				uint32_t dist = p - x; int c = 0; while (c < 5 && dist > lim[c]) c++;
				cls[c]++; if ((uint32_t)c > far) far = c;
				uint32_t l = 0; while (l < cap && d[x + l] == d[p + l]) l++;
				if (l > best) { best = l; if (best >= 48) break; }
				x = link[x]; chain--;
			}
		}
		tokfar[far]++;
		uint32_t inch = 65536 - (p & 65535);
		if (best >= 3) { if (best > inch) best = inch; p += best; } else p++;
	}
	printf("%-9s tokens/B %.3f cand/token %.2f | cand by distance <=4K %.3f <=8K %.3f <=16K %.3f <=24K %.3f <=32K %.3f >32K %.3f | tokens whose farthest cand: ", argv[1] + 12, tok / N, cand / tok,
	       cls[0] / cand, cls[1] / cand, cls[2] / cand, cls[3] / cand, cls[4] / cand, cls[5] / cand);
	for (int i = 0; i < 6; ++i) printf("%.3f ", tokfar[i] / tok);
	printf("\n");
	return 0;
}

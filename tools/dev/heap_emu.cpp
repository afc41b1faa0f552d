// CPU emulation of the wave-wide heap operations with kept choice bits (xhuff.hip hh_push / hh_pop, round 5) against the plain heap
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef uint64_t u64;
struct E { uint32_t x, y; };
static const uint32_t SENT = 0xFFFFFFFFu;
struct Choice { u64 m[4]; };
static E heap[520];
static void set_any(Choice& c, uint32_t idx, bool bit) { u64 one = 1ull << (idx & 63); if (bit) c.m[idx >> 6] |= one; else c.m[idx >> 6] &= ~one; }
static void set_level(Choice& c, int L, uint32_t idx, bool bit) {
	int q = L <= 5 ? 0 : L == 6 ? 1 : (idx < 192 ? 2 : 3);
	if ((int)(idx >> 6) != q) { printf("LEVEL/WORD MISMATCH L=%d idx=%u\n", L, idx); exit(1); }
	set_any(c, idx, bit);
}
static void push(Choice& c, uint32_t hl_new, E e) {
	E par[64], sib[64]; uint32_t pc[64]; bool anc[64];
	u64 upb = 0;
	for (uint32_t lane = 0; lane < 64; ++lane) {
		uint32_t a = lane < 32 ? hl_new >> lane : 0;
		anc[lane] = lane >= 1 && lane <= 9;
		par[lane] = E{0, 0}; sib[lane] = E{SENT, 0};
		pc[lane] = anc[lane] ? hl_new >> (lane - 1) : 0;
		if (anc[lane]) { par[lane] = heap[a]; sib[lane] = heap[pc[lane] ^ 1]; }
		if (anc[lane] && e.x < par[lane].x) upb |= 1ull << lane;
	}
	u64 up = upb >> 1;
	uint32_t m = __builtin_ctzll(~up);
	for (uint32_t lane = 1; lane <= m && lane < 64; ++lane) heap[hl_new >> (lane - 1)] = par[lane];
	heap[hl_new >> m] = e;
	u64 B = 0;
	for (uint32_t lane = 0; lane < 64; ++lane) {
		uint32_t nw = lane <= m ? par[lane].x : e.x;
		bool b = (pc[lane] & 1) ? (nw < sib[lane].x) : (sib[lane].x < nw);
		if (b) B |= 1ull << lane;
	}
	int lj = 31 - __builtin_clz(hl_new);
	for (int L = 0; L <= 7; ++L) { int l = lj - L; if (l >= 1 && l <= (int)m + 1) { set_level(c, L, hl_new >> l, (B >> l) & 1); } }
}
static E pop(Choice& c, uint32_t hl_old) {
	E top = heap[1], t = heap[hl_old];
	heap[hl_old] = E{SENT, 0};
	if (hl_old >= 2 && hl_old < 512) set_any(c, hl_old >> 1, false);
	uint32_t i = 1;
	for (int s = 0; s < 6; ++s) i = 2 * i + ((c.m[0] >> i) & 1);
	i = 2 * i + ((c.m[1] >> (i - 64)) & 1);
	{ u64 mk = i < 192 ? c.m[2] : c.m[3]; i = 2 * i + ((mk >> (i & 63)) & 1); }
	uint32_t F = i;
	E k[64]; uint32_t sibw[64], gx[64]; bool right[64]; u64 stop = 0;
	for (uint32_t lane = 0; lane < 64; ++lane) {
		bool on = lane >= 1 && lane <= 8;
		uint32_t al = on ? F >> (8 - lane) : 2;
		E l = E{SENT, 0}, r = E{SENT, 0}; gx[lane] = SENT;
		if (on) { l = heap[al & ~1u]; r = heap[(al & ~1u) + 1]; }
		if (on && lane <= 7) gx[lane] = heap[F >> (7 - lane)].x;
		right[lane] = al & 1;
		k[lane] = right[lane] ? r : l; sibw[lane] = right[lane] ? l.x : r.x;
		if (on && t.x < k[lane].x) stop |= 1ull << lane;
	}
	uint32_t m = stop ? __builtin_ctzll(stop) : 9;
	for (uint32_t lane = 1; lane < m; ++lane) heap[F >> (9 - lane)] = k[lane];
	heap[F >> (9 - m)] = t;
	u64 B = 0;
	for (uint32_t lane = 0; lane < 64; ++lane) { uint32_t nw = (lane + 2 <= m) ? gx[lane] : t.x; bool b = right[lane] ? (nw < sibw[lane]) : (sibw[lane] < nw); if (b) B |= 1ull << lane; }
	for (int L = 0; L <= 7; ++L) { if ((uint32_t)L + 2 <= m) set_level(c, L, F >> (8 - L), (B >> (L + 1)) & 1); }
	return top;
}
// plain reference (HuffmanEncoder.h:31-55) on node ids with a weight table
static uint32_t W[1100]; static uint32_t rh[600]; static uint32_t rlen;
static void rpush(uint32_t x) { rh[++rlen] = x; uint32_t j = rlen; while (W[x] < W[rh[j >> 1]]) { rh[j] = rh[j >> 1]; j >>= 1; } rh[j] = x; }
static uint32_t rpop() { uint32_t top = rh[1]; uint32_t i = 1, t = rh[1] = rh[rlen--]; for (;;) { uint32_t j = i << 1; if (j > rlen) break; if (j < rlen && W[rh[j + 1]] < W[rh[j]]) ++j; if (W[t] < W[rh[j]]) break; rh[i] = rh[j]; i = j; } rh[i] = t; return top; }
static void check(const Choice& c, uint32_t hl, const char* where, int step) {
	for (uint32_t i = 1; i <= hl; ++i) { if (heap[i].y != rh[i] || heap[i].x != W[rh[i]]) { printf("heap mismatch %s step %d slot %u\n", where, step, i); exit(1); } }
	for (uint32_t i = hl + 1; i < 516; ++i) { if (i == 1 && hl == 0) continue; if (heap[i].x != SENT) { printf("not SENT beyond the heap %s step %d slot %u\n", where, step, i); exit(1); } }
	for (uint32_t i = 1; i <= 255; ++i) {
		bool want = heap[2 * i + 1].x < heap[2 * i].x;
		if (hl == 0 && i == 0) continue;
		bool have = (c.m[i >> 6] >> (i & 63)) & 1;
		if (want != have) { printf("choice mismatch %s step %d node %u want %d\n", where, step, i, want); exit(1); }
	}
}
int main() {
	srand(12345);
	for (int trial = 0; trial < 400; ++trial) {
		int kind = trial % 5;
		for (int i = 0; i < 516; ++i) heap[i] = i ? E{SENT, 0} : E{0, 0};
		Choice c; memset(&c, 0, sizeof c);
		W[0] = 0; rh[0] = 0; rlen = 0;
		for (uint32_t s = 1; s <= 512; ++s) {
			uint32_t cnt = kind == 0 ? 1 : kind == 1 ? (rand() % 4) : kind == 2 ? (rand() % 60000) : kind == 3 ? ((rand() % 16 == 0) ? rand() % 50000 : 0) : (1u << (rand() % 16));
			W[s] = (cnt ? cnt : 1) << 8;
		}
		for (uint32_t s = 1; s <= 512; ++s) { push(c, s, E{W[s], s}); rpush(s); check(c, s, "build", s); }
		uint32_t hl = 512, nn = 512; int step = 0;
		while (hl > 1) {
			E ea = pop(c, hl); --hl; uint32_t ra = rpop(); if (ea.y != ra) { printf("pop a mismatch\n"); return 1; } check(c, hl, "popA", step);
			E eb = pop(c, hl); --hl; uint32_t rb = rpop(); if (eb.y != rb) { printf("pop b mismatch\n"); return 1; } if (hl) check(c, hl, "popB", step);
			++nn; uint32_t da = ea.x & 0xFF, db = eb.x & 0xFF;
			W[nn] = ((ea.x & ~0xFFu) + (eb.x & ~0xFFu)) | (1 + (da > db ? da : db));
			++hl; push(c, hl, E{W[nn], nn}); rpush(nn); check(c, hl, "push", step);
			++step;
		}
	}
	printf("heap emulation OK\n");
	return 0;
}

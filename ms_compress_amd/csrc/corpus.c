/* corpus.c -- deterministic "Silesia-shaped" synthetic corpus.
 *
 * The real Silesia corpus is not available offline, so bench.py / the tests feed the codecs 12 pseudo-files
 * with exactly the Silesia sizes, produced by an integer-only generator (splitmix64, seed 0x5113514 +
 * file index). Classes: text (Zipf-ranked synthetic word list), markup, fixed-width records (nci-like),
 * tar-of-mixed (512 B aligned members: text, opcode soup, zero padding, one >64 KiB zero run and one
 * 128 KiB incompressible member so the Xpress lagging-fill rule and the XH fallback are exercised),
 * binary records and 16-bit random-walk images. Bench input only: nothing here is on the codec path.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rnd(rng_t* r)
{
	uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static inline uint32_t below(rng_t* r, uint32_t n) { return (uint32_t)(((rnd(r) >> 32) * (uint64_t)n) >> 32); }

/* ---- shared word list (4096 words, Zipf ranked) ---- */
#define NWORDS 4096
static char     g_words[NWORDS][14];
static uint8_t  g_wlen[NWORDS];
static uint32_t g_cdf[NWORDS];
static int      g_init = 0;
static const char LETTERS[] = "eeeeeeeeeeeetttttttttaaaaaaaaooooooooiiiiiiinnnnnnnsssssshhhhhhrrrrrrddddllllcccuuummmwwfffggyyppbbvk";
static void init_words(void)
{
	if (g_init) { return; }
	rng_t r = { 0x5113514ull ^ 0xABCDEF };
	uint64_t acc = 0;
	for (int i = 0; i < NWORDS; ++i) {
		int len = 2 + (int)below(&r, 4) + (i > 64 ? (int)below(&r, 4) : 0) + (i > 1024 ? (int)below(&r, 4) : 0);
		for (int k = 0; k < len; ++k) { g_words[i][k] = LETTERS[below(&r, sizeof(LETTERS) - 1)]; }
		g_wlen[i] = (uint8_t)len;
		acc += 0x40000000ull / (uint64_t)(i + 2);
		g_cdf[i] = (uint32_t)(acc >> 4);
	}
	g_init = 1;
}
static int zipf_word(rng_t* r)
{
	const uint32_t x = below(r, g_cdf[NWORDS - 1]);
	int lo = 0, hi = NWORDS - 1;
	while (lo < hi) { const int mid = (lo + hi) >> 1; if (g_cdf[mid] > x) { hi = mid; } else { lo = mid + 1; } }
	return lo;
}

typedef struct { uint8_t* p; size_t n, w; } sink_t;
static inline void put(sink_t* s, uint8_t b) { if (s->w < s->n) { s->p[s->w] = b; } ++s->w; }
static inline void puts_n(sink_t* s, const char* t, size_t k) { for (size_t i = 0; i < k; ++i) { put(s, (uint8_t)t[i]); } }
static inline void puts_z(sink_t* s, const char* t) { puts_n(s, t, strlen(t)); }
static void put_num(sink_t* s, uint32_t v, int width)
{
	char b[12]; int k = 0;
	do { b[k++] = (char)('0' + v % 10); v /= 10; } while (v && k < 11);
	while (k < width) { b[k++] = '0'; }
	while (k) { put(s, (uint8_t)b[--k]); }
}

static void gen_text(rng_t* r, sink_t* s, size_t upto, int line_words)
{
	int col = 0, start = 1;
	while (s->w < upto) {
		const int w = zipf_word(r);
		if (start) { put(s, (uint8_t)(g_words[w][0] - 32)); puts_n(s, g_words[w] + 1, g_wlen[w] - 1u); start = 0; }
		else { puts_n(s, g_words[w], g_wlen[w]); }
		const uint32_t e = below(r, 64);
		if (e == 0) { puts_z(s, ". "); start = 1; }
		else if (e < 4) { puts_z(s, ", "); }
		else { put(s, ' '); }
		if (++col >= line_words + (int)below(r, 3)) { put(s, '\n'); col = 0; if (below(r, 12) == 0) { put(s, '\n'); } }
	}
}

static void gen_markup(rng_t* r, sink_t* s, size_t upto)
{
	static const char* tags[8] = { "entry", "name", "value", "item", "section", "title", "ref", "note" };
	uint32_t id = 1000;
	while (s->w < upto) {
		const int depth = 1 + (int)below(r, 3);
		for (int d = 0; d < depth; ++d) { puts_z(s, "  "); }
		const char* t = tags[below(r, 8)];
		put(s, '<'); puts_z(s, t); puts_z(s, " id=\""); put_num(s, id++, 6); puts_z(s, "\" type=\""); puts_z(s, tags[below(r, 4)]); puts_z(s, "\">");
		const int nw = 1 + (int)below(r, 6);
		for (int k = 0; k < nw; ++k) { const int w = zipf_word(r); puts_n(s, g_words[w], g_wlen[w]); if (k + 1 < nw) { put(s, ' '); } }
		puts_z(s, "</"); puts_z(s, t); puts_z(s, ">\n");
	}
}

static void gen_fixed_records(rng_t* r, sink_t* s, size_t upto)   /* nci-like: very repetitive fixed-width lines */
{
	uint32_t atom = 1;
	while (s->w < upto) {
		const uint32_t natoms = 8 + below(r, 24);
		puts_z(s, "  -ISIS-  "); put_num(s, 10000000u + below(r, 900000u), 8); puts_z(s, "2D\n\n");
		put_num(s, natoms, 3); put(s, ' '); put_num(s, natoms + below(r, 3), 3); puts_z(s, "  0  0  0  0  0  0  0  0999 V2000\n");
		for (uint32_t a = 0; a < natoms; ++a) {
			puts_z(s, "   "); put_num(s, below(r, 10), 1); put(s, '.'); put_num(s, below(r, 10000) / 25 * 25, 4);
			puts_z(s, "   -"); put_num(s, below(r, 10), 1); put(s, '.'); put_num(s, below(r, 10000) / 25 * 25, 4);
			puts_z(s, "    0.0000 "); put(s, (uint8_t)"CCCCCCNOOHS"[below(r, 11)]); puts_z(s, "   0  0  0  0  0  0  0  0  0  0  0  0\n");
		}
		for (uint32_t a = 1; a < natoms; ++a) { put_num(s, a, 3); put(s, ' '); put_num(s, a + 1, 3); puts_z(s, "  1  0  0  0  0\n"); }
		puts_z(s, "M  END\n>  <NSC>\n"); put_num(s, atom++, 6); puts_z(s, "\n\n$$$$\n");
	}
}

static const uint8_t OPS[64] = { 0x8B, 0x89, 0x48, 0x48, 0xE8, 0xFF, 0x83, 0x0F, 0x85, 0x74, 0x75, 0xEB, 0x24, 0x44, 0x4C, 0x8D,
	0x00, 0x00, 0x00, 0x01, 0x45, 0x5D, 0xC3, 0x55, 0x53, 0x41, 0x5C, 0x10, 0x08, 0x04, 0x20, 0x40,
	0xC7, 0x05, 0x31, 0xC0, 0x84, 0x3B, 0x39, 0x7C, 0x7E, 0x90, 0xCC, 0x66, 0xF3, 0xE9, 0x50, 0x58,
	0x8B, 0x45, 0xFC, 0x89, 0x4D, 0xF8, 0x00, 0x00, 0xFF, 0xFF, 0x15, 0x25, 0x0C, 0x18, 0x28, 0x30 };
static void gen_opcodes(rng_t* r, sink_t* s, size_t upto)
{
	const size_t base = s->w;
	while (s->w < upto) {
		const uint32_t e = below(r, 16);
		if (e < 3 && s->w - base > 64) {                 /* re-use of an earlier instruction sequence */
			const size_t back = 4 + below(r, (uint32_t)((s->w - base < 6000 ? s->w - base : 6000) - 4));
			const size_t len = 3 + below(r, 14);
			for (size_t k = 0; k < len; ++k) { const size_t src = s->w - back; put(s, src < s->n ? s->p[src] : 0); }
		} else if (e < 5) { const uint32_t v = below(r, 4096) * 4; put(s, (uint8_t)v); put(s, (uint8_t)(v >> 8)); put(s, 0); put(s, 0); }
		else if (e < 12) { put(s, OPS[below(r, 64)]); put(s, OPS[below(r, 64)]); }
		else { put(s, OPS[below(r, 64)]); put(s, (uint8_t)rnd(r)); }
	}
}

static void gen_zero(sink_t* s, size_t upto) { while (s->w < upto) { put(s, 0); } }
static void gen_random(rng_t* r, sink_t* s, size_t upto) { while (s->w < upto) { put(s, (uint8_t)(rnd(r) >> 24)); } }

static void gen_bin_records(rng_t* r, sink_t* s, size_t upto, uint32_t rec)
{
	uint32_t ctr = 1; uint32_t f[8] = { 100, 2000, 30000, 7, 0, 0, 0, 0 };
	while (s->w < upto) {
		const size_t start = s->w;
		for (int k = 0; k < 4; ++k) { put(s, (uint8_t)(ctr >> (8 * k))); }
		++ctr;
		for (int i = 0; i < 4; ++i) {
			if (below(r, 8) == 0) { f[i] += below(r, 5) - 2; }
			for (int k = 0; k < 4; ++k) { put(s, (uint8_t)(f[i] >> (8 * k))); }
		}
		for (int i = 0; i < 12; ++i) { put(s, (uint8_t)(rnd(r) >> (below(r, 4) ? 60 : 56))); }
		const int w = zipf_word(r) & 255; puts_n(s, g_words[w], g_wlen[w]);
		while (s->w - start < rec) { put(s, (s->w - start) & 16 ? 0x20 : 0); }
	}
}

static void gen_image16(rng_t* r, sink_t* s, size_t upto, int sigma)
{
	int32_t v = 2048;
	while (s->w < upto) {
		int32_t d = 0;
		for (int k = 0; k < 4; ++k) { d += (int32_t)below(r, (uint32_t)(2 * sigma + 1)) - sigma; }
		v += d / 2;
		if (v < 0) { v = 0; }
		if (v > 4095) { v = 4095; }
		put(s, (uint8_t)v); put(s, (uint8_t)(v >> 8));
	}
}

static void gen_mixed(rng_t* r, sink_t* s, size_t upto, int special)
{
	int member = 0;
	while (s->w < upto) {
		const size_t hdr_end = s->w + 512;
		const int w = zipf_word(r); puts_z(s, "pkg/"); puts_n(s, g_words[w], g_wlen[w]); puts_z(s, ".bin");
		gen_zero(s, hdr_end - 388); puts_z(s, "0000644 0001750 0001750 "); put_num(s, (uint32_t)member, 11); gen_zero(s, hdr_end);
		size_t len = 1024 + below(r, 180 * 1024);
		uint32_t kind = below(r, 16);
		if (special && member == 3) { kind = 100; len = 100 * 1024; }      /* zero run > 64 KiB */
		if (special && member == 5) { kind = 101; len = 128 * 1024; }      /* incompressible 128 KiB */
		size_t end = s->w + len;
		if (end > upto) { end = upto; }
		if (kind == 100) { gen_zero(s, end); }
		else if (kind == 101) { gen_random(r, s, end); }
		else if (kind < 7) { gen_opcodes(r, s, end); }
		else if (kind < 11) { gen_text(r, s, end, 10); }
		else if (kind < 13) { gen_bin_records(r, s, end, 64 + 32 * below(r, 6)); }
		else if (kind < 15) { const size_t z = s->w + 512 + below(r, 8192); gen_zero(s, z < end ? z : end); gen_markup(r, s, end); }
		else { gen_image16(r, s, end, 12); }
		gen_zero(s, (s->w + 511) / 512 * 512 < upto ? (s->w + 511) / 512 * 512 : upto);
		++member;
	}
}

/* The 12 Silesia members, in the corpus' usual order. */
static const struct { const char* name; uint64_t size; int cls; } FILES[12] = {
	{ "dickens", 10192446, 0 }, { "mozilla", 51220480, 3 }, { "mr", 9970564, 5 }, { "nci", 33553445, 2 },
	{ "ooffice", 6152192, 3 }, { "osdb", 10085684, 4 }, { "reymont", 6627202, 0 }, { "samba", 21606400, 3 },
	{ "sao", 7251944, 4 }, { "webster", 41458703, 0 }, { "xml", 5345280, 1 }, { "x-ray", 8474240, 5 },
};

int         mscorpus_num_files(void)       { return 12; }
const char* mscorpus_file_name(int i)      { return (i >= 0 && i < 12) ? FILES[i].name : ""; }
uint64_t    mscorpus_file_size(int i)      { return (i >= 0 && i < 12) ? FILES[i].size : 0; }

/* Generate the first n bytes of pseudo-file i (the stream is prefix-stable: a shorter n yields a prefix). */
void mscorpus_generate(int i, uint8_t* out, uint64_t n)
{
	if (i < 0 || i >= 12) { return; }
	init_words();
	rng_t r = { 0x5113514ull + (uint64_t)i };
	sink_t s = { out, (size_t)n, 0 };
	switch (FILES[i].cls) {
	case 0: gen_text(&r, &s, n, i == 9 ? 7 : 12); break;
	case 1: gen_markup(&r, &s, n); break;
	case 2: gen_fixed_records(&r, &s, n); break;
	case 3: gen_mixed(&r, &s, n, 1); break;
	case 4: gen_bin_records(&r, &s, n, i == 5 ? 96 : 160); break;
	default: gen_image16(&r, &s, n, i == 2 ? 6 : 14); break;
	}
}

/* oracle/mscomp_oracle.c -- TEST INFRASTRUCTURE ONLY (see mscomp_oracle.h for the rules).
 *
 * Plain-C restatement of the reference's one-shot encoders, written from the validated pseudocode in
 * SURVEY.md section 8a. Every function cites the reference file:line whose behaviour it restates.
 * Data structures are our own (ascending per-key lists for LZNT1, int32 predecessor links for the
 * Xpress hash chains, an explicit token array for Xpress-Huffman) -- only the observable bytes match.
 *
 * Parity status: PINNED against the compiled reference (oracle/_ref) and the SURVEY 8c KAT table;
 * see tests/test_oracle_vs_ref.py and tests/test_oracle_golden.py. The decoders (with the reference's status codes) are
 * pinned the same way: test_decompress_semantics and tests/golden/decode_streams.json.
 */
#include "mscomp_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

static inline void put16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static inline void put32(uint8_t* p, uint32_t v) { put16(p, v); put16(p + 2, v >> 16); }
static inline uint32_t get16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static inline uint32_t get32(const uint8_t* p) { return get16(p) | (get16(p + 2) << 16); }
static inline unsigned ilog2(uint32_t v) { return 31u - (unsigned)__builtin_clz(v); }

/* ===================================================================================================
 * sizes  (lznt1_compress.cpp:27, xpress_compress.cpp:30, xpress_huff_compress.cpp:46, mscomp.cpp:33,96-100)
 * =================================================================================================*/
size_t orc_max_compressed_size(int format, size_t n)
{
	switch (format) {
	case ORC_NONE:        return n;
	case ORC_LZNT1:       return n + 3 + 2 * ((n + 4095) / 4096);
	case ORC_XPRESS:      return n + 4 + 4 * (n / 32);
	case ORC_XPRESS_HUFF: return n + 34 + 258 + 258 * (n / 65536);
	default:              return (size_t)-1;
	}
}

/* ===================================================================================================
 * LZNT1
 * =================================================================================================*/
#define LZ_NONE 0xFFFFu
typedef struct {
	uint32_t stamp[65536];   /* generation in which first[] was last written (avoids the reference's */
	uint16_t first[65536];   /*  128 KiB memset per chunk, LZNT1Dictionary.h:98)                       */
	uint16_t next[4096];     /* next position (ascending) with the same 2-byte key                    */
	uint32_t gen;
} lznt1_dict;

/* LZNT1Dictionary::Fill (LZNT1Dictionary.h:93-106): every p with p+2<n, keyed by (c[p],c[p+1]),
 * lists in ascending position order. */
static void lznt1_fill(lznt1_dict* d, const uint8_t* c, unsigned n)
{
	if (++d->gen == 0) { memset(d->stamp, 0, sizeof d->stamp); d->gen = 1; }
	for (int p = (int)n - 3; p >= 0; --p) {
		const unsigned k = ((unsigned)c[p] << 8) | c[p + 1];
		d->next[p] = (d->stamp[k] == d->gen) ? d->first[k] : LZ_NONE;
		d->first[k] = (uint16_t)p; d->stamp[k] = d->gen;
	}
}

/* LZNT1Dictionary::Find (LZNT1Dictionary.h:114-143): ascending scan of the earlier positions with the
 * same 2-byte key; third byte must match; strictly-longer wins (oldest on ties); stop at max_len. */
static unsigned lznt1_find(const lznt1_dict* d, const uint8_t* c, unsigned p, unsigned max_len, unsigned* off)
{
	unsigned best = 0;
	if (max_len < 3 || p == 0) { return 0; }
	const unsigned k = ((unsigned)c[p] << 8) | c[p + 1];
	for (unsigned q = d->first[k]; q < p; q = d->next[q]) {
		if (c[q + 2] != c[p + 2]) { continue; }
		unsigned l = 3;
		while (l < max_len && c[q + l] == c[p + l]) { ++l; }
		if (l > best) { best = l; *off = p - q; if (best == max_len) { break; } }
	}
	return best;
}


/* ---- LZNT1, suffix-array dictionary flavour (the reference built with MSCOMP_WITH_LZNT1_SA_DICT: include/mscomp/LZNT1Dictionary_SA.h) ----
 * Same chunk loop, another Find: the longest match is taken from the two LEXICOGRAPHIC neighbours of the position's suffix that start
 * earlier -- the nearest one before it in suffix-array order, the nearest one after it -- the one before winning ties (:424-476), so the
 * length is the same "optimal" one but the offset differs from the default dictionary (which takes the OLDEST candidate of that length).
 * Restated with a plain construction (prefix doubling with qsort, brute-force LCP): the suffix array of a string is unique -- a suffix
 * that is a prefix of another sorts first (:360, the theoretical terminator) -- so SA-IS (:49-354) and this give the same arrays. */
static int orc_lznt1_sa = 0;
void orc_set_lznt1_sa_dict(int on) { orc_lznt1_sa = on; }
typedef struct { int16_t sa[4096], inv[4096], lcp[4096]; unsigned n; } lznt1_sa_dict;
static const uint32_t* sa_key_for_sort;
static int sa_cmp(const void* a, const void* b)
{
	const uint32_t ka = sa_key_for_sort[*(const int16_t*)a], kb = sa_key_for_sort[*(const int16_t*)b];
	return ka < kb ? -1 : ka > kb;
}
static void lznt1_sa_fill(lznt1_sa_dict* d, const uint8_t* c, unsigned n)        /* Fill (:404-419) */
{
	static __thread uint32_t key[4096];
	static __thread uint16_t rank[4096], nr[4096];
	d->n = n;
	if (n <= 3) { return; }
	for (unsigned i = 0; i < n; ++i) { rank[i] = (uint16_t)(c[i] + 1u); d->sa[i] = (int16_t)i; }
	for (unsigned k = 1;; k <<= 1) {
		for (unsigned i = 0; i < n; ++i) { key[i] = ((uint32_t)rank[i] << 13) | (i + k < n ? rank[i + k] : 0u); }
		sa_key_for_sort = key;
		qsort(d->sa, n, sizeof(int16_t), sa_cmp);
		unsigned r = 1; nr[d->sa[0]] = 1;
		for (unsigned i = 1; i < n; ++i) { if (key[d->sa[i]] != key[d->sa[i - 1]]) { ++r; } nr[d->sa[i]] = (uint16_t)r; }
		memcpy(rank, nr, n * sizeof(uint16_t));
		if (r == n || k >= n) { break; }
	}
	for (unsigned i = 0; i < n; ++i) { d->inv[d->sa[i]] = (int16_t)i; }
	d->lcp[0] = 0;
	for (unsigned i = 1; i < n; ++i) {                                               /* calc_lcp (:361-384) gives exactly these values */
		const unsigned a = (unsigned)d->sa[i - 1], b = (unsigned)d->sa[i], m = n - (a > b ? a : b);
		unsigned l = 0;
		while (l < m && c[a + l] == c[b + l]) { ++l; }
		d->lcp[i] = (int16_t)l;
	}
}
static unsigned lznt1_sa_find(const lznt1_sa_dict* d, unsigned pos, unsigned max_len, unsigned* off)   /* Find (:424-476) */
{
	if (max_len < 3 || pos == 0) { return 0; }
	const int si = d->inv[pos], n = (int)d->n;
	int len = 2, found = 0;
	int min_lcp = d->lcp[si];
	if (min_lcp > 2) {
		if ((unsigned)d->sa[si - 1] < pos) { len = min_lcp; found = d->sa[si - 1]; }
		else {
			for (int i = si - 2; i >= 0; --i) {
				const int l = d->lcp[i + 1];
				if (l < min_lcp) { if (l <= 2) { break; } min_lcp = l; }
				if ((unsigned)d->sa[i] < pos) { len = min_lcp; found = d->sa[i]; break; }
			}
		}
	}
	if (si != n - 1) {
		min_lcp = d->lcp[si + 1];
		if (min_lcp > len) {
			if ((unsigned)d->sa[si + 1] < pos) { len = min_lcp; found = d->sa[si + 1]; }
			else {
				for (int i = si + 2; i < n; ++i) {
					const int l = d->lcp[i];
					if (l < min_lcp) { if (l <= len) { break; } min_lcp = l; }
					if ((unsigned)d->sa[i] < pos) { len = min_lcp; found = d->sa[i]; break; }
				}
			}
		}
	}
	if (len > 2) { *off = pos - (unsigned)found; return (unsigned)len > max_len ? max_len : (unsigned)len; }
	return 0;
}

/* position-dependent split of the 16-bit match token (lznt1_compress.cpp:51,66): a pure function of pos */
static inline void lznt1_split(unsigned pos, unsigned* shift, unsigned* mask3)
{
	unsigned pow2 = 0x10, m = 0x1002, s = 12;
	while (pow2 < pos) { pow2 <<= 1; m = (m >> 1) + 1; --s; }
	*shift = s; *mask3 = m;
}

/* lznt1_compress_chunk (lznt1_compress.cpp:49-94). Returns the payload size, or 0 when the chunk must be
 * stored raw (running size reaches n, :85-86). out must have room for n bytes. */
static unsigned lznt1_chunk(lznt1_dict* d, const uint8_t* c, unsigned n, uint8_t* out)
{
	unsigned pos = 0, o = 0;
	static __thread lznt1_sa_dict sd;
	const int sa = orc_lznt1_sa;
	if (sa) { lznt1_sa_fill(&sd, c, n); } else { lznt1_fill(d, c, n); }
	while (pos < n) {
		uint8_t grp[16]; unsigned g = 0, flags = 0, i;
		for (i = 0; i < 8 && pos < n; ++i) {
			unsigned shift, mask3, off = 0;
			lznt1_split(pos, &shift, &mask3);
			const unsigned rem = n - pos, ml = rem < mask3 ? rem : mask3;
			const unsigned len = sa ? lznt1_sa_find(&sd, pos, ml, &off) : lznt1_find(d, c, pos, ml, &off);
			if (len >= 3) { put16(grp + g, ((off - 1) << shift) | (len - 3)); g += 2; flags |= 1u << i; pos += len; }
			else { grp[g++] = c[pos++]; }
		}
		if (o + 1 + g >= n) { return 0; }
		out[o++] = (uint8_t)flags; memcpy(out + o, grp, g); o += g;
	}
	return o;
}

/* lznt1_compress (lznt1_compress.cpp:233-273) */
static int lznt1_compress_o(const uint8_t* in, size_t n, uint8_t* out, size_t* out_len)
{
	const size_t cap = *out_len;
	size_t ip = 0, op = 0;
	lznt1_dict* d = (lznt1_dict*)calloc(1, sizeof *d);
	uint8_t tmp[4096 + 16];
	if (!d) { return ORC_MEM_ERROR; }
	while (ip < n) {
		const unsigned sz = (unsigned)((n - ip < 4096) ? n - ip : 4096);
		unsigned csz = lznt1_chunk(d, in + ip, sz, tmp);
		const unsigned psz = csz ? csz : sz;
		if (op + 2 + psz > cap) { free(d); return ORC_BUF_ERROR; }
		put16(out + op, (csz ? 0xB000u : 0x3000u) | (psz - 1));
		memcpy(out + op + 2, csz ? tmp : in + ip, psz);
		op += 2 + psz; ip += sz;
	}
	if (cap - op >= 2) { out[op] = out[op + 1] = 0; }   /* End_of_buffer, not counted (:270-271) */
	*out_len = op;
	free(d);
	return ORC_OK;
}

void orc_lznt1_match_table(const uint8_t* c, unsigned n, uint16_t* len, uint16_t* off)
{
	lznt1_dict* d = (lznt1_dict*)calloc(1, sizeof *d);
	lznt1_fill(d, c, n);
	for (unsigned p = 0; p < n; ++p) {
		unsigned shift, mask3, o = 0;
		lznt1_split(p, &shift, &mask3);
		const unsigned rem = n - p, l = lznt1_find(d, c, p, rem < mask3 ? rem : mask3, &o);
		len[p] = (uint16_t)l; off[p] = (uint16_t)(l ? o : 0);
	}
	free(d);
}

/* LZNT1 decoder with the one-shot semantics of the reference: lznt1_decompress is the streaming inflate run once over the
 * whole buffer (ALL_AT_ONCE_WRAPPER_DECOMPRESS, internal.h:616-630, over lznt1_inflate, lznt1_decompress.cpp:230-273, and
 * lznt1_decompress_chunk_read, :122-209). One place of the reference is undefined behaviour: a literal token when the 4096-byte
 * chunk limit is already reached is stored without a bounds check (:113). We report DATA_ERROR there and raise
 * orc_last_undefined() so that the tests do not ask the compiled reference about such streams. */
static __thread int orc_undefined_flag;
int orc_last_undefined(void) { return orc_undefined_flag; }

/* lznt1_decompress_chunk (:37-121) against a 4096-byte limit (:158,:179): 0 or ORC_DATA_ERROR (the callers turn every chunk
 * error into DATA_ERROR, :160-166) */
static int lznt1_chunk_decode_o(const uint8_t* in, size_t n, uint8_t* out, size_t* out_size)
{
	size_t ip = 0, op = 0;
	while (ip < n) {
		unsigned flags = in[ip++];
		for (unsigned i = 0; i < 8; ++i, flags >>= 1) {
			if (ip == n) { *out_size = op; return ORC_OK; }                                     /* :84 */
			if (!(flags & 1)) {
				if (op >= 4096) { orc_undefined_flag = 1; return ORC_DATA_ERROR; }              /* :113 (unchecked in the reference) */
				out[op++] = in[ip++]; continue;
			}
			if (ip + 2 > n) { return ORC_DATA_ERROR; }                                          /* :88 */
			unsigned shift, mask3; lznt1_split((unsigned)op, &shift, &mask3);                   /* :89 */
			const uint32_t t = get16(in + ip); ip += 2;
			const size_t off = (t >> shift) + 1; size_t len = (t & ((1u << shift) - 1)) + 3;
			if (off > op) { return ORC_DATA_ERROR; }                                            /* :96 */
			if (op + len > 4096) { return ORC_DATA_ERROR; }                                     /* :97 */
			while (len--) { out[op] = out[op - off]; ++op; }
		}
	}
	*out_size = op;
	return ORC_OK;
}

static int lznt1_decompress_o(const uint8_t* in, size_t n, uint8_t* out, size_t* out_len)
{
	const size_t cap = *out_len;
	size_t ip = 0, op = 0;
	uint8_t tmp[4096 + 32];
	orc_undefined_flag = 0;
	while (cap - op && n - ip >= 2) {                                                           /* :252 */
		const uint32_t hdr = get16(in + ip);
		if (hdr == 0) {                                                                         /* :128-134 */
			if (n - ip != 2) { return ORC_DATA_ERROR; }
			*out_len = op; return ORC_OK;                                                       /* STREAM_END -> inflate_end OK */
		}
		const size_t in_size = (hdr & 0xFFF) + 3;
		if (in_size > n - ip) { return ORC_BUF_ERROR; }                                         /* :136-143: partial chunk kept as state, the call ends with MSCOMP_OK -> BUF_ERROR (wrapper :627) */
		if ((hdr & 0x7000) != 0x3000) { return ORC_DATA_ERROR; }                                /* :151 */
		size_t size;
		if (hdr & 0x8000) {
			if (lznt1_chunk_decode_o(in + ip + 2, in_size - 2, tmp, &size) != ORC_OK) { return ORC_DATA_ERROR; }   /* :160-166, :181-187 */
			const size_t copy = size < cap - op ? size : cap - op;
			memcpy(out + op, tmp, copy); op += copy;
			if (copy < size) { return ORC_BUF_ERROR; }                                          /* :167-171: the rest waits in the state -> not a stream end */
		} else {
			size = in_size - 2;                                                                 /* :192-209 */
			const size_t copy = size < cap - op ? size : cap - op;
			memcpy(out + op, in + ip + 2, copy); op += copy;
			if (copy < size) { return ORC_BUF_ERROR; }
		}
		ip += in_size;
	}
	/* lznt1_is_possible_stream_end (:223-227): nothing left, or a single 0 byte */
	if (n - ip == 0 || (n - ip == 1 && in[ip] == 0)) { *out_len = op; return ORC_OK; }
	return ORC_BUF_ERROR;
}

/* ===================================================================================================
 * Xpress hash chains (shared by Xpress and Xpress-Huffman)
 * =================================================================================================*/
#define XP_HASH(d, p) (((((uint32_t)(d)[p] & 0x1F) << 10) ^ ((uint32_t)(d)[(p) + 1] << 5) ^ (d)[(p) + 2]) & 0x7FFF)
typedef struct { int32_t head[32768]; int32_t* pred; size_t inserted; } xp_links;

static int xp_links_init(xp_links* L, size_t n)
{
	memset(L->head, 0xFF, sizeof L->head);
	L->pred = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
	L->inserted = 0;
	return L->pred ? 0 : -1;
}
/* XpressDictionary::Fill (XpressDictionary.h:104-118): insert positions [inserted, upto), upto <= n-2 */
static void xp_links_insert(xp_links* L, const uint8_t* d, size_t upto)
{
	for (size_t p = L->inserted; p < upto; ++p) {
		const uint32_t h = XP_HASH(d, p);
		L->pred[p] = L->head[h]; L->head[h] = (int32_t)p;
	}
	if (upto > L->inserted) { L->inserted = upto; }
}
/* XpressDictionary::Find + GetMatchLength (XpressDictionary.h:145-183, :72-94), Level 3:
 * MaxChain 11, NiceLength 48 (:30). Returns 2 when nothing of length >= 3 is found. */
static uint32_t xp_find(const xp_links* L, const uint8_t* d, size_t n, size_t p, uint32_t max_off, uint32_t* off)
{
	uint32_t best = 2; int chain = 11;
	const size_t lim = n - p - 1;                 /* the buffer's final byte is never counted (:88-93) */
	for (int32_t x = L->pred[p]; chain > 0 && x >= 0 && (size_t)x + max_off >= p; x = L->pred[x], --chain) {
		if (d[x] != d[p] || d[x + 1] != d[p + 1]) { continue; }
		size_t l = 2;                             /* third byte is implied equal by the hash (:164) */
		while (l < lim && d[x + l] == d[p + l]) { ++l; }
		if (l > best) { best = (uint32_t)l; *off = (uint32_t)(p - x); if (best >= 48) { break; } }
	}
	return best;
}

void orc_xpress_match_table(const uint8_t* d, size_t n, uint32_t max_off, uint32_t* len, uint32_t* off)
{
	xp_links L;
	if (xp_links_init(&L, n)) { return; }
	if (n > 2) { xp_links_insert(&L, d, n - 2); }
	for (size_t p = 0; p < n; ++p) {
		uint32_t o = 0, l = (p + 2 < n) ? xp_find(&L, d, n, p, max_off, &o) : 2;
		len[p] = l; off[p] = l >= 3 ? o : 0;
	}
	free(L.pred);
}

/* ===================================================================================================
 * Xpress (plain LZ77)   xpress_compress (xpress_compress.cpp:240-347)
 * =================================================================================================*/
static int xpress_compress_o(const uint8_t* d, size_t n, uint8_t* out, size_t* out_len)
{
	const size_t cap = *out_len, bound = orc_max_compressed_size(ORC_XPRESS, n);
	if (n == 0) { if (cap < 4) { return ORC_BUF_ERROR; } put32(out, 0xFFFFFFFFu); *out_len = 4; return ORC_OK; }
	if (n >= 0x7FFFFFF0u) { return ORC_ARG_ERROR; }
	uint8_t* o = (cap >= bound) ? out : (uint8_t*)malloc(bound);
	xp_links L;
	if (!o || xp_links_init(&L, n)) { if (o && o != out) { free(o); } return ORC_MEM_ERROR; }

	const size_t end2 = n >= 2 ? n - 2 : 0;
	size_t w = 4, slot = 0, p = 1, filled = 0, half = 0;
	uint32_t flags = 0; unsigned cnt = 1; int have_half = 0;
	o[w++] = d[0];
	while (p < end2) {
		if (filled <= p) {                       /* ONE lazy 8 KiB Fill per token (:269) */
			filled = (filled + 0x2000 < end2) ? filled + 0x2000 : end2;
			xp_links_insert(&L, d, filled);
		}
		flags <<= 1;
		uint32_t off = 0, len = (p < filled) ? xp_find(&L, d, n, p, 0x2000, &off) : 2;  /* lagging fill => literal */
		if (len < 3) { o[w++] = d[p++]; }
		else {
			uint32_t l = len - 3;
			p += len;
			put16(o + w, ((off - 1) << 3) | (l < 7 ? l : 7)); w += 2;
			if (l >= 7) {
				l -= 7;
				if (have_half) { o[half] |= (uint8_t)((l < 15 ? l : 15) << 4); have_half = 0; }
				else { half = w; have_half = 1; o[w++] = (uint8_t)(l < 15 ? l : 15); }
				if (l >= 15) {
					l -= 15;
					o[w++] = (uint8_t)(l < 255 ? l : 255);
					if (l >= 255) {
						l += 22;                 /* back to len-3 (:297) */
						if (l <= 0xFFFF) { put16(o + w, l); w += 2; }
						else { put16(o + w, 0); put32(o + w + 2, l); w += 6; }
					}
				}
			}
			flags |= 1;
		}
		if (++cnt == 32) { put32(o + slot, flags); cnt = 0; slot = w; w += 4; }
	}
	while (p < n) {
		o[w++] = d[p++]; flags <<= 1;
		if (++cnt == 32) { put32(o + slot, flags); cnt = 0; slot = w; w += 4; }
	}
	flags = cnt ? (flags << (32 - cnt)) | ((1u << (32 - cnt)) - 1) : 0xFFFFFFFFu;
	put32(o + slot, flags);
	free(L.pred);
	if (o != out) {
		if (w > cap) { free(o); return ORC_BUF_ERROR; }
		memcpy(out, o, w); free(o);
	}
	*out_len = w;
	return ORC_OK;
}

/* Xpress decoder with the semantics of the reference's one-shot xpress_decompress (xpress_decompress.cpp:405-462, the
 * MSCOMP_WITH_OPT_DECOMPRESS build that config.h:47-48 selects; READ_SYMBOL :62-107). Its fast loop (:148-196) differs from
 * the checked loop only by what it may skip, not by what it decides, so the checked loop is restated for every token. */
static __thread uint8_t* orc_xp_mark; static __thread uint32_t* orc_xp_halfpos;   /* research: flag-word starts of the true parse */
static int xp_set_bits_are_highest(uint32_t x) { x = ~x; return !((x + 1) & x); }                 /* :41 */
static int xpress_decompress_o(const uint8_t* in, size_t n, uint8_t* out, size_t* out_len)
{
	const size_t cap = *out_len;
	size_t ip = 0, op = 0, half = 0; int have_half = 0;
	if (n < 5) {                                                                                   /* :414-418 */
		if (n == 0 || (n == 4 && get32(in) != 0xFFFFFFFFu)) { *out_len = 0; return ORC_OK; }
		return ORC_DATA_ERROR;
	}
	while (ip + 4 <= n) {                                                                          /* :425 */
		if (orc_xp_mark) { orc_xp_mark[ip] = have_half ? 2 : 1; if (have_half) { orc_xp_halfpos[ip] = (uint32_t)half; } }   /* research hook */
		uint32_t flags = get32(in + ip), flagged = flags & 0x80000000u;
		flags = (flags << 1) | 1; ip += 4;
		do {
			if (ip == n) {                                                                         /* :433-438 */
				if (!flagged || !xp_set_bits_are_highest(flags)) { return ORC_DATA_ERROR; }
				*out_len = op; return ORC_OK;
			} else if (flagged) {
				uint32_t len;                                                                      /* uint32_t like the reference: the sums below wrap */
				if (ip + 2 > n) { return ORC_DATA_ERROR; }                                         /* :82 */
				const uint32_t sym = get16(in + ip); ip += 2;
				const size_t off = (sym >> 3) + 1; len = sym & 7;
				if (len == 7) {
					if (have_half) { len = in[half] >> 4; have_half = 0; }                         /* :89 */
					else if (ip == n) { return ORC_DATA_ERROR; }
					else { half = ip; have_half = 1; len = in[ip++] & 0xF; }
					if (len == 0xF) {
						if (ip == n) { return ORC_DATA_ERROR; }                                    /* :94 */
						if ((len = in[ip++]) == 0xFF) {
							if (ip + 2 > n) { return ORC_DATA_ERROR; }
							len = get16(in + ip); ip += 2;
							if (len == 0) { if (ip + 4 > n) { return ORC_DATA_ERROR; } len = get32(in + ip); ip += 4; }
							if (len < 0xF + 0x7) { return ORC_DATA_ERROR; }                        /* :104 */
							len -= 0xF + 0x7;
						}
						len += 0xF;
					}
					len += 0x7;
				}
				len += 0x3;
				if (off > op) { return ORC_DATA_ERROR; }                                           /* :442 */
				if (len > cap - op) { return ORC_BUF_ERROR; }                                      /* :443 */
				for (uint32_t i = 0; i < len; ++i) { out[op] = out[op - off]; ++op; }
			} else if (op == cap) { return ORC_BUF_ERROR; }                                        /* :455 */
			else { out[op++] = in[ip++]; }
			flagged = flags & 0x80000000u; flags <<= 1;
		} while (flags);
	}
	return ORC_DATA_ERROR;                                                                         /* :461 */
}

/* ===================================================================================================
 * Huffman code construction
 * =================================================================================================*/
/* HuffmanEncoder<15,512>::CreateCodes (HuffmanEncoder.h:58-127) with HEAP_PUSH/HEAP_POP (:31-55).
 * Node ids 1..512 are the symbols, 513.. are internal; key = (count<<8)|depth. */
typedef struct { uint32_t w[1024]; uint16_t heap[516]; unsigned len; } hheap;
static inline void hh_push(hheap* h, unsigned x)
{
	unsigned j = ++h->len;
	while (h->w[x] < h->w[h->heap[j >> 1]]) { h->heap[j] = h->heap[j >> 1]; j >>= 1; }
	h->heap[j] = (uint16_t)x;
}
static inline unsigned hh_pop(hheap* h)
{
	const unsigned top = h->heap[1], t = h->heap[h->len--];
	unsigned i = 1;
	for (;;) {
		unsigned j = i << 1;
		if (j > h->len) { break; }
		if (j < h->len && h->w[h->heap[j + 1]] < h->w[h->heap[j]]) { ++j; }
		if (h->w[t] < h->w[h->heap[j]]) { break; }
		h->heap[i] = h->heap[j]; i = j;
	}
	h->heap[i] = (uint16_t)t;
	return top;
}
void orc_huff_lengths(const uint32_t counts[512], uint8_t lens[512])
{
	hheap h; uint16_t parent[1024];
	h.w[0] = 0;
	for (unsigned i = 0; i < 512; ++i) { h.w[i + 1] = (counts[i] ? counts[i] : 1u) << 8; }   /* :69 */
	for (;;) {
		h.len = 0; h.heap[0] = 0;
		for (unsigned i = 1; i <= 512; ++i) { hh_push(&h, i); }
		unsigned nn = 512;
		memset(parent, 0, sizeof parent);
		while (h.len > 1) {
			const unsigned a = hh_pop(&h), b = hh_pop(&h);
			const uint32_t da = h.w[a] & 0xFF, db = h.w[b] & 0xFF;
			++nn; parent[a] = parent[b] = (uint16_t)nn;
			h.w[nn] = ((h.w[a] & ~0xFFu) + (h.w[b] & ~0xFFu)) | (1 + (da > db ? da : db));
			hh_push(&h, nn);
		}
		int too_long = 0;
		for (unsigned i = 1; i <= 512; ++i) {
			unsigned depth = 0;
			for (unsigned k = i; parent[k]; k = parent[k]) { ++depth; }
			lens[i - 1] = (uint8_t)depth;
			if (depth > 15) { too_long = 1; }
		}
		if (!too_long) { return; }
		for (unsigned i = 1; i <= 512; ++i) { h.w[i] = (1 + (h.w[i] >> 9)) << 8; }          /* :100-105 */
	}
}

/* stable sort of symbol ids by key[] ascending (sorting.h:28-66 are both stable, so any stable sort agrees) */
static void stable_sort_syms(uint16_t* s, unsigned n, const uint32_t* key)
{
	for (unsigned i = 1; i < n; ++i) {
		const uint16_t x = s[i]; unsigned j = i;
		while (j > 0 && key[s[j - 1]] > key[x]) { s[j] = s[j - 1]; --j; }
		s[j] = x;
	}
}

/* HuffmanEncoder<15,512>::CreateCodesSlow (HuffmanEncoder.h:129-226): package-merge over the symbols that
 * are present; a package is (count, per-symbol multiplicity vector). */
typedef struct { uint64_t count; uint8_t mult[512]; } pm_pkg;
void orc_huff_lengths_slow(const uint32_t counts[512], uint8_t lens[512])
{
	uint16_t leaf[512]; unsigned nleaf = 0;
	memset(lens, 0, 512);
	for (unsigned i = 0; i < 512; ++i) { if (counts[i]) { leaf[nleaf++] = (uint16_t)i; lens[i] = 15; } }
	stable_sort_syms(leaf, nleaf, counts);
	if (nleaf == 0) { return; }
	if (nleaf == 1) { lens[leaf[0]] = 1; return; }
	pm_pkg* cur = (pm_pkg*)malloc(2 * 512 * sizeof(pm_pkg)), *nxt = cur + 512;
	unsigned ncur = 0;
	for (unsigned round = 0; round < 15; ++round) {
		unsigned ci = 0, li = 0, nn = 0;
		while ((ncur - ci) + (nleaf - li) > 1) {
			pm_pkg* g = &nxt[nn++];
			memset(g, 0, sizeof *g);
			for (unsigned k = 0; k < 2; ++k) {
				if (li >= nleaf || (ci < ncur && cur[ci].count < counts[leaf[li]])) {   /* strict: leaf wins ties */
					g->count += cur[ci].count;
					for (unsigned s = 0; s < 512; ++s) { g->mult[s] = (uint8_t)(g->mult[s] + cur[ci].mult[s]); }
					++ci;
				} else { g->count += counts[leaf[li]]; ++g->mult[leaf[li]]; ++li; }
			}
		}
		if (ci < ncur) { for (unsigned s = 0; s < 512; ++s) { lens[s] = (uint8_t)(lens[s] - cur[ci].mult[s]); } }
		else if (li < nleaf) { --lens[leaf[li]]; }
		pm_pkg* t = cur; cur = nxt; nxt = t; ncur = nn;
	}
	free(cur < nxt ? cur : nxt);
}

/* canonical codes by (length, symbol); identical to both HuffmanEncoder.h:109-123 and :214-222 */
static void canonical_codes(const uint8_t lens[512], uint16_t codes[512])
{
	uint32_t code = 0;
	memset(codes, 0, 512 * sizeof(uint16_t));
	for (unsigned l = 1; l <= 15; ++l) {
		for (unsigned s = 0; s < 512; ++s) { if (lens[s] == l) { codes[s] = (uint16_t)code++; } }
		code <<= 1;
	}
}

/* ===================================================================================================
 * Xpress-Huffman   xpress_huff_compress (xpress_huff_compress.cpp:247-331)
 * =================================================================================================*/
typedef struct { uint16_t sym; uint16_t offlow; uint32_t L; } xh_tok;   /* L = len-3 (matches only) */

/* OutputBitstream (Bitstream.h:111-148): MSB-first bits into LE16 words, two word slots reserved ahead,
 * raw bytes interleaved at the cursor. */
typedef struct { uint8_t* base; size_t s0, s1, cur; uint32_t acc; unsigned nbits; } obits;
static void ob_init(obits* b, uint8_t* base) { b->base = base; b->s0 = 0; b->s1 = 2; b->cur = 4; b->acc = 0; b->nbits = 0; }
static void ob_bits(obits* b, uint32_t v, unsigned k)
{
	b->nbits += k;
	b->acc |= (k ? v : 0) << (32 - b->nbits);
	if (b->nbits > 16) {
		put16(b->base + b->s0, b->acc >> 16);
		b->acc <<= 16; b->nbits -= 16;
		b->s0 = b->s1; b->s1 = b->cur; b->cur += 2;
	}
}
static size_t ob_finish(obits* b) { put16(b->base + b->s0, b->acc >> 16); put16(b->base + b->s1, 0); return b->cur; }

static size_t xh_emit(const xh_tok* t, size_t nt, const uint8_t* lens, const uint16_t* codes, uint8_t* out)
{
	obits b; ob_init(&b, out);
	for (size_t i = 0; i < nt; ++i) {                                   /* xh_compress_encode (:195-245) */
		const unsigned s = t[i].sym;
		ob_bits(&b, codes[s], lens[s]);
		if (s >= 0x100) {
			const uint32_t L = t[i].L;
			if ((s & 0xF) == 0xF) {
				if (L > 0xFFFF) { out[b.cur] = 0xFF; put16(out + b.cur + 1, 0); put32(out + b.cur + 3, L); b.cur += 7; }
				else if (L >= 270) { out[b.cur] = 0xFF; put16(out + b.cur + 1, L); b.cur += 3; }
				else { out[b.cur++] = (uint8_t)(L - 15); }
			}
			ob_bits(&b, t[i].offlow, (s >> 4) & 0xF);
		}
	}
	return ob_finish(&b);
}

static int xpress_huff_compress_o(const uint8_t* d, size_t n, uint8_t* out, size_t* out_len)
{
	const size_t cap = *out_len;
	if (n == 0) { *out_len = 0; return ORC_OK; }                         /* :249 */
	if (n >= 0x7FFFFFF0u) { return ORC_ARG_ERROR; }
	xp_links L;
	xh_tok* tok = (xh_tok*)malloc(65537 * sizeof(xh_tok));
	uint8_t* tmp = (uint8_t*)malloc(65536 + 600);
	if (!tok || !tmp || xp_links_init(&L, n)) { free(tok); free(tmp); return ORC_MEM_ERROR; }
	if (n > 2) { xp_links_insert(&L, d, n - 2); }                       /* links are causal: fill everything up front */
	size_t op = 0;
	for (size_t cs = 0; cs < n; cs += 65536) {
		const size_t ce = (n - cs > 65536) ? cs + 65536 : n;
		const int last = (ce == n);
		uint32_t counts[512]; uint8_t lens[512]; uint16_t codes[512];
		size_t nt = 0, extra = 0;
		memset(counts, 0, sizeof counts);
		for (size_t p = cs; p < ce; ) {                                   /* xh_compress_lz77 (:52-153) */
			const size_t rem = ce - p;
			uint32_t off = 0, len = (rem >= 3) ? xp_find(&L, d, n, p, 0xFFFF, &off) : 2;
			if (len >= 3) {
				if (len > rem) { len = (uint32_t)rem; }
				const unsigned ob = ilog2(off);
				const uint32_t l3 = len - 3;
				tok[nt].sym = (uint16_t)(0x100 | (ob << 4) | (l3 < 15 ? l3 : 15));
				tok[nt].offlow = (uint16_t)(off ^ (1u << ob)); tok[nt].L = l3;
				extra += (l3 > 0xFFFF) ? 7 : (l3 >= 270) ? 3 : (l3 >= 15) ? 1 : 0;
				p += len;
			} else { tok[nt].sym = d[p++]; tok[nt].offlow = 0; tok[nt].L = 0; }
			++counts[tok[nt++].sym];
		}
		if (last) { tok[nt].sym = 0x100; tok[nt].offlow = 0; tok[nt].L = 0; ++counts[0x100]; ++nt; }   /* EOS (:127-144) */
		orc_huff_lengths(counts, lens);
		size_t bits = 16;                                                 /* xh_calc_compressed_len (:181-188) */
		for (unsigned s = 0; s < 512; ++s) { bits += (size_t)(lens[s] + (s >= 0x100 ? ((s >> 4) & 0xF) : 0)) * counts[s]; }
		size_t comp = (bits + 15) / 16 * 2 + extra;
		if (comp > (last ? (ce - cs) + 36 : 65538)) {                     /* fallback (:274-280 / :310-316) */
			memset(counts, 0, sizeof counts);
			nt = 0;
			for (size_t p = cs; p < ce; ++p) { tok[nt].sym = d[p]; tok[nt].offlow = 0; tok[nt].L = 0; ++counts[d[p]]; ++nt; }
			if (last) { tok[nt].sym = 0x100; tok[nt].offlow = 0; tok[nt].L = 0; ++counts[0x100]; ++nt; }
			orc_huff_lengths_slow(counts, lens);
			bits = 16;
			for (unsigned s = 0; s <= 0x100; ++s) { bits += (size_t)lens[s] * counts[s]; }
			comp = (bits + 15) / 16 * 2;
		}
		if (cap - op < 256 + comp) { free(tok); free(tmp); free(L.pred); return ORC_BUF_ERROR; }
		canonical_codes(lens, codes);
		for (unsigned i = 0; i < 256; ++i) { out[op + i] = (uint8_t)(lens[2 * i] | (lens[2 * i + 1] << 4)); }
		const size_t wrote = xh_emit(tok, nt, lens, codes, tmp);
		if (wrote != comp) { free(tok); free(tmp); free(L.pred); return ORC_DATA_ERROR; }   /* self-check */
		memcpy(out + op + 256, tmp, comp);
		op += 256 + comp;
	}
	free(tok); free(tmp); free(L.pred);
	*out_len = op;
	return ORC_OK;
}

/* Xpress-Huffman decoder with the semantics of the reference's xpress_huff_decompress (xpress_huff_decompress.cpp:130-162) and
 * its chunk loop (:39-129). The fast loop (:50-84) decides nothing differently from the checked loop (:87-127) while it runs
 * (>= 13 input bytes and >= 16 output bytes away from the ends), so the checked loop is restated for every symbol.
 * InputBitstream: Bitstream.h:34-106; HuffmanDecoder<15,512>: HuffmanDecoder.h:28-114. */
/* research hooks (tools/xh_marker_study.py, tools/xh_depth_study.py): chunk starts, copy-chain depth per output byte */
static __thread uint64_t* orc_xh_starts; static __thread size_t orc_xh_nstarts, orc_xh_maxstarts;
static __thread uint32_t* orc_xh_depth;      /* research: copy-chain depth per output byte (0 = literal) */
static __thread int orc_xh_depth_mode;
static __thread uint32_t* orc_xh_src;        /* research: source of every output byte (itself for a literal) */
typedef struct { const uint8_t* in; const uint8_t* end; uint32_t mask; unsigned bits; } xh_ibs;
static inline uint32_t xh_peek(const xh_ibs* b, unsigned n) { return (b->mask >> 16) >> (16 - n); }              /* Bitstream.h:55 */
static inline int xh_mask_is_zero(const xh_ibs* b) { return b->bits == 0 || (b->mask >> (32 - b->bits)) == 0; } /* :58 */
static inline void xh_skip(xh_ibs* b, unsigned n)                                                                /* :63-74 */
{
	b->mask <<= n; b->bits -= n;
	if (b->bits < 16 && b->in + 2 <= b->end) { b->mask |= get16(b->in) << (16 - b->bits); b->bits |= 0x10; b->in += 2; }
}
typedef struct { uint32_t lims[16], poss[16]; uint16_t syms[512]; uint8_t lens[512]; } xh_dec;
static int xh_set_code_lengths(xh_dec* d, const uint8_t cl[512])                                                  /* HuffmanDecoder.h:42-90 */
{
	uint32_t cnts[16] = { 0 }, last = 0, index = 0;
	memset(d->syms, 0xFF, sizeof d->syms);
	for (unsigned s = 0; s < 512; ++s) { ++cnts[cl[s]]; }
	cnts[0] = 0;
	d->lims[0] = 0; d->lims[15] = 32768;
	for (unsigned len = 1; len <= 9; ++len) {
		const uint32_t inc = cnts[len] << (15 - len);
		if (last + inc > 32768) { return 0; }
		d->lims[len] = (last += inc);
		const uint32_t limit = last >> 6;
		memset(d->lens + index, (int)len, limit - index); index = limit;
	}
	for (unsigned len = 10; len < 15; ++len) {
		const uint32_t inc = cnts[len] << (15 - len);
		if (last + inc > 32768) { return 0; }
		d->lims[len] = (last += inc);
	}
	if (last + cnts[15] > 32768) { return 0; }
	d->poss[0] = 0;
	for (unsigned len = 1; len <= 15; ++len) { d->poss[len] = d->poss[len - 1] + cnts[len - 1]; }
	memcpy(cnts, d->poss, sizeof cnts);
	for (unsigned s = 0; s < 512; ++s) { if (cl[s]) { d->syms[cnts[cl[s]]++] = (uint16_t)s; } }
	return 1;
}
#define XH_INVALID 0xFFFFu
static unsigned xh_decode_symbol(const xh_dec* d, xh_ibs* b)                                                      /* HuffmanDecoder.h:92-102 */
{
	unsigned n; const unsigned r = b->bits;
	const uint32_t x = r < 15 ? (xh_peek(b, r) << (15 - r)) : xh_peek(b, 15);
	if (x < d->lims[9]) { n = d->lens[x >> 6]; } else { for (n = 10; x >= d->lims[n]; ++n) { } }
	if (n > r) { return XH_INVALID; }
	xh_skip(b, n);
	const uint32_t s = d->poss[n] + ((x - d->lims[n - 1]) >> (15 - n));
	return s >= 512 ? XH_INVALID : d->syms[s];
}
/* xpress_huff_decompress_chunk: 0 = chunk done, 1 = stream end, < 0 = error. */
static int xh_decompress_chunk_o(const uint8_t** pin, const uint8_t* in_end, uint8_t* out_base, size_t* pop, size_t cap, const xh_dec* d)
{
	xh_ibs b = { *pin + 4, in_end, (get16(*pin) << 16) | get16(*pin + 2), 32 };                                   /* Bitstream.h:44 */
	size_t op = *pop;
	const size_t chunk_end = op + 65536;
	while (op < chunk_end || !xh_mask_is_zero(&b)) {                                                              /* :87 */
		const unsigned sym = xh_decode_symbol(d, &b);
		if (sym == XH_INVALID) { return ORC_DATA_ERROR; }
		if (sym == 0x100 && b.in == b.end && xh_mask_is_zero(&b)) { *pin = b.in; *pop = op; return 1; }           /* :91 */
		if (sym < 0x100) {
			if (op == cap) { return ORC_BUF_ERROR; }
			if (orc_xh_depth) { orc_xh_depth[op] = 0; }
			if (orc_xh_src) { orc_xh_src[op] = (uint32_t)op; }
			out_base[op++] = (uint8_t)sym;
		} else {
			uint32_t len = sym & 0xF, off;
			if (len == 0xF) {
				if (b.end - b.in < 1) { return ORC_DATA_ERROR; }
				if ((len = *b.in++) == 0xFF) {
					if (b.end - b.in < 2) { return ORC_DATA_ERROR; }
					len = get16(b.in); b.in += 2;
					if (len == 0) {
						if (b.end - b.in < 4) { return ORC_DATA_ERROR; }
						len = get32(b.in); b.in += 4;
					}
					if (len < 0xF) { return ORC_DATA_ERROR; }
					len -= 0xF;
				}
				len += 0xF;
			}
			len += 3;
			{
				const unsigned off_bits = (sym >> 4) & 0xF;
				if (off_bits > b.bits) { return ORC_DATA_ERROR; }                                                 /* :117 */
				off = xh_peek(&b, off_bits) + (1u << off_bits); xh_skip(&b, off_bits);
			}
			if (off > op) { return ORC_DATA_ERROR; }                                                              /* :120 */
			if (len > cap - op) { return ORC_BUF_ERROR; }                                                         /* :121 */
			if (orc_xh_src) { for (uint32_t i = 0; i < len; ++i) { orc_xh_src[op + i] = (uint32_t)(op - off + (i % off)); } }
			if (orc_xh_depth) {   /* mode 0: byte i copies byte i - off; mode 1: byte i of a match copies byte (i mod off) of its first period */
				for (uint32_t i = 0; i < len; ++i) { orc_xh_depth[op + i] = orc_xh_depth[orc_xh_depth_mode ? op - off + (i % off) : op + i - off] + 1; }
			}
			for (uint32_t i = 0; i < len; ++i) { out_base[op] = out_base[op - off]; ++op; }
		}
	}
	*pop = op; *pin = b.in;
	if (xh_decode_symbol(d, &b) == 0x100 && b.in == b.end && xh_mask_is_zero(&b)) { *pin = b.in; return 1; }     /* :130-134 */
	return 0;
}
static int xpress_huff_decompress_o(const uint8_t* in, size_t n, uint8_t* out, size_t* out_len)
{
	const size_t cap = *out_len;
	const uint8_t* ip = in; const uint8_t* in_end = in + n;
	size_t op = 0;
	int status;
	xh_dec* d = (xh_dec*)malloc(sizeof *d);
	if (!d) { return ORC_MEM_ERROR; }
	do {
		uint8_t cl[512];
		if (in_end - ip < 260) {                                                                                  /* :140-144 */
			if (ip != in_end) { free(d); return ORC_DATA_ERROR; }
			break;
		}
		if (orc_xh_starts && orc_xh_nstarts < orc_xh_maxstarts) { orc_xh_starts[orc_xh_nstarts] = (uint64_t)(ip - in); } if (orc_xh_starts) { ++orc_xh_nstarts; }
		for (unsigned i = 0; i < 256; ++i) { cl[2 * i] = ip[i] & 0xF; cl[2 * i + 1] = ip[i] >> 4; }
		ip += 256;
		if (!xh_set_code_lengths(d, cl)) { free(d); return ORC_DATA_ERROR; }
		status = xh_decompress_chunk_o(&ip, in_end, out, &op, cap, d);
		if (status < 0) { free(d); return status; }
	} while (status != 1);
	free(d);
	*out_len = op;
	return ORC_OK;
}

/* ===================================================================================================
 * dispatch   (mscomp.cpp:104-117, :119-134)
 * =================================================================================================*/
int orc_compress(int format, const uint8_t* in, size_t n, uint8_t* out, size_t* out_len)
{
	switch (format) {
	case ORC_NONE: if (n > *out_len) { return ORC_BUF_ERROR; } memcpy(out, in, n); *out_len = n; return ORC_OK;
	case ORC_LZNT1: return lznt1_compress_o(in, n, out, out_len);
	case ORC_XPRESS: return xpress_compress_o(in, n, out, out_len);
	case ORC_XPRESS_HUFF: return xpress_huff_compress_o(in, n, out, out_len);
	default: return ORC_ARG_ERROR;
	}
}
int orc_decompress(int format, const uint8_t* in, size_t n, uint8_t* out, size_t* out_len)
{
	switch (format) {
	case ORC_NONE: if (n > *out_len) { return ORC_BUF_ERROR; } memcpy(out, in, n); *out_len = n; return ORC_OK;
	case ORC_LZNT1: return lznt1_decompress_o(in, n, out, out_len);
	case ORC_XPRESS: return xpress_decompress_o(in, n, out, out_len);
	case ORC_XPRESS_HUFF: return xpress_huff_decompress_o(in, n, out, out_len);
	default: return ORC_ARG_ERROR;
	}
}

/* ===================================================================================================
 * multi-threaded independent-unit driver (cpu_baseline leg of bench.py, batch parity tests)
 * =================================================================================================*/
typedef struct {
	int format; const uint8_t* in; const uint64_t* in_off; size_t n_units;
	uint8_t* out; const uint64_t* out_off; uint64_t* out_len; int32_t* status;
	size_t next; pthread_mutex_t mu;
} units_job;
static void* units_worker(void* arg)
{
	units_job* j = (units_job*)arg;
	for (;;) {
		pthread_mutex_lock(&j->mu);
		const size_t i0 = j->next, i1 = (i0 + 16 < j->n_units) ? i0 + 16 : j->n_units;
		j->next = i1;
		pthread_mutex_unlock(&j->mu);
		if (i0 >= j->n_units) { return NULL; }
		for (size_t i = i0; i < i1; ++i) {
			size_t ol = (size_t)(j->out_off[i + 1] - j->out_off[i]);
			const int st = orc_compress(j->format, j->in + j->in_off[i], (size_t)(j->in_off[i + 1] - j->in_off[i]),
			                            j->out + j->out_off[i], &ol);
			j->status[i] = st; j->out_len[i] = st == ORC_OK ? ol : 0;
		}
	}
}
int orc_compress_units(int format, const uint8_t* in, const uint64_t* in_off, size_t n_units,
                       uint8_t* out, const uint64_t* out_off, uint64_t* out_len, int32_t* status, int threads)
{
	units_job j = { format, in, in_off, n_units, out, out_off, out_len, status, 0, PTHREAD_MUTEX_INITIALIZER };
	if (threads < 1) { threads = 1; }
	if (threads > 256) { threads = 256; }
	pthread_t th[256]; pthread_attr_t at;
	pthread_attr_init(&at); pthread_attr_setstacksize(&at, 8u << 20);
	for (int t = 1; t < threads; ++t) { if (pthread_create(&th[t], &at, units_worker, &j)) { threads = t; break; } }
	units_worker(&j);
	for (int t = 1; t < threads; ++t) { pthread_join(th[t], NULL); }
	pthread_attr_destroy(&at);
	return 0;
}

/* The same driver for ANY function with the one-shot signature (the compiled reference's ms_compress / ms_decompress, passed
 * as a pointer by bench.py): `passes` passes over the units, seconds spent inside (no interpreter between the calls). */
#include <time.h>
typedef int (*one_shot_fn)(int, const uint8_t*, size_t, uint8_t*, size_t*);
typedef struct { units_job j; one_shot_fn fn; } fn_job;
static void* fn_worker(void* arg)
{
	fn_job* f = (fn_job*)arg; units_job* j = &f->j;
	for (;;) {
		pthread_mutex_lock(&j->mu);
		const size_t i0 = j->next, i1 = (i0 + 4 < j->n_units) ? i0 + 4 : j->n_units;
		j->next = i1;
		pthread_mutex_unlock(&j->mu);
		if (i0 >= j->n_units) { return NULL; }
		for (size_t i = i0; i < i1; ++i) {
			size_t ol = (size_t)(j->out_off[i + 1] - j->out_off[i]);
			const int st = f->fn(j->format, j->in + j->in_off[i], (size_t)(j->in_off[i + 1] - j->in_off[i]), j->out + j->out_off[i], &ol);
			j->status[i] = st; j->out_len[i] = st == ORC_OK ? ol : 0;
		}
	}
}
double orc_time_units(void* fn, int format, const uint8_t* in, const uint64_t* in_off, size_t n_units,
                      uint8_t* out, const uint64_t* out_off, uint64_t* out_len, int32_t* status, int threads, int passes)
{
	struct timespec t0, t1;
	if (threads < 1) { threads = 1; }
	if (threads > 256) { threads = 256; }
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int p = 0; p < passes; ++p) {
		fn_job f = { { format, in, in_off, n_units, out, out_off, out_len, status, 0, PTHREAD_MUTEX_INITIALIZER }, fn ? (one_shot_fn)fn : (one_shot_fn)orc_compress };
		pthread_t th[256]; pthread_attr_t at; int started = threads;
		pthread_attr_init(&at); pthread_attr_setstacksize(&at, 8u << 20);
		for (int t = 1; t < threads; ++t) { if (pthread_create(&th[t], &at, fn_worker, &f)) { started = t; break; } }
		fn_worker(&f);
		for (int t = 1; t < started; ++t) { pthread_join(th[t], NULL); }
		pthread_attr_destroy(&at);
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* The same with explicit unit lengths and one unit per grab: units may alias each other in `in` (the replicas of bench.py's
 * job are the same 12 files) and are handed out one at a time in the order given (bench.py passes whole files longest first). */
typedef struct { int format; const uint8_t* in; const uint64_t* in_off; const uint64_t* in_len; size_t n_units; uint8_t* out; const uint64_t* out_off;
                 const uint64_t* out_cap; uint64_t* out_len; int32_t* status; size_t next; pthread_mutex_t mu; one_shot_fn fn; double busy; } ex_job;
/* seconds spent INSIDE fn, summed over the threads and passes of ONE orc_time_units_ex2 call: bytes x threads / this = the rate the same cores
 * would give on a queue long enough to keep every thread busy (bench.py's load-balanced CPU figure). Kept in the job and returned through an out
 * parameter: two concurrent calls do not see each other's figure. */
static void* ex_worker(void* arg)
{
	ex_job* j = (ex_job*)arg;
	double busy = 0.0;
	for (;;) {
		pthread_mutex_lock(&j->mu);
		const size_t i = j->next++;
		pthread_mutex_unlock(&j->mu);
		if (i >= j->n_units) { break; }
		size_t ol = (size_t)j->out_cap[i];
		struct timespec a, b;
		clock_gettime(CLOCK_MONOTONIC, &a);
		const int st = j->fn(j->format, j->in + j->in_off[i], (size_t)j->in_len[i], j->out + j->out_off[i], &ol);
		clock_gettime(CLOCK_MONOTONIC, &b);
		j->status[i] = st; j->out_len[i] = st == ORC_OK ? ol : 0;
		busy += (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
	}
	pthread_mutex_lock(&j->mu);
	j->busy += busy;
	pthread_mutex_unlock(&j->mu);
	return NULL;
}
double orc_time_units_ex2(void* fn, int format, const uint8_t* in, const uint64_t* in_off, const uint64_t* in_len, size_t n_units,
                          uint8_t* out, const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, int32_t* status, int threads, int passes,
                          double* busy_seconds)
{
	struct timespec t0, t1;
	double busy = 0.0;
	if (threads < 1) { threads = 1; }
	if (threads > 256) { threads = 256; }
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int p = 0; p < passes; ++p) {
		ex_job f = { format, in, in_off, in_len, n_units, out, out_off, out_cap, out_len, status, 0, PTHREAD_MUTEX_INITIALIZER, fn ? (one_shot_fn)fn : (one_shot_fn)orc_compress, 0.0 };
		pthread_t th[256]; pthread_attr_t at; int started = threads;
		pthread_attr_init(&at); pthread_attr_setstacksize(&at, 8u << 20);
		for (int t = 1; t < threads; ++t) { if (pthread_create(&th[t], &at, ex_worker, &f)) { started = t; break; } }
		ex_worker(&f);
		for (int t = 1; t < started; ++t) { pthread_join(th[t], NULL); }
		pthread_attr_destroy(&at);
		busy += f.busy;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if (busy_seconds) { *busy_seconds = busy; }
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
double orc_time_units_ex(void* fn, int format, const uint8_t* in, const uint64_t* in_off, const uint64_t* in_len, size_t n_units,
                         uint8_t* out, const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, int32_t* status, int threads, int passes)
{
	return orc_time_units_ex2(fn, format, in, in_off, in_len, n_units, out, out_off, out_cap, out_len, status, threads, passes, NULL);
}

/* research helper (DESIGN_DECODERS.md, chunk-parallel Xpress-Huffman decoding): where do the chunks of a stream start? Decodes the stream
 * (cap bytes of room) and records the input offset of every chunk's table; returns the number of chunks or a negative status. */
long orc_xh_chunk_starts(const uint8_t* in, size_t n, size_t cap, uint64_t* starts, size_t max_starts)
{
	uint8_t* out = (uint8_t*)malloc(cap + 64);
	if (!out) { return ORC_MEM_ERROR; }
	size_t len = cap;
	orc_xh_starts = starts; orc_xh_nstarts = 0; orc_xh_maxstarts = max_starts;
	const int st = xpress_huff_decompress_o(in, n, out, &len);
	const long k = (long)orc_xh_nstarts;
	orc_xh_starts = NULL;
	free(out);
	return st == ORC_OK ? k : st;
}

/* research helper (DESIGN_DECODERS.md): how long are the copy chains of a stream? depth[i] = 0 for a literal byte, else depth of its source + 1
 * (depth: cap entries). Returns the decoded length or a negative status. */
void orc_xh_depth_first_period(int on) { orc_xh_depth_mode = on; }
long long orc_xh_copy_depths(const uint8_t* in, size_t n, size_t cap, uint32_t* depth)
{
	uint8_t* out = (uint8_t*)malloc(cap + 64);
	if (!out) { return ORC_MEM_ERROR; }
	size_t len = cap;
	orc_xh_depth = depth;
	const int st = xpress_huff_decompress_o(in, n, out, &len);
	orc_xh_depth = NULL;
	free(out);
	return st == ORC_OK ? (long long)len : st;
}

/* research helper (DESIGN_DECODERS.md): ONE chunk of an Xpress-Huffman stream, parsed without its output -- what a speculative chunk wave would
 * compute. `at` = offset of the chunk's 256-byte table. res[0] = offset where the next chunk starts, res[1] = bytes the chunk produces,
 * res[2] = how far its matches reach in front of the chunk's first byte (max of offset - bytes produced so far, 0 if none), res[3] = tokens.
 * Returns 0 (chunk done), 1 (the stream ends here) or a negative status (bad table / bad data; capacity is not checked). */
int orc_xh_parse_chunk(const uint8_t* in, size_t n, size_t at, uint64_t res[4])
{
	xh_dec d; uint8_t cl[512];
	res[0] = res[1] = res[2] = res[3] = 0;
	if (n - at < 260) { return ORC_DATA_ERROR; }
	for (unsigned i = 0; i < 256; ++i) { cl[2 * i] = in[at + i] & 0xF; cl[2 * i + 1] = in[at + i] >> 4; }
	if (!xh_set_code_lengths(&d, cl)) { return ORC_DATA_ERROR; }
	const uint8_t* p = in + at + 256; const uint8_t* in_end = in + n;
	xh_ibs b = { p + 4, in_end, (get16(p) << 16) | get16(p + 2), 32 };
	uint64_t op = 0, reach = 0, ntok = 0;
	int ended = 0;
	while (op < 65536 || !xh_mask_is_zero(&b)) {
		const unsigned sym = xh_decode_symbol(&d, &b);
		if (sym == XH_INVALID) { return ORC_DATA_ERROR; }
		if (sym == 0x100 && b.in == b.end && xh_mask_is_zero(&b)) { ended = 1; break; }
		++ntok;
		if (sym < 0x100) { ++op; continue; }
		uint32_t len = sym & 0xF, off;
		if (len == 0xF) {
			if (b.end - b.in < 1) { return ORC_DATA_ERROR; }
			if ((len = *b.in++) == 0xFF) {
				if (b.end - b.in < 2) { return ORC_DATA_ERROR; }
				len = get16(b.in); b.in += 2;
				if (len == 0) { if (b.end - b.in < 4) { return ORC_DATA_ERROR; } len = get32(b.in); b.in += 4; }
				if (len < 0xF) { return ORC_DATA_ERROR; }
				len -= 0xF;
			}
			len += 0xF;
		}
		len += 3;
		const unsigned off_bits = (sym >> 4) & 0xF;
		if (off_bits > b.bits) { return ORC_DATA_ERROR; }
		off = xh_peek(&b, off_bits) + (1u << off_bits); xh_skip(&b, off_bits);
		if (off > op && off - op > reach) { reach = off - op; }
		op += len;
	}
	if (!ended) {
		const uint8_t* keep = b.in;
		if (xh_decode_symbol(&d, &b) == 0x100 && b.in == b.end && xh_mask_is_zero(&b)) { ended = 1; } else { b.in = keep; }
	}
	res[0] = (uint64_t)(b.in - in); res[1] = op; res[2] = reach; res[3] = ntok;
	return ended;
}

/* research helper (DESIGN_DECODERS.md): src[i] = i for a literal byte, else the byte of the match's first period it copies (< i); out: the bytes. */
long long orc_xh_sources(const uint8_t* in, size_t n, size_t cap, uint32_t* src, uint8_t* out)
{
	size_t len = cap;
	orc_xh_src = src;
	const int st = xpress_huff_decompress_o(in, n, out, &len);
	orc_xh_src = NULL;
	return st == ORC_OK ? (long long)len : st;
}

/* research helpers (would ONE Xpress stream parse in parallel?). orc_xp_flag_starts: mark[p] = 1 / 2 where the true parse reads a flag word
 * (2: a length nibble is pending, its byte at halfpos[p]). orc_xp_sync: parse (input side only) from `start` as if a flag word began there
 * with no nibble pending; returns the offset of the first flag word at which this parse is in the state of the true one, or n if it never is
 * (it ran off the input or into an impossible token). */
long long orc_xp_flag_starts(const uint8_t* in, size_t n, size_t cap, uint8_t* mark, uint32_t* halfpos)
{
	uint8_t* out = (uint8_t*)malloc(cap + 64);
	if (!out) { return ORC_MEM_ERROR; }
	size_t len = cap;
	orc_xp_mark = mark; orc_xp_halfpos = halfpos;
	const int st = xpress_decompress_o(in, n, out, &len);
	orc_xp_mark = NULL; orc_xp_halfpos = NULL;
	free(out);
	return st == ORC_OK ? (long long)len : st;
}
size_t orc_xp_sync2(const uint8_t* in, size_t n, size_t start, const uint8_t* mark, const uint32_t* halfpos, int pending);
size_t orc_xp_sync(const uint8_t* in, size_t n, size_t start, const uint8_t* mark, const uint32_t* halfpos) { return orc_xp_sync2(in, n, start, mark, halfpos, 0); }
/* pending = 1: the parse starts as if a nibble were pending in a byte it has not seen (its value then counts as "not 15": right 15 times in 16) */
size_t orc_xp_sync2(const uint8_t* in, size_t n, size_t start, const uint8_t* mark, const uint32_t* halfpos, int pending)
{
	const size_t UNSEEN = (size_t)-1;
	size_t ip = start, half = pending ? UNSEEN : 0; int have_half = pending ? 1 : 0;
	while (ip + 4 <= n) {
		if ((mark[ip] == 1 && !have_half) || (mark[ip] == 2 && have_half && (half == UNSEEN || halfpos[ip] == (uint32_t)half))) { return ip; }
		uint32_t flags = get32(in + ip);
		ip += 4;
		for (int t = 0; t < 32; ++t, flags <<= 1) {
			if (ip >= n) { return n; }
			if (!(flags & 0x80000000u)) { ++ip; continue; }
			if (ip + 2 > n) { return n; }
			const uint32_t sym = get16(in + ip); ip += 2;
			if ((sym & 7) != 7) { continue; }
			uint32_t len;
			if (have_half) { len = half == UNSEEN ? 0 : in[half] >> 4; have_half = 0; }
			else { if (ip >= n) { return n; } half = ip; have_half = 1; len = in[ip++] & 0xF; }
			if (len != 0xF) { continue; }
			if (ip >= n) { return n; }
			if (in[ip++] != 0xFF) { continue; }
			if (ip + 2 > n) { return n; }
			const uint32_t l16 = get16(in + ip); ip += 2;
			if (l16 == 0) { if (ip + 4 > n) { return n; } ip += 4; }
		}
	}
	return n;
}

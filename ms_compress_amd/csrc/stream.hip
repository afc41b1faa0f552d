// stream.hip -- the reference's streaming compression entry points for LZNT1, host side.
//
// ms_deflate_init / ms_deflate / ms_deflate_end (/root/reference/src/mscomp.cpp:136-165) and lznt1_deflate_init / lznt1_deflate /
// lznt1_deflate_end (/root/reference/src/lznt1_compress.cpp:132-231, chunk writer :95-131, stream macros
// /root/reference/include/mscomp/internal.h:473-533). LZNT1 is the only format whose streaming compressor works in the reference
// (Xpress: xpress_compress.cpp:52-73,219-221 return errors; Xpress+Huffman: none, mscomp.cpp:141).
//
// The stream object is a host-side state machine over independent 4 KiB chunks: what is buffered when, when a short chunk is cut
// (MSCOMP_FLUSH / MSCOMP_FINISH), how a chunk that does not fit the caller's output window is handed out piecewise. That logic is
// restated here; every chunk image (header + payload) is produced by the HIP compressor: all full chunks that are certain to fit the
// output window go to the GPU as ONE batch (a unit of k x 4096 bytes is exactly the concatenation of its k chunk images).
#include "../../include/mscomp_amd.h"
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

const size_t CHUNK = 4096;

struct DeflateState {                       // what lznt1_compress.cpp:32-40 keeps: one partial input chunk, one pending output chunk
	bool finished = false;
	uint8_t in[CHUNK];
	size_t in_needed = 0, in_avail = 0;
	uint8_t out[CHUNK + 2];
	size_t out_pos = 0, out_avail = 0;
	uint8_t* scratch = nullptr; size_t scratch_cap = 0;      // host staging of a batch's output
	// Look-ahead: when the caller's window forces chunk-by-chunk hand-over, the whole chunks that follow in the caller's input are
	// compressed in the SAME GPU call (up to AHEAD of them) and kept with a copy of their input; a later chunk is served from here
	// when its 4096 input bytes are still the ones that were compressed. Nothing the caller can observe changes: the stream's
	// pointers and counters move exactly as if every chunk had been compressed when the reference compresses it.
	std::vector<uint8_t> ah_in, ah_img;
	std::vector<uint32_t> ah_off;                            // image k = ah_img[ah_off[k], ah_off[k + 1])
	size_t ah_next = 0;
};
const size_t AHEAD = 256;

bool stream_ok(const mscomp_stream* s, MSCompFormat f, bool compressing = true)     // CHECK_STREAM (internal.h:481)
{
	return s && s->format == f && s->compressing == compressing && !(s->in == nullptr && s->in_avail != 0) && !(s->out == nullptr && s->out_avail != 0);
}
void take_in(mscomp_stream* s, size_t n)  { s->in += n;  s->in_total += n;  s->in_avail -= n; }
void give_out(mscomp_stream* s, size_t n) { s->out += n; s->out_total += n; s->out_avail -= n; }

// k full chunks (or one short chunk) -> their images, back to back, compressed on the GPU
MSCompStatus gpu_images(DeflateState* st, const uint8_t* in, size_t n, uint8_t** img, size_t* img_len)
{
	const size_t cap = lznt1_max_compressed_size(n);
	if (st->scratch_cap < cap) {
		free(st->scratch);
		st->scratch = static_cast<uint8_t*>(malloc(cap + 64)); st->scratch_cap = st->scratch ? cap : 0;
		if (!st->scratch) { return MSCOMP_MEM_ERROR; }
	}
	size_t len = cap;
	const MSCompStatus r = lznt1_compress(in, n, st->scratch, &len);
	if (r != MSCOMP_OK) { return r; }
	*img = st->scratch; *img_len = len;
	return MSCOMP_OK;
}

// lznt1_compress_chunk_write (:95-131): one chunk of n <= 4096 bytes; goes to the caller's window when header + n bytes are sure to
// fit, else through the state's buffer
// `following` = bytes of caller input that lie behind the chunk at `in` (0 when the chunk comes from the state's buffer)
MSCompStatus write_chunk(mscomp_stream* s, DeflateState* st, const uint8_t* in, size_t n, size_t following = 0)
{
	uint8_t* img; size_t len;
	if (n == CHUNK && st->ah_next + 1 < st->ah_off.size() && memcmp(in, st->ah_in.data() + st->ah_next * CHUNK, CHUNK) == 0) {
		img = st->ah_img.data() + st->ah_off[st->ah_next]; len = st->ah_off[st->ah_next + 1] - st->ah_off[st->ah_next];
		++st->ah_next;
	} else {
		st->ah_off.clear(); st->ah_next = 0;
		size_t k = n == CHUNK ? 1 + following / CHUNK : 1;
		if (k > AHEAD) { k = AHEAD; }
		const MSCompStatus r = gpu_images(st, in, k > 1 ? k * CHUNK : n, &img, &len);
		if (r != MSCOMP_OK) { return r; }
		if (k > 1) {                                         // keep all k images (an image says how long it is: header & 0xFFF + 3)
			st->ah_in.assign(in, in + k * CHUNK);
			st->ah_img.assign(img, img + len);
			st->ah_off.assign(1, 0u);
			size_t o = 0;
			for (size_t i = 0; i < k; ++i) { o += (((size_t)st->ah_img[o] | ((size_t)st->ah_img[o + 1] << 8)) & 0x0FFFu) + 3; st->ah_off.push_back((uint32_t)o); }
			img = st->ah_img.data(); len = st->ah_off[1]; st->ah_next = 1;
		}
	}
	if (s->out_avail >= n + 2) { memcpy(s->out, img, len); give_out(s, len); return MSCOMP_OK; }
	memcpy(st->out, img, len);
	const size_t copy = len < s->out_avail ? len : s->out_avail;
	memcpy(s->out, st->out, copy);
	give_out(s, copy);
	st->out_pos = copy; st->out_avail = len - copy;
	return MSCOMP_OK;
}

// ---- streaming decompression (lznt1_decompress.cpp:122-290) ----
struct InflateState {                       // lznt1_decompress.cpp:27-34: one partial input chunk, one pending output chunk
	bool end_of_stream = false;
	uint8_t in[CHUNK + 2];
	size_t in_needed = 0, in_avail = 0;
	uint8_t out[CHUNK];
	size_t out_pos = 0, out_avail = 0;
	// Look-ahead, as on the compressing side: the complete compressed chunks that follow in the caller's input are decoded in the same
	// GPU call (when they all decode to 4096 bytes, i.e. their boundaries in the output are known) and served later if their bytes
	// are still the same.
	std::vector<uint8_t> ah_in, ah_out;
	std::vector<uint32_t> ah_off;                            // compressed chunk k = ah_in[ah_off[k], ah_off[k + 1]), its bytes = ah_out[4096 k ...)
	size_t ah_next = 0;
};
uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// one complete chunk (header included) -> its bytes, decoded on the GPU against the 4096-byte limit; any failure is DATA_ERROR (:160-166)
MSCompStatus gpu_chunk(InflateState* st, const uint8_t* chunk, size_t in_size, size_t following, uint8_t* out, size_t* out_size)
{
	if (st->ah_next + 1 < st->ah_off.size() && st->ah_off[st->ah_next + 1] - st->ah_off[st->ah_next] == in_size &&
	    memcmp(chunk, st->ah_in.data() + st->ah_off[st->ah_next], in_size) == 0) {
		memcpy(out, st->ah_out.data() + st->ah_next * CHUNK, CHUNK);
		*out_size = CHUNK; ++st->ah_next;
		return MSCOMP_OK;
	}
	st->ah_off.clear(); st->ah_next = 0;
	{	// complete, well-formed compressed chunks behind this one
		size_t k = 1, used = in_size;
		while (k < AHEAD && following - (used - in_size) >= 2) {
			const uint32_t h = rd16(chunk + used);
			const size_t sz = (h & 0x0FFFu) + 3;
			if (h == 0 || (h & 0xF000u) != 0xB000u || sz > following - (used - in_size)) { break; }
			used += sz; ++k;
		}
		if (k >= 2) {
			st->ah_out.resize(k * CHUNK);
			size_t len = k * CHUNK;
			if (lznt1_decompress(chunk, used, st->ah_out.data(), &len) == MSCOMP_OK && len == k * CHUNK) {   // every chunk gave 4096 bytes: boundaries known
				st->ah_in.assign(chunk, chunk + used);
				st->ah_off.assign(1, 0u);
				size_t o = 0;
				for (size_t i = 0; i < k; ++i) { o += (rd16(st->ah_in.data() + o) & 0x0FFFu) + 3; st->ah_off.push_back((uint32_t)o); }
				memcpy(out, st->ah_out.data(), CHUNK);
				*out_size = CHUNK; st->ah_next = 1;
				return MSCOMP_OK;
			}
		}
	}
	size_t len = CHUNK;
	const MSCompStatus r = lznt1_decompress(chunk, in_size, out, &len);
	if (r == MSCOMP_ERRNO || r == MSCOMP_MEM_ERROR) { return r; }
	if (r != MSCOMP_OK) { return MSCOMP_DATA_ERROR; }
	*out_size = len;
	return MSCOMP_OK;
}

// lznt1_decompress_chunk_read (:122-209): the chunk at `in` (*in_len bytes are there); on return *in_len = bytes of it that were used
MSCompStatus read_chunk(mscomp_stream* s, InflateState* st, const uint8_t* in, size_t* in_len)
{
	const size_t have = *in_len;                                     // (caller input: everything that is there; state buffer: the chunk only)
	const uint32_t header = rd16(in);
	if (header == 0) {                                               // :128-134
		if (s->in_avail + st->in_avail != 2) { return MSCOMP_DATA_ERROR; }
		*in_len = 2; st->end_of_stream = true;
		return MSCOMP_OK;
	}
	const size_t in_size = (header & 0x0FFFu) + 3;
	if (in_size > *in_len) {                                         // not all of the chunk is here yet (:136-143)
		if (in != st->in) { memcpy(st->in, in, *in_len); }
		st->in_needed = in_size - *in_len; st->in_avail = *in_len;
		return MSCOMP_OK;
	}
	*in_len = in_size;
	if ((header & 0x7000u) != 0x3000u) { return MSCOMP_DATA_ERROR; }   // :151
	if (header & 0x8000u) {
		size_t out_size = 0;
		if (s->out_avail < CHUNK) {                                  // through the state's buffer (:155-172)
			const MSCompStatus r = gpu_chunk(st, in, in_size, in != st->in ? have - in_size : 0, st->out, &out_size);
			if (r != MSCOMP_OK) { return r; }
			const size_t copy = out_size < s->out_avail ? out_size : s->out_avail;
			memcpy(s->out, st->out, copy);
			st->out_pos = copy; st->out_avail = out_size - copy;
			give_out(s, copy);
		} else {                                                     // straight into the caller's window (:175-188)
			const MSCompStatus r = gpu_chunk(st, in, in_size, in != st->in ? have - in_size : 0, s->out, &out_size);
			if (r != MSCOMP_OK) { return r; }
			give_out(s, out_size);
		}
	} else {                                                         // stored chunk (:190-207)
		const size_t out_size = in_size - 2;
		if (s->out_avail < out_size) {
			const size_t first = s->out_avail;
			memcpy(s->out, in + 2, first);
			memcpy(st->out, in + 2 + first, out_size - first);
			st->out_pos = 0; st->out_avail = out_size - first;
			give_out(s, first);
		} else { memcpy(s->out, in + 2, out_size); give_out(s, out_size); }
	}
	return MSCOMP_OK;
}
bool possible_end(const mscomp_stream* s, const InflateState* st)     // :223-227
{
	return (!s->in_avail || (s->in_avail == 1 && s->in[0] == 0)) && (!st->in_avail || (st->in_avail == 1 && st->in[0] == 0)) && !st->out_avail;
}

} // namespace

extern "C" {

MSCompStatus lznt1_inflate_init(mscomp_stream* s)
{
	if (!s) { return MSCOMP_ARG_ERROR; }                     // INIT_STREAM (internal.h:473-480)
	s->format = MSCOMP_LZNT1; s->compressing = false;
	s->in = nullptr; s->out = nullptr; s->in_avail = 0; s->out_avail = 0; s->in_total = 0; s->out_total = 0;
	s->error[0] = 0; s->warning[0] = 0; s->state = nullptr;
	InflateState* st = new (std::nothrow) InflateState();
	if (!st) { return MSCOMP_MEM_ERROR; }
	s->state = reinterpret_cast<mscomp_internal_state*>(st);
	return MSCOMP_OK;
}

MSCompStatus lznt1_inflate(mscomp_stream* s)
{
	InflateState* st = s ? reinterpret_cast<InflateState*>(s->state) : nullptr;
	if (!stream_ok(s, MSCOMP_LZNT1, false) || !st) { return MSCOMP_ARG_ERROR; }               // :232
	MSCompStatus r;
	if (st->out_avail) {                                             // DUMP_OUT (internal.h:493-513)
		const size_t k = st->out_avail < s->out_avail ? st->out_avail : s->out_avail;
		memcpy(s->out, st->out + st->out_pos, k);
		s->out += k; s->out_total += k;
		if (st->out_avail == k) { s->out_avail -= k; st->out_avail = 0; }
		else { s->out_avail = 0; st->out_pos += k; st->out_avail -= k; return MSCOMP_OK; }
	}
	if (st->end_of_stream) { return (s->in_avail || st->in_avail) ? MSCOMP_DATA_ERROR : MSCOMP_STREAM_END; }   // :237-241
	if (st->in_avail) {                                              // APPEND_IN (internal.h:517-533) with the body of :243-252
		for (;;) {
			const size_t k = st->in_needed < s->in_avail ? st->in_needed : s->in_avail;
			if (k) { memcpy(st->in + st->in_avail, s->in, k); st->in_avail += k; st->in_needed -= k; take_in(s, k); }
			if (st->in_needed) { return MSCOMP_OK; }
			size_t in_len = st->in_avail;
			r = read_chunk(s, st, st->in, &in_len);
			if (in_len != st->in_avail) { return MSCOMP_ARG_ERROR; }
			if (r != MSCOMP_OK) { return r; }
			if (st->end_of_stream) { return MSCOMP_STREAM_END; }
			if (st->in_needed) { continue; }                         // that was only the header: now the rest of the chunk
			break;
		}
		st->in_avail = 0;
	}
	while (s->out_avail && s->in_avail >= 2) {                       // :255-262
		// a run of complete, well-formed chunks with 4096 bytes of room for each would be decoded one by one straight into the
		// window: the same bytes come from ONE GPU batch (if it reports anything but success, go chunk by chunk to stop where the
		// reference stops)
		{
			size_t k = 0, used = 0;
			while ((k + 1) * CHUNK <= s->out_avail && s->in_avail - used >= 2) {
				const uint32_t h = rd16(s->in + used);
				const size_t sz = (h & 0x0FFFu) + 3;
				if (h == 0 || (h & 0x7000u) != 0x3000u || sz > s->in_avail - used) { break; }
				used += sz; ++k;
			}
			if (k >= 2) {
				size_t len = k * CHUNK;
				if (lznt1_decompress(s->in, used, s->out, &len) == MSCOMP_OK) { take_in(s, used); give_out(s, len); continue; }
			}
		}
		size_t in_size = s->in_avail;
		r = read_chunk(s, st, s->in, &in_size);
		if (r != MSCOMP_OK) { return r; }
		take_in(s, in_size);
		if (st->end_of_stream) { return MSCOMP_STREAM_END; }
	}
	if (s->out_avail && s->in_avail) {                               // one byte left: half a header (:264-271)
		st->in[0] = s->in[0]; st->in_needed = 1; st->in_avail = 1;
		s->in += 1; s->in_total += 1; s->in_avail = 0;
	}
	return possible_end(s, st) ? MSCOMP_POSSIBLE_STREAM_END : MSCOMP_OK;
}

MSCompStatus lznt1_inflate_end(mscomp_stream* s)
{
	InflateState* st = s ? reinterpret_cast<InflateState*>(s->state) : nullptr;
	if (!stream_ok(s, MSCOMP_LZNT1, false) || !st) { return MSCOMP_ARG_ERROR; }               // :276
	const MSCompStatus r = possible_end(s, st) ? MSCOMP_OK : MSCOMP_DATA_ERROR;                // :281-285
	delete st;
	s->state = nullptr;
	return r;
}

MSCompStatus lznt1_deflate_init(mscomp_stream* s)
{
	if (!s) { return MSCOMP_ARG_ERROR; }                     // INIT_STREAM (internal.h:473-480)
	s->format = MSCOMP_LZNT1; s->compressing = true;
	s->in = nullptr; s->out = nullptr; s->in_avail = 0; s->out_avail = 0; s->in_total = 0; s->out_total = 0;
	s->error[0] = 0; s->warning[0] = 0; s->state = nullptr;
	DeflateState* st = new (std::nothrow) DeflateState();
	if (!st) { return MSCOMP_MEM_ERROR; }
	s->state = reinterpret_cast<mscomp_internal_state*>(st);
	return MSCOMP_OK;
}

MSCompStatus lznt1_deflate(mscomp_stream* s, MSCompFlush flush)
{
	DeflateState* st = s ? reinterpret_cast<DeflateState*>(s->state) : nullptr;
	if (!stream_ok(s, MSCOMP_LZNT1) || !st || st->finished) { return MSCOMP_ARG_ERROR; }     // :149
	MSCompStatus r;
	// output of an earlier call that did not fit then (DUMP_OUT, internal.h:493-513)
	if (st->out_avail) {
		const size_t k = st->out_avail < s->out_avail ? st->out_avail : s->out_avail;
		memcpy(s->out, st->out + st->out_pos, k);
		s->out += k; s->out_total += k;
		if (st->out_avail == k) { s->out_avail -= k; st->out_avail = 0; }
		else { s->out_avail = 0; st->out_pos += k; st->out_avail -= k; return MSCOMP_OK; }
	}
	// a partial chunk kept from an earlier call (APPEND_IN, internal.h:517-533, with the body of :152-170)
	if (st->in_avail) {
		const size_t k = st->in_needed < s->in_avail ? st->in_needed : s->in_avail;
		if (k) { memcpy(st->in + st->in_avail, s->in, k); st->in_avail += k; st->in_needed -= k; take_in(s, k); }
		if (st->in_needed) {
			if (flush != MSCOMP_NO_FLUSH) {                  // cut a short chunk here
				if ((r = write_chunk(s, st, st->in, st->in_avail)) != MSCOMP_OK) { return r; }
				st->in_avail = 0; st->in_needed = 0;
				if (flush == MSCOMP_FINISH && !st->out_avail) { goto stream_end; }
			}
			return MSCOMP_OK;                                // still not a whole chunk
		}
		if ((r = write_chunk(s, st, st->in, CHUNK)) != MSCOMP_OK) { return r; }
		st->in_avail = 0;
	}
	// whole chunks while there is room (:173-177). As many as are CERTAIN to land in the window unbuffered (4098 bytes each at most)
	// are one GPU batch; the reference would have written the same bytes chunk by chunk.
	while (s->out_avail && s->in_avail >= CHUNK) {
		size_t k = s->in_avail / CHUNK;
		const size_t sure = s->out_avail / (CHUNK + 2);
		if (sure < k) { k = sure; }
		if (k == 0) {
			if ((r = write_chunk(s, st, s->in, CHUNK, s->in_avail - CHUNK)) != MSCOMP_OK) { return r; }
			take_in(s, CHUNK);
			continue;
		}
		uint8_t* img; size_t len;
		if ((r = gpu_images(st, s->in, k * CHUNK, &img, &len)) != MSCOMP_OK) { return r; }
		memcpy(s->out, img, len); give_out(s, len); take_in(s, k * CHUNK);
	}
	// the rest: less than a chunk of input, or no room (:180-195)
	if (s->out_avail && s->in_avail) {
		if (flush != MSCOMP_NO_FLUSH) {
			if ((r = write_chunk(s, st, s->in, s->in_avail)) != MSCOMP_OK) { return r; }
		} else {
			memcpy(st->in, s->in, s->in_avail);
			st->in_avail = s->in_avail; st->in_needed = CHUNK - st->in_avail;
		}
		take_in(s, s->in_avail);
	}
	if (flush == MSCOMP_FINISH && !s->in_avail && !st->in_avail && !st->out_avail) {
stream_end:
		st->finished = true;
		if (s->out_avail >= 2) { s->out[0] = 0; s->out[1] = 0; }     // End_of_buffer, not counted (:202-204)
		return MSCOMP_STREAM_END;
	}
	return MSCOMP_OK;
}

MSCompStatus lznt1_deflate_end(mscomp_stream* s)
{
	DeflateState* st = s ? reinterpret_cast<DeflateState*>(s->state) : nullptr;
	if (!stream_ok(s, MSCOMP_LZNT1) || !st) { return MSCOMP_ARG_ERROR; }                     // :211
	const MSCompStatus r = (!st->finished || s->in_avail || st->in_avail || st->out_avail) ? MSCOMP_DATA_ERROR : MSCOMP_OK;   // :216
	free(st->scratch);
	delete st;
	s->state = nullptr;
	return r;
}

// The reference's Xpress streaming COMPRESSOR is unfinished: in the default (non-_DEBUG) build xpress_deflate_init returns
// MSCOMP_MEM_ERROR without touching the stream (xpress_compress.cpp:52-73), xpress_deflate returns MSCOMP_ARG_ERROR (:74-218) and
// xpress_deflate_end checks the stream and returns MSCOMP_OK (:219-235). A program that links the drop-in and names these symbols
// must link and see the same statuses.
MSCompStatus xpress_deflate_init(mscomp_stream*) { return MSCOMP_MEM_ERROR; }
MSCompStatus xpress_deflate(mscomp_stream*, MSCompFlush) { return MSCOMP_ARG_ERROR; }
MSCompStatus xpress_deflate_end(mscomp_stream* s)
{
	return (!stream_ok(s, MSCOMP_XPRESS, true) || s->state == nullptr) ? MSCOMP_ARG_ERROR : MSCOMP_OK;
}

// The Xpress streaming DECOMPRESSOR (include/xpress.h:56-58, xpress_decompress.cpp:45-403) is not offloaded: one Xpress stream is a
// serial chain of tokens and the streaming calls hand it over a few bytes at a time -- nothing for a GPU. The three symbols exist so
// that a program naming them links against the drop-in; the init call answers MSCOMP_MEM_ERROR (the status the reference's own
// unfinished streaming entry point uses, xpress_compress.cpp:72), writes the reason into stream->error and otherwise leaves the stream untouched, so a caller falls into its error
// path at once instead of decoding nothing. A caller that needs streaming Xpress decompression keeps the reference's
// src/xpress_decompress.cpp in its link and builds this library with -DMSCOMP_AMD_NO_XPRESS_INFLATE.
#ifndef MSCOMP_AMD_NO_XPRESS_INFLATE
MSCompStatus xpress_inflate_init(mscomp_stream* s)
{
	// say why in the stream's own message field (general.h:95: `char error[256]`), so that the caller's log names the cause
	if (s) { snprintf(s->error, sizeof s->error, "libmscomp_amd: streaming Xpress decompression is not offloaded; use xpress_decompress / ms_decompress (one-shot), or link the reference's xpress_decompress.cpp and build with -DMSCOMP_AMD_NO_XPRESS_INFLATE"); }
	return MSCOMP_MEM_ERROR;
}
MSCompStatus xpress_inflate(mscomp_stream*) { return MSCOMP_ARG_ERROR; }
MSCompStatus xpress_inflate_end(mscomp_stream*) { return MSCOMP_ARG_ERROR; }
#endif

#ifndef MSCOMP_AMD_NO_FACADE
// mscomp.cpp:136-165 with its "copy" codec (:33-47,:60) for MSCOMP_NONE
MSCompStatus ms_deflate_init(MSCompFormat format, mscomp_stream* s)
{
	if (format == MSCOMP_LZNT1) { return lznt1_deflate_init(s); }
	if (format == MSCOMP_NONE) {
		if (!s) { return MSCOMP_ARG_ERROR; }
		s->format = MSCOMP_NONE; s->compressing = true;
		s->in = nullptr; s->out = nullptr; s->in_avail = 0; s->out_avail = 0; s->in_total = 0; s->out_total = 0;
		s->error[0] = 0; s->warning[0] = 0; s->state = nullptr;
		return MSCOMP_OK;
	}
	if (format == MSCOMP_XPRESS) { return xpress_deflate_init(s); }
	return MSCOMP_ARG_ERROR;                                  // Xpress+Huffman: no entry in the reference's table (mscomp.cpp:141,147)
}
MSCompStatus ms_deflate(mscomp_stream* s, MSCompFlush flush)
{
	if (!s) { return MSCOMP_ARG_ERROR; }
	if (s->format == MSCOMP_LZNT1) { return lznt1_deflate(s, flush); }
	if (s->format == MSCOMP_NONE) {
		if (!stream_ok(s, MSCOMP_NONE)) { return MSCOMP_ARG_ERROR; }
		const size_t n = s->in_avail < s->out_avail ? s->in_avail : s->out_avail;
		if (n) { memcpy(s->out, s->in, n); }
		give_out(s, n); take_in(s, n);
		return (flush == MSCOMP_FINISH && !s->in_avail) ? MSCOMP_STREAM_END : MSCOMP_OK;
	}
	if (s->format == MSCOMP_XPRESS) { return xpress_deflate(s, flush); }
	return MSCOMP_ARG_ERROR;
}
MSCompStatus ms_deflate_end(mscomp_stream* s)
{
	if (!s) { return MSCOMP_ARG_ERROR; }
	if (s->format == MSCOMP_LZNT1) { return lznt1_deflate_end(s); }
	if (s->format == MSCOMP_NONE) { return stream_ok(s, MSCOMP_NONE) ? MSCOMP_OK : MSCOMP_ARG_ERROR; }
	if (s->format == MSCOMP_XPRESS) { return xpress_deflate_end(s); }
	return MSCOMP_ARG_ERROR;
}
// mscomp.cpp:167-196; the copy codec initialises and checks its inflate streams as COMPRESSING ones (mscomp.cpp:33,48-59)
MSCompStatus ms_inflate_init(MSCompFormat format, mscomp_stream* s)
{
	if (format == MSCOMP_LZNT1) { return lznt1_inflate_init(s); }
	if (format == MSCOMP_NONE) { return ms_deflate_init(MSCOMP_NONE, s); }
#ifndef MSCOMP_AMD_NO_XPRESS_INFLATE
	if (format == MSCOMP_XPRESS) { return xpress_inflate_init(s); }   // not offloaded: MSCOMP_MEM_ERROR (see above)
#endif
	return MSCOMP_ARG_ERROR;                                  // Xpress+Huffman: none in the reference either (mscomp.cpp:172,178)
}
MSCompStatus ms_inflate(mscomp_stream* s)
{
	if (!s) { return MSCOMP_ARG_ERROR; }
	if (s->format == MSCOMP_LZNT1) { return lznt1_inflate(s); }
	if (s->format == MSCOMP_NONE) {
		if (!stream_ok(s, MSCOMP_NONE)) { return MSCOMP_ARG_ERROR; }
		const size_t n = s->in_avail < s->out_avail ? s->in_avail : s->out_avail;
		if (n) { memcpy(s->out, s->in, n); }
		give_out(s, n); take_in(s, n);
		return MSCOMP_OK;
	}
	return MSCOMP_ARG_ERROR;
}
MSCompStatus ms_inflate_end(mscomp_stream* s)
{
	if (!s) { return MSCOMP_ARG_ERROR; }
	if (s->format == MSCOMP_LZNT1) { return lznt1_inflate_end(s); }
	if (s->format == MSCOMP_NONE) { return stream_ok(s, MSCOMP_NONE) ? MSCOMP_OK : MSCOMP_ARG_ERROR; }
	return MSCOMP_ARG_ERROR;
}
#endif

} // extern "C"

// hostbatch.hip -- mscomp_amd_compress_units_host: n independent units given by HOST pointers, compressed on one or several GPUs of this
// process (SURVEY.md 8e "Multi-GPU" + 8f-3 "host pipeline"): the caller-facing form of the per-GPU shard that bench.py runs one process per
// GPU. Host orchestration only -- every output byte comes from the HIP kernels behind the batch interface of api.hip.
//
// The reference has no counterpart: ms_compress (/root/reference/src/mscomp.cpp:113-117) is one buffer per call on one thread. Each unit
// here is exactly one such call (same bytes, same MSCOMP_OK / MSCOMP_BUF_ERROR, the uncounted LZNT1 00 00 behind the stream when the
// capacity has room: lznt1_compress.cpp:270).
//
//   * the units are cut into n_dev contiguous ranges with near-equal input bytes (the rule of ms_compress_amd/sharding.py shard_ranges;
//     chunks and files are independent, so there is no exchange step and no collective);
//   * one host thread per range; the range runs as sub-batches of at most MSCOMP_AMD_HOST_BATCH_MB (default 32) MiB of input, up to 8 in flight,
//     each in its own slot (context + stream + device buffers, kept in a per-device pool between calls): an uploader thread, the range's
//     thread (plans and launches) and a downloader thread move them along, so uploads, the kernels of several sub-batches and downloads
//     overlap (run_range below);
//   * runs of units that lie back to back in
//     the caller's memory travel as ONE copy each way (a file cut into 64 KiB units is one upload; outputs laid out capacity after capacity
//     come back as one download of the range that holds streams), other units one copy each.
//
// This file is compiled WITH C++ exceptions (csrc/Makefile; the rest of the library is -fno-exceptions): it builds std::vector tables and
// starts std::threads inside extern "C" entries, and an allocation or thread-creation failure there must come back as MSCOMP_MEM_ERROR,
// not end the caller's process. Every thread body and both entries catch; threads that did start are always joined.
#include "../../include/mscomp_amd.h"
#include <hip/hip_runtime.h>
#include <new>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

namespace msc {       // api.hip: registrations (page-locks) other threads' large one-shot calls hold on caller memory
void* host_pins_hold(const void* ptr, size_t n);
void host_pins_release(void* token);
}

namespace {

#define HB_SLOTS 8                                        // sub-batches in flight per device range

// One sub-batch in flight: its own context (stream `ex`, scratch), device buffers and pinned result words. Kept between calls.
struct Slot {
	mscomp_amd_ctx* ctx = nullptr;
	hipStream_t ex = nullptr;
	hipEvent_t ev_ex = nullptr;
	void* d_in = nullptr; void* d_out = nullptr; void* d_meta = nullptr;
	size_t in_cap = 0, out_cap = 0, meta_cap = 0;
	uint64_t* h_meta = nullptr; size_t h_cap = 0;          // pinned: out_len (u64) x n | status (i32) x n
	bool init(int dev)
	{
		if (hipStreamCreateWithFlags(&ex, hipStreamNonBlocking) != hipSuccess) { return false; }
		if (hipEventCreateWithFlags(&ev_ex, hipEventDisableTiming) != hipSuccess) { return false; }
		return mscomp_amd_ctx_create(dev, ex, &ctx) == MSCOMP_OK;
	}
	static bool grow(void** p, size_t* cap, size_t need)
	{
		if (need <= *cap) { return true; }
		if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
		const size_t want = (need + need / 8 + 256 + 3) & ~(size_t)3;          // (a multiple of 4: the word-wise clears cover the whole buffer)
		if (hipMalloc(p, want) != hipSuccess) { *p = nullptr; (void)hipGetLastError(); return false; }
		*cap = want; return true;
	}
	// The mirror-layout download copies the slack between the streams of a batch to the caller too: the output buffer of a slot is all zeros
	// wherever no stream of the current batch lies (cleared when it is allocated and again, off the critical path, when a batch has left it).
	bool reserve_out(size_t need)
	{
		if (need <= out_cap) { return true; }
		if (!grow(&d_out, &out_cap, need)) { return false; }
		return hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_out), 0, out_cap / 4, ex) == hipSuccess;
	}
	bool reserve(size_t in_b, size_t out_b, size_t n)
	{
		if (!grow(&d_in, &in_cap, in_b + 64) || !reserve_out(out_b + 64) || !grow(&d_meta, &meta_cap, n * 16 + 64)) { return false; }
		if (h_cap < n) {
			if (h_meta) { (void)hipHostFree(h_meta); h_meta = nullptr; h_cap = 0; }
			void* q = nullptr;
			if (hipHostMalloc(&q, (n + n / 4 + 64) * 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
			h_meta = static_cast<uint64_t*>(q); h_cap = n + n / 4 + 64;
		}
		return true;
	}
	void destroy()
	{
		if (ctx) { mscomp_amd_ctx_destroy(ctx); ctx = nullptr; }
		if (d_in) { (void)hipFree(d_in); } if (d_out) { (void)hipFree(d_out); } if (d_meta) { (void)hipFree(d_meta); }
		if (h_meta) { (void)hipHostFree(h_meta); }
		if (ev_ex) { (void)hipEventDestroy(ev_ex); }
		if (ex) { (void)hipStreamDestroy(ex); }
	}
};

struct Worker {                                           // everything one device range needs, kept between calls
	int device = -1;
	hipStream_t up = nullptr, dn = nullptr;                  // all uploads / all downloads of the range (one thread each issues them in order)
	Slot slot[HB_SLOTS]; bool have[HB_SLOTS] = {};
	bool init(int dev)
	{
		device = dev;
		return hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&up, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&dn, hipStreamNonBlocking) == hipSuccess;
	}
	Slot* get(int s)                                         // slots come to life when a job is large enough to reach them
	{
		if (!have[s]) { if (!slot[s].init(device)) { slot[s].destroy(); slot[s] = Slot(); return nullptr; } have[s] = true; }
		return &slot[s];
	}
	void destroy()
	{
		if (device >= 0) { (void)hipSetDevice(device); }
		for (int s = 0; s < HB_SLOTS; ++s) { if (have[s]) { slot[s].destroy(); have[s] = false; } }
		if (up) { (void)hipStreamDestroy(up); up = nullptr; } if (dn) { (void)hipStreamDestroy(dn); dn = nullptr; }
	}
};

// idle workers by device: a call takes one per range (two ranges on the same device get two workers) and gives them back
std::mutex g_pool_mu;
std::vector<Worker*> g_pool;
Worker* take_worker(int dev)
{
	{
		std::lock_guard<std::mutex> lk(g_pool_mu);
		for (size_t i = 0; i < g_pool.size(); ++i) { if (g_pool[i]->device == dev) { Worker* w = g_pool[i]; g_pool.erase(g_pool.begin() + (long)i); return w; } }
	}
	Worker* w = new (std::nothrow) Worker();
	if (w && !w->init(dev)) { w->destroy(); delete w; w = nullptr; }
	return w;
}
void give_worker(Worker* w) { std::lock_guard<std::mutex> lk(g_pool_mu); g_pool.push_back(w); }

struct Job {
	MSCompFormat format; bool decompress; size_t n;
	const uint8_t* const* in_ptrs; const size_t* in_lens; uint8_t* const* out_ptrs; const size_t* out_caps; size_t* out_lens; MSCompStatus* statuses;
};
struct Batch { size_t b0, b1; std::vector<uint64_t> in_off, in_len, out_off, out_cap; uint64_t in_total, out_total; bool out_mirror; mscomp_amd_plan* plan; };

// input bytes per sub-batch: MSCOMP_AMD_HOST_BATCH_MB, else 32 MiB -- 96 MiB for Xpress+Huffman compression, whose one-wave-per-chunk stages
// (parse, Huffman, encode) are latency-bound in small batches: many small batches side by side cost 13 ms of GPU time for the 212 MB corpus
// where one batch costs 8.6
size_t batch_bytes(MSCompFormat format, bool decompress)
{
	static const long env = [] { const char* e = getenv("MSCOMP_AMD_HOST_BATCH_MB"); const long x = e ? atol(e) : 0; return (x >= 1 && x <= 65536) ? x : 0L; }();
	if (env) { return (size_t)env << 20; }
	return (size_t)((format == MSCOMP_XPRESS_HUFF && !decompress) ? 96 : 32) << 20;
}

// units [u0, u1) on worker w: MSCOMP_OK, or the first error that stopped the range (HIP failure / out of memory / bad argument).
//
// The range is cut into sub-batches of about MSCOMP_AMD_HOST_BATCH_MB (default 32) MiB of input, up to HB_SLOTS of them in flight, each in
// its own slot (context + stream + buffers), and three host threads move them along:
//   uploader   : sub-batch i's inputs host -> device (copies from pageable memory hold the calling thread, so they get their own);
//   launcher   : this thread -- plan + kernels of sub-batch i as soon as its input is up, on the slot's stream: the kernels of the sub-batches
//                in flight overlap on the GPU (the serial stages of a small batch are latency-bound: one wave per unit / chunk);
//   downloader : results + streams of sub-batch i device -> host when its kernels are through, then the slot is free again.
// A sub-batch's download therefore runs under the kernels of the ones behind it and its upload under those in front of it.
MSCompStatus run_range(Worker* w, const Job& j, size_t u0, size_t u1, bool on_calling_thread)
{
	if (hipSetDevice(w->device) != hipSuccess) { return MSCOMP_ERRNO; }
	// ONE large LZNT1 unit: the one-shot call has the better path for it (the caller's buffers mapped, one launch reading and writing them over
	// PCIe: 2.1 ms for 51 MB against 3.0 ms through a staged sub-batch)
	// Only on the CALLING thread (range 0): the one-shot call keeps its context, streams and staging in thread-local storage, which a
	// range thread started for this call would build and tear down again (ms-scale, with device-wide waits that stall the other ranges).
	// MSCOMP_AMD_HOST_BATCH_MB / MSCOMP_AMD_HOST_SLOTS do not apply on this path.
	if (on_calling_thread && u1 - u0 == 1 && !j.decompress && j.format == MSCOMP_LZNT1 && j.in_lens[u0] >= ((size_t)36 << 20)) {
		size_t len = j.out_caps[u0];
		const MSCompStatus r = lznt1_compress(j.in_ptrs[u0], j.in_lens[u0], j.out_ptrs[u0], &len);   // (the codec's own entry: exists in every build flavour of the library)
		j.statuses[u0] = r; j.out_lens[u0] = r == MSCOMP_OK ? len : 0;
		return (r == MSCOMP_OK || r == MSCOMP_BUF_ERROR) ? MSCOMP_OK : r;
	}
	const size_t limit = batch_bytes(j.format, j.decompress);
	std::vector<Batch> batches;
	for (size_t i = u0; i < u1;) {                               // sub-batches: runs of units up to `limit` input bytes (a larger unit is one of its own)
		Batch b; b.b0 = i; b.in_total = 0; b.out_total = 0; b.plan = nullptr; b.out_mirror = true;
		uint64_t in_pos = 0;
		while (i < u1 && (i == b.b0 || (in_pos + j.in_lens[i] <= limit && i - b.b0 < (1u << 22)))) {
			b.in_off.push_back(in_pos); b.in_len.push_back(j.in_lens[i]);
			in_pos += (j.in_lens[i] + 15u) & ~(uint64_t)15u;
			if (i > b.b0 && j.out_ptrs[i] != j.out_ptrs[i - 1] + j.out_caps[i - 1]) { b.out_mirror = false; }
			++i;
		}
		b.b1 = i; b.in_total = in_pos;
		uint64_t out_pos = 0;                                      // outputs: the caller's layout when it is capacity after capacity (one download), else 16-byte aligned starts
		// (a compressor's device capacity is what the format can produce, never a generous caller capacity -- *out_len = 1 << 40 is legal in the
		// reference and must not become an allocation; a unit that fits that bound gets the status and bytes it would get with any larger capacity)
		std::vector<uint64_t> dcap(b.b1 - b.b0);
		for (size_t k = b.b0; k < b.b1; ++k) {
			uint64_t c = j.out_caps[k];
			uint64_t most;
			if (!j.decompress) { most = (uint64_t)ms_max_compressed_size(j.format, j.in_lens[k]) + 2u; }
			else if (j.format == MSCOMP_LZNT1) { most = ((uint64_t)j.in_lens[k] / 3u + 1u) * 4096u; }            // a chunk takes at least 3 bytes and gives at most 4096
			else { most = (uint64_t)j.in_lens[k] * 16u + (1u << 20); }     // Xpress formats have no bound (a 10-byte token can stand for 4 GiB): 16 x the input first; a unit that
			                                                              // answers MSCOMP_BUF_ERROR below the caller's capacity is decoded again, alone, with room (after the pipeline)
			if (most < c) { c = most; b.out_mirror = false; }
			dcap[k - b.b0] = c;
		}
		for (size_t k = b.b0; k < b.b1; ++k) { const uint64_t c = dcap[k - b.b0]; b.out_off.push_back(out_pos); b.out_cap.push_back(c); out_pos += b.out_mirror ? c : ((c + 15u) & ~(uint64_t)15u); }
		b.out_total = out_pos;
		batches.push_back(std::move(b));
	}
	const size_t nb = batches.size();
	if (nb == 0) { return MSCOMP_OK; }
	// the sub-batch with the largest UNIT first: the kernels of a large file (Xpress+Huffman: its chunks form one chain of serial stages) are the
	// critical path of the range, everything else runs beside them (stable: equal sub-batches keep the caller's order)
	if (!(getenv("MSCOMP_AMD_HOST_ORDER") && *getenv("MSCOMP_AMD_HOST_ORDER") == '0')) {
		std::vector<uint64_t> big(nb, 0);
		for (size_t k = 0; k < nb; ++k) { for (uint64_t l : batches[k].in_len) { if (l > big[k]) { big[k] = l; } } }
		std::vector<size_t> ord(nb);
		for (size_t k = 0; k < nb; ++k) { ord[k] = k; }
		std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return big[a] > big[b]; });
		std::vector<Batch> sorted; sorted.reserve(nb);
		for (size_t k = 0; k < nb; ++k) { sorted.push_back(std::move(batches[ord[k]])); }
		batches.swap(sorted);
	}
	static const size_t nslots = [] { const char* e = getenv("MSCOMP_AMD_HOST_SLOTS"); const long x = e ? atol(e) : HB_SLOTS; return (size_t)(x >= 1 && x <= HB_SLOTS ? x : HB_SLOTS); }();   // sub-batches in flight (212 MB corpus, 32 MiB sub-batches: 2 / 4 / 8 slots -> Xpress 11.7 / 9.3 / 8.5 ms, Xpress+Huffman 23.9 / 16.4 / 14.2 ms)

	std::mutex mu; std::condition_variable cv;
	std::vector<char> uploaded(nb, 0), launched(nb, 0);
	static const bool trace = getenv("MSCOMP_AMD_HOST_TRACE") != nullptr;   // dev: a time line of the sub-batches on stderr
	const auto t_zero = std::chrono::steady_clock::now();
	auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_zero).count(); };
	std::vector<double> tl(trace ? nb * 6 : 0, 0.0);                      // upload begin / end, launch begin / end, kernels done, download end
	bool busy[HB_SLOTS] = {};
	MSCompStatus failed = MSCOMP_OK;                             // first error (under mu); every loop stops when it is set
	std::vector<size_t> again;                                   // decompression: units whose DEVICE capacity (not the caller's) was too small (downloader thread only)
	auto fail = [&](MSCompStatus r) { std::lock_guard<std::mutex> lk(mu); if (failed == MSCOMP_OK) { failed = r; } cv.notify_all(); };
	auto is_failed = [&]() { std::lock_guard<std::mutex> lk(mu); return failed != MSCOMP_OK; };

	auto up_body = [&] {
		if (hipSetDevice(w->device) != hipSuccess) { fail(MSCOMP_ERRNO); return; }
		for (size_t k = 0; k < nb; ++k) {
			const int s = (int)(k % nslots);
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !busy[s] || failed != MSCOMP_OK; }); if (failed != MSCOMP_OK) { return; } busy[s] = true; }
			Batch& b = batches[k];
			const size_t n = b.b1 - b.b0;
			if (trace) { tl[k * 6] = now_ms(); }
			Slot* sl = w->get(s);
			if (!sl) { fail(MSCOMP_ERRNO); return; }
			if (!sl->reserve((size_t)b.in_total, (size_t)b.out_total, n)) { fail(MSCOMP_MEM_ERROR); return; }
			uint8_t* d_in = static_cast<uint8_t*>(sl->d_in);
			// (another thread's large one-shot call may have page-locked some of these inputs: its registration must outlive our copies. RAII: the
			// references go back on EVERY way out of this iteration, a std::bad_alloc from push_back included -- ADVICE r05)
			struct Held { std::vector<void*> v; ~Held() { drop(); } void drop() { for (void* t : v) { msc::host_pins_release(t); } v.clear(); } } held;
			held.v.reserve(n);
			for (size_t i = 0; i < n;) {                               // units that lie back to back in the caller's memory AND on the device go as one copy
				size_t e = i + 1;
				uint64_t bytes = b.in_len[i];
				while (e < n && (b.in_len[e - 1] & 15u) == 0 && j.in_ptrs[b.b0 + e] == j.in_ptrs[b.b0 + e - 1] + b.in_len[e - 1]) { bytes += b.in_len[e]; ++e; }
				if (bytes) { void* t = msc::host_pins_hold(j.in_ptrs[b.b0 + i], (size_t)bytes); if (t) { held.v.push_back(t); } }
				if (bytes && hipMemcpyAsync(d_in + b.in_off[i], j.in_ptrs[b.b0 + i], (size_t)bytes, hipMemcpyHostToDevice, w->up) != hipSuccess) { fail(MSCOMP_ERRNO); return; }
				i = e;
			}
			const bool up_ok = hipStreamSynchronize(w->up) == hipSuccess;
			held.drop();
			if (!up_ok) { fail(MSCOMP_ERRNO); return; }
			if (trace) { tl[k * 6 + 1] = now_ms(); }
			{ std::lock_guard<std::mutex> lk(mu); uploaded[k] = 1; }
			cv.notify_all();
		}
	};

	auto dn_body = [&] {
		if (hipSetDevice(w->device) != hipSuccess) { fail(MSCOMP_ERRNO); return; }
		for (size_t k = 0; k < nb; ++k) {
			const int s = (int)(k % nslots);
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return launched[k] || failed != MSCOMP_OK; }); if (!launched[k]) { return; } }
			Batch& b = batches[k];
			Slot* sl = &w->slot[s];
			if (hipEventSynchronize(sl->ev_ex) != hipSuccess) { fail(MSCOMP_ERRNO); return; }
			if (trace) { tl[k * 6 + 4] = now_ms(); }
			const size_t n = b.b1 - b.b0;
			const uint64_t* h_len = sl->h_meta; const int32_t* h_st = reinterpret_cast<const int32_t*>(h_len + n);
			const uint8_t* d_out = static_cast<const uint8_t*>(sl->d_out);
			size_t last_ok = n;                                        // mirror layout: one copy up to the end of the last stream
			for (size_t i = 0; i < n; ++i) { if (h_st[i] == MSCOMP_OK) { last_ok = i; } }
			auto span = [&](size_t i) -> size_t {                      // bytes of unit i to bring home: the stream, and for LZNT1 the two uncounted 00 00 behind it
				size_t len = (size_t)h_len[i];
				if (!j.decompress && j.format == MSCOMP_LZNT1 && j.out_caps[b.b0 + i] - len >= 2) { len += 2; }
				return len;
			};
			bool ok = true;
			// a few large units whose streams are much shorter than their capacities (whole files): one copy per stream moves fewer bytes
			// than the one copy of the capacity span
			bool mirror = b.out_mirror;
			if (mirror && last_ok != n && n <= 64) {
				size_t streams = 0;
				for (size_t i = 0; i < n; ++i) { if (h_st[i] == MSCOMP_OK) { streams += span(i); } }
				if (streams + streams / 4 < (size_t)b.out_off[last_ok] + span(last_ok)) { mirror = false; }
			}
			if (mirror && last_ok != n) {
				const size_t bytes = (size_t)b.out_off[last_ok] + span(last_ok);
				if (bytes && hipMemcpyAsync(j.out_ptrs[b.b0], d_out, bytes, hipMemcpyDeviceToHost, w->dn) != hipSuccess) { ok = false; }
			} else if (!mirror) {
				for (size_t i = 0; i < n && ok; ++i) {
					if (h_st[i] != MSCOMP_OK) { continue; }
					const size_t bytes = span(i);
					if (bytes && hipMemcpyAsync(j.out_ptrs[b.b0 + i], d_out + b.out_off[i], bytes, hipMemcpyDeviceToHost, w->dn) != hipSuccess) { ok = false; }
				}
			}
			if (!ok || hipStreamSynchronize(w->dn) != hipSuccess) { fail(MSCOMP_ERRNO); return; }
			if (trace) { tl[k * 6 + 5] = now_ms(); }
			for (size_t i = 0; i < n; ++i) {                           // results to the caller once the bytes are home
				const size_t u = b.b0 + i;
				j.statuses[u] = (MSCompStatus)h_st[i];
				j.out_lens[u] = h_st[i] == MSCOMP_OK ? (size_t)h_len[i] : 0;
				if (j.decompress && h_st[i] == MSCOMP_BUF_ERROR && b.out_cap[i] < j.out_caps[u]) { again.push_back(u); }
			}
			mscomp_amd_plan_destroy(b.plan); b.plan = nullptr;         // (its tables go back to the slot's context: no hipFree on this path)
			// zeros again where this batch's streams were (stream-ordered in front of the slot's next kernels, under the other slots' work)
			// (the kernels may write into the 64 bytes of slack behind the last stream: they are cleared too, so that a later, larger batch in this slot
			// never carries another call's bytes into the caller's inter-stream slack)
			const size_t clr = std::min(sl->out_cap, (size_t)b.out_total + 64u);
			if (clr && hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(sl->d_out), 0, (clr + 3) / 4, sl->ex) != hipSuccess) { fail(MSCOMP_ERRNO); return; }
			{ std::lock_guard<std::mutex> lk(mu); busy[s] = false; }
			cv.notify_all();
		}
	};

	// (a thread body must not let an exception out: std::bad_alloc from a table that grows becomes the range's MSCOMP_MEM_ERROR)
	auto up_fn = [&] { try { up_body(); } catch (const std::bad_alloc&) { fail(MSCOMP_MEM_ERROR); } catch (...) { fail(MSCOMP_ERRNO); } };
	auto dn_fn = [&] { try { dn_body(); } catch (const std::bad_alloc&) { fail(MSCOMP_MEM_ERROR); } catch (...) { fail(MSCOMP_ERRNO); } };
	// one sub-batch: the three stages one after the other on this thread (nothing to overlap, no threads to start)
	const bool inline_stages = nb == 1;
	std::thread uploader, downloader;
	if (inline_stages) { up_fn(); }
	else {
		try { uploader = std::thread(up_fn); downloader = std::thread(dn_fn); }
		catch (...) { fail(MSCOMP_MEM_ERROR); }                     // the system refused a thread: whichever started sees `failed` and returns; both are joined below
	}
	try {
	for (size_t k = 0; k < nb; ++k) {                            // launcher
		const int s = (int)(k % nslots);
		{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return uploaded[k] || failed != MSCOMP_OK; }); if (!uploaded[k]) { break; } }
		Batch& b = batches[k];
		Slot* sl = &w->slot[s];
		const size_t n = b.b1 - b.b0;
		if (trace) { tl[k * 6 + 2] = now_ms(); }
		MSCompStatus rs = j.decompress ? mscomp_amd_plan_create_decompress(sl->ctx, j.format, n, b.in_off.data(), b.in_len.data(), b.out_off.data(), b.out_cap.data(), &b.plan)
		                               : mscomp_amd_plan_create(sl->ctx, j.format, n, b.in_off.data(), b.in_len.data(), b.out_off.data(), b.out_cap.data(), &b.plan);
		uint64_t* d_len = static_cast<uint64_t*>(sl->d_meta); int32_t* d_st = reinterpret_cast<int32_t*>(d_len + n);
		if (rs == MSCOMP_OK) { rs = mscomp_amd_plan_execute(b.plan, static_cast<uint8_t*>(sl->d_in), static_cast<uint8_t*>(sl->d_out), d_len, d_st); }
		if (rs == MSCOMP_OK && (hipMemcpyAsync(sl->h_meta, d_len, n * 12, hipMemcpyDeviceToHost, sl->ex) != hipSuccess || hipEventRecord(sl->ev_ex, sl->ex) != hipSuccess)) { rs = MSCOMP_ERRNO; }
		if (rs != MSCOMP_OK) { fail(rs); break; }
		if (trace) { tl[k * 6 + 3] = now_ms(); }
		{ std::lock_guard<std::mutex> lk(mu); launched[k] = 1; }
		cv.notify_all();
	}
	} catch (const std::bad_alloc&) { fail(MSCOMP_MEM_ERROR); } catch (...) { fail(MSCOMP_ERRNO); }
	if (inline_stages) { dn_fn(); } else { if (uploader.joinable()) { uploader.join(); } if (downloader.joinable()) { downloader.join(); } }
	(void)hipStreamSynchronize(w->up); (void)hipStreamSynchronize(w->dn);
	for (int s = 0; s < HB_SLOTS; ++s) { if (w->have[s]) { (void)hipStreamSynchronize(w->slot[s].ex); } }
	for (auto& b : batches) { if (b.plan) { mscomp_amd_plan_destroy(b.plan); b.plan = nullptr; } }
	(void)hipGetLastError();
	(void)is_failed;
	if (failed == MSCOMP_OK) {                                   // the few units that need more room than 16 x their input: the one-shot decoder (it grows its staging: api.hip one_shot)
		for (size_t u : again) {
			size_t len = j.out_caps[u];
			const MSCompStatus r = j.format == MSCOMP_LZNT1 ? lznt1_decompress(j.in_ptrs[u], j.in_lens[u], j.out_ptrs[u], &len)
			                     : j.format == MSCOMP_XPRESS ? xpress_decompress(j.in_ptrs[u], j.in_lens[u], j.out_ptrs[u], &len)
			                                                 : xpress_huff_decompress(j.in_ptrs[u], j.in_lens[u], j.out_ptrs[u], &len);
			j.statuses[u] = r; j.out_lens[u] = r == MSCOMP_OK ? len : 0;
		}
	}
	if (trace) {
		for (size_t k = 0; k < nb; ++k) {
			fprintf(stderr, "hostbatch %2zu: %8.0f KB in | up %7.3f-%7.3f | launch %7.3f-%7.3f | kernels done %7.3f | down %7.3f\n", k, (double)batches[k].in_total / 1024.0,
			        tl[k * 6], tl[k * 6 + 1], tl[k * 6 + 2], tl[k * 6 + 3], tl[k * 6 + 4], tl[k * 6 + 5]);
		}
		fprintf(stderr, "hostbatch total %.3f ms\n", now_ms());
	}
	return failed;
}

} // namespace

extern "C" {

static MSCompStatus units_host(MSCompFormat format, bool decompress, int n_dev, const int* devices, size_t n_units,
                               const uint8_t* const* in_ptrs, const size_t* in_lens, uint8_t* const* out_ptrs, const size_t* out_caps,
                               size_t* out_lens, MSCompStatus* statuses)
{
	if (format != MSCOMP_LZNT1 && format != MSCOMP_XPRESS && format != MSCOMP_XPRESS_HUFF) { return MSCOMP_ARG_ERROR; }      // mscomp.cpp:115
	if (n_dev < 1 || n_dev > 64 || (n_units && (!in_ptrs || !in_lens || !out_ptrs || !out_caps || !out_lens || !statuses))) { return MSCOMP_ARG_ERROR; }
	for (size_t i = 0; i < n_units; ++i) { if ((in_lens[i] && !in_ptrs[i]) || (out_caps[i] && !out_ptrs[i])) { return MSCOMP_ARG_ERROR; } statuses[i] = MSCOMP_ERRNO; out_lens[i] = 0; }
	int prev = 0;
	if (hipGetDevice(&prev) != hipSuccess) { return MSCOMP_ERRNO; }             // no GPU / no HIP runtime: fail loudly, there is no CPU encoder here
	try {
	int ndev_sys = 0;
	if (hipGetDeviceCount(&ndev_sys) != hipSuccess) { return MSCOMP_ERRNO; }
	std::vector<int> dev(n_dev);
	for (int r = 0; r < n_dev; ++r) { dev[r] = devices ? devices[r] : r; if (dev[r] < 0 || dev[r] >= ndev_sys) { return MSCOMP_ARG_ERROR; } }
	// contiguous ranges with near-equal input bytes: the boundary closest to r / n_dev of the total (sharding.shard_ranges)
	std::vector<uint64_t> cum(n_units + 1, 0);
	for (size_t i = 0; i < n_units; ++i) { cum[i + 1] = cum[i] + in_lens[i]; }
	std::vector<size_t> cuts(n_dev + 1, 0);
	for (int r = 1; r < n_dev; ++r) {
		const long double target = (long double)cum[n_units] * r / n_dev;
		size_t lo = 0, hi = n_units;                                 // first k with cum[k] >= target
		while (lo < hi) { const size_t mid = (lo + hi) / 2; if ((long double)cum[mid] < target) { lo = mid + 1; } else { hi = mid; } }
		size_t k = lo;
		if (k > 0) { const long double a = target - (long double)cum[k - 1], b = (long double)cum[k < n_units ? k : n_units] - target; if ((a < 0 ? -a : a) <= (b < 0 ? -b : b)) { --k; } }
		if (k < cuts[r - 1]) { k = cuts[r - 1]; }
		cuts[r] = k < n_units ? k : n_units;
	}
	cuts[n_dev] = n_units;
	const Job job = { format, decompress, n_units, in_ptrs, in_lens, out_ptrs, out_caps, out_lens, statuses };
	std::vector<MSCompStatus> res(n_dev, MSCOMP_OK);
	std::vector<std::thread> threads;
	threads.reserve((size_t)n_dev);
	struct JoinAll { std::vector<std::thread>& t; ~JoinAll() { for (auto& x : t) { if (x.joinable()) { x.join(); } } } } join_all{threads};   // also when an exception unwinds this frame
	auto body = [&](int r) {
		try {
			if (cuts[r] == cuts[r + 1]) { return; }
			Worker* w = take_worker(dev[r]);
			if (!w) { res[r] = MSCOMP_ERRNO; return; }
			try { res[r] = run_range(w, job, cuts[r], cuts[r + 1], r == 0); }
			catch (const std::bad_alloc&) { res[r] = MSCOMP_MEM_ERROR; } catch (...) { res[r] = MSCOMP_ERRNO; }
			if (res[r] == MSCOMP_OK) { give_worker(w); } else { w->destroy(); delete w; }   // (a range that failed may leave a sticky HIP error or half-grown buffers behind: its worker is not reused)
		} catch (const std::bad_alloc&) { res[r] = MSCOMP_MEM_ERROR; } catch (...) { res[r] = MSCOMP_ERRNO; }
	};
	int started = 1;
	try { for (int r = 1; r < n_dev; ++r) { threads.emplace_back(body, r); ++started; } }
	catch (...) { for (int r = started; r < n_dev; ++r) { res[r] = MSCOMP_MEM_ERROR; } }   // the system refused a thread: the ranges that have none report it, the others finish
	body(0);                                                         // range 0 on the calling thread
	for (auto& t : threads) { t.join(); }
	(void)hipSetDevice(prev);
	for (int r = 0; r < n_dev; ++r) { if (res[r] != MSCOMP_OK) { return res[r]; } }
	return MSCOMP_OK;
	} catch (const std::bad_alloc&) { (void)hipSetDevice(prev); return MSCOMP_MEM_ERROR; }
	catch (...) { (void)hipSetDevice(prev); return MSCOMP_ERRNO; }
}

MSCompStatus mscomp_amd_compress_units_host(MSCompFormat format, int n_dev, const int* devices, size_t n_units,
                                            const uint8_t* const* in_ptrs, const size_t* in_lens, uint8_t* const* out_ptrs, const size_t* out_caps,
                                            size_t* out_lens, MSCompStatus* statuses)
{
	return units_host(format, false, n_dev, devices, n_units, in_ptrs, in_lens, out_ptrs, out_caps, out_lens, statuses);
}

// the same for the decoders: unit i is one ms_decompress(format, in_ptrs[i], in_lens[i], out_ptrs[i], &out_lens[i]) call with *out_len = out_caps[i]
// (mscomp.h:79): statuses[i] = MSCOMP_OK / MSCOMP_BUF_ERROR / MSCOMP_DATA_ERROR as the reference's one-shot decoder returns them
MSCompStatus mscomp_amd_decompress_units_host(MSCompFormat format, int n_dev, const int* devices, size_t n_units,
                                              const uint8_t* const* in_ptrs, const size_t* in_lens, uint8_t* const* out_ptrs, const size_t* out_caps,
                                              size_t* out_lens, MSCompStatus* statuses)
{
	return units_host(format, true, n_dev, devices, n_units, in_ptrs, in_lens, out_ptrs, out_caps, out_lens, statuses);
}

// releases the pooled workers (contexts, streams, staging) of every device; safe while no call is running
void mscomp_amd_host_pool_release(void)
{
	std::vector<Worker*> all;
	{ std::lock_guard<std::mutex> lk(g_pool_mu); all.swap(g_pool); }
	int prev = 0; (void)hipGetDevice(&prev);
	for (Worker* w : all) { w->destroy(); delete w; }
	(void)hipSetDevice(prev);
}

} // extern "C"

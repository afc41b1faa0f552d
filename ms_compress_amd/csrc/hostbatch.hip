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
//   * one host thread per range: its own context (kept in a per-device pool between calls: the scratch stays allocated), three streams
//     (uploads / kernels / downloads) and two sets of device buffers, so that batch k + 1 goes up and batch k - 1 comes down while batch
//     k is in the kernels;
//   * a batch is a run of units of at most MSCOMP_AMD_HOST_BATCH_MB (default 512) MiB of input; runs of units that lie back to back in
//     the caller's memory travel as ONE copy each way (a file cut into 64 KiB units is one upload; outputs laid out capacity after capacity
//     come back as one download of the range that holds streams), other units one copy each.
#include "../../include/mscomp_amd.h"
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Worker {                                           // everything one device range needs, kept between calls
	int device = -1;
	mscomp_amd_ctx* ctx = nullptr;
	hipStream_t up = nullptr, ex = nullptr, dn = nullptr;
	hipEvent_t ev_up[2] = { nullptr, nullptr }, ev_ex[2] = { nullptr, nullptr }, ev_dn[2] = { nullptr, nullptr };
	void* d_in[2] = { nullptr, nullptr }; void* d_out[2] = { nullptr, nullptr }; void* d_meta[2] = { nullptr, nullptr };
	size_t in_cap[2] = { 0, 0 }, out_cap[2] = { 0, 0 }, meta_cap[2] = { 0, 0 };
	uint64_t* h_meta[2] = { nullptr, nullptr }; size_t h_cap[2] = { 0, 0 };      // pinned: out_len (u64) x n | status (i32) x n
	bool init(int dev)
	{
		device = dev;
		if (hipSetDevice(dev) != hipSuccess) { return false; }
		if (hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&ex, hipStreamNonBlocking) != hipSuccess ||
		    hipStreamCreateWithFlags(&dn, hipStreamNonBlocking) != hipSuccess) { return false; }
		for (int i = 0; i < 2; ++i) {
			if (hipEventCreateWithFlags(&ev_up[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_ex[i], hipEventDisableTiming) != hipSuccess ||
			    hipEventCreateWithFlags(&ev_dn[i], hipEventDisableTiming) != hipSuccess) { return false; }
		}
		return mscomp_amd_ctx_create(dev, ex, &ctx) == MSCOMP_OK;
	}
	static bool grow(void** p, size_t* cap, size_t need)
	{
		if (need <= *cap) { return true; }
		if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
		const size_t want = need + need / 8 + 256;
		if (hipMalloc(p, want) != hipSuccess) { *p = nullptr; (void)hipGetLastError(); return false; }
		*cap = want; return true;
	}
	bool reserve(int slot, size_t in_b, size_t out_b, size_t n)
	{
		if (!grow(&d_in[slot], &in_cap[slot], in_b + 64) || !grow(&d_out[slot], &out_cap[slot], out_b + 64) || !grow(&d_meta[slot], &meta_cap[slot], n * 16 + 64)) { return false; }
		if (h_cap[slot] < n) {
			if (h_meta[slot]) { (void)hipHostFree(h_meta[slot]); h_meta[slot] = nullptr; h_cap[slot] = 0; }
			void* q = nullptr;
			if (hipHostMalloc(&q, (n + n / 4 + 64) * 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
			h_meta[slot] = static_cast<uint64_t*>(q); h_cap[slot] = n + n / 4 + 64;
		}
		return true;
	}
	void destroy()
	{
		if (device >= 0) { (void)hipSetDevice(device); }
		if (ctx) { mscomp_amd_ctx_destroy(ctx); ctx = nullptr; }
		for (int i = 0; i < 2; ++i) {
			if (d_in[i]) { (void)hipFree(d_in[i]); } if (d_out[i]) { (void)hipFree(d_out[i]); } if (d_meta[i]) { (void)hipFree(d_meta[i]); }
			if (h_meta[i]) { (void)hipHostFree(h_meta[i]); }
			if (ev_up[i]) { (void)hipEventDestroy(ev_up[i]); } if (ev_ex[i]) { (void)hipEventDestroy(ev_ex[i]); } if (ev_dn[i]) { (void)hipEventDestroy(ev_dn[i]); }
		}
		if (up) { (void)hipStreamDestroy(up); } if (ex) { (void)hipStreamDestroy(ex); } if (dn) { (void)hipStreamDestroy(dn); }
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

size_t batch_bytes()
{
	static const size_t v = [] { const char* e = getenv("MSCOMP_AMD_HOST_BATCH_MB"); const long x = e ? atol(e) : 512; return (size_t)(x >= 1 && x <= 65536 ? x : 512) << 20; }();
	return v;
}

// units [u0, u1) on worker w: MSCOMP_OK, or the first error that stopped the range (HIP failure / out of memory / bad argument)
MSCompStatus run_range(Worker* w, const Job& j, size_t u0, size_t u1)
{
	if (hipSetDevice(w->device) != hipSuccess) { return MSCOMP_ERRNO; }
	const size_t limit = batch_bytes();
	std::vector<Batch> batches;
	for (size_t i = u0; i < u1;) {                               // batches: runs of units up to `limit` input bytes (a larger unit is a batch of its own)
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
			if (!j.decompress) { const uint64_t most = (uint64_t)ms_max_compressed_size(j.format, j.in_lens[k]) + 2u; if (most < c) { c = most; b.out_mirror = false; } }
			dcap[k - b.b0] = c;
		}
		for (size_t k = b.b0; k < b.b1; ++k) { const uint64_t c = dcap[k - b.b0]; b.out_off.push_back(out_pos); b.out_cap.push_back(c); out_pos += b.out_mirror ? c : ((c + 15u) & ~(uint64_t)15u); }
		b.out_total = out_pos;
		batches.push_back(std::move(b));
	}
	MSCompStatus rs = MSCOMP_OK;
	const size_t nb = batches.size();
	auto finish = [&](size_t k) -> MSCompStatus {                // batch k is through the kernels: results to the caller, outputs on their way down
		Batch& b = batches[k];
		const int slot = (int)(k & 1u);
		if (hipEventSynchronize(w->ev_ex[slot]) != hipSuccess) { return MSCOMP_ERRNO; }
		const size_t n = b.b1 - b.b0;
		const uint64_t* h_len = w->h_meta[slot]; const int32_t* h_st = reinterpret_cast<const int32_t*>(h_len + n);
		const uint8_t* d_out = static_cast<const uint8_t*>(w->d_out[slot]);
		MSCompStatus r = MSCOMP_OK;
		size_t last_ok = n;                                        // mirror layout: one copy up to the end of the last stream
		for (size_t i = 0; i < n; ++i) {
			const size_t u = b.b0 + i;
			j.statuses[u] = (MSCompStatus)h_st[i];
			j.out_lens[u] = h_st[i] == MSCOMP_OK ? (size_t)h_len[i] : 0;
			if (h_st[i] == MSCOMP_OK) { last_ok = i; }
		}
		auto span = [&](size_t i) -> size_t {                      // bytes of unit i to bring home: the stream, and for LZNT1 the two uncounted 00 00 behind it
			size_t len = (size_t)h_len[i];
			if (!j.decompress && j.format == MSCOMP_LZNT1 && j.out_caps[b.b0 + i] - len >= 2) { len += 2; }
			return len;
		};
		if (b.out_mirror && last_ok != n) {
			const size_t bytes = (size_t)b.out_off[last_ok] + span(last_ok);
			if (bytes && hipMemcpyAsync(j.out_ptrs[b.b0], d_out, bytes, hipMemcpyDeviceToHost, w->dn) != hipSuccess) { r = MSCOMP_ERRNO; }
		} else if (!b.out_mirror) {
			for (size_t i = 0; i < n && r == MSCOMP_OK; ++i) {
				if (h_st[i] != MSCOMP_OK) { continue; }
				const size_t bytes = span(i);
				if (bytes && hipMemcpyAsync(j.out_ptrs[b.b0 + i], d_out + b.out_off[i], bytes, hipMemcpyDeviceToHost, w->dn) != hipSuccess) { r = MSCOMP_ERRNO; }
			}
		}
		if (hipEventRecord(w->ev_dn[slot], w->dn) != hipSuccess) { r = MSCOMP_ERRNO; }
		mscomp_amd_plan_destroy(b.plan); b.plan = nullptr;
		return r;
	};
	bool used[2] = { false, false };
	for (size_t k = 0; k < nb && rs == MSCOMP_OK; ++k) {
		Batch& b = batches[k];
		const int slot = (int)(k & 1u);
		const size_t n = b.b1 - b.b0;
		if (used[slot] && hipEventSynchronize(w->ev_dn[slot]) != hipSuccess) { rs = MSCOMP_ERRNO; break; }     // the slot's buffers are free again
		if (!w->reserve(slot, (size_t)b.in_total, (size_t)b.out_total, n)) { rs = MSCOMP_MEM_ERROR; break; }
		used[slot] = true;
		uint8_t* d_in = static_cast<uint8_t*>(w->d_in[slot]);
		for (size_t i = 0; i < n && rs == MSCOMP_OK;) {             // uploads: units that lie back to back in the caller's memory AND on the device go as one copy
			size_t e = i + 1;
			uint64_t bytes = b.in_len[i];
			while (e < n && (b.in_len[e - 1] & 15u) == 0 && j.in_ptrs[b.b0 + e] == j.in_ptrs[b.b0 + e - 1] + b.in_len[e - 1]) { bytes += b.in_len[e]; ++e; }
			if (bytes && hipMemcpyAsync(d_in + b.in_off[i], j.in_ptrs[b.b0 + i], (size_t)bytes, hipMemcpyHostToDevice, w->up) != hipSuccess) { rs = MSCOMP_ERRNO; }
			i = e;
		}
		if (rs == MSCOMP_OK && hipEventRecord(w->ev_up[slot], w->up) != hipSuccess) { rs = MSCOMP_ERRNO; }
		if (rs != MSCOMP_OK) { break; }
		// (the plan's tables go up on the kernel stream and wait for it: batch k - 1 is in the kernels meanwhile, batch k on its way up)
		rs = j.decompress ? mscomp_amd_plan_create_decompress(w->ctx, j.format, n, b.in_off.data(), b.in_len.data(), b.out_off.data(), b.out_cap.data(), &b.plan)
		                  : mscomp_amd_plan_create(w->ctx, j.format, n, b.in_off.data(), b.in_len.data(), b.out_off.data(), b.out_cap.data(), &b.plan);
		if (rs != MSCOMP_OK) { break; }
		uint64_t* d_len = static_cast<uint64_t*>(w->d_meta[slot]); int32_t* d_st = reinterpret_cast<int32_t*>(d_len + n);
		if (hipStreamWaitEvent(w->ex, w->ev_up[slot], 0) != hipSuccess) { rs = MSCOMP_ERRNO; break; }
		rs = mscomp_amd_plan_execute(b.plan, d_in, static_cast<uint8_t*>(w->d_out[slot]), d_len, d_st);
		if (rs != MSCOMP_OK) { break; }
		if (hipMemcpyAsync(w->h_meta[slot], d_len, n * 12, hipMemcpyDeviceToHost, w->ex) != hipSuccess || hipEventRecord(w->ev_ex[slot], w->ex) != hipSuccess) { rs = MSCOMP_ERRNO; break; }
		// (no stream wait of `dn` on ev_ex here: finish(k) waits for that event on the HOST before it queues batch k's copies, and a wait queued now
		//  would put batch k - 1's downloads behind batch k's kernels)
		if (k >= 1) { rs = finish(k - 1); }
	}
	if (rs == MSCOMP_OK && nb) { rs = finish(nb - 1); }
	(void)hipStreamSynchronize(w->up); (void)hipStreamSynchronize(w->ex);
	if (hipStreamSynchronize(w->dn) != hipSuccess && rs == MSCOMP_OK) { rs = MSCOMP_ERRNO; }
	for (auto& b : batches) { if (b.plan) { mscomp_amd_plan_destroy(b.plan); b.plan = nullptr; } }
	(void)hipGetLastError();
	return rs;
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
	auto body = [&](int r) {
		if (cuts[r] == cuts[r + 1]) { return; }
		Worker* w = take_worker(dev[r]);
		if (!w) { res[r] = MSCOMP_ERRNO; return; }
		res[r] = run_range(w, job, cuts[r], cuts[r + 1]);
		give_worker(w);
	};
	for (int r = 1; r < n_dev; ++r) { threads.emplace_back(body, r); }
	body(0);                                                         // range 0 on the calling thread
	for (auto& t : threads) { t.join(); }
	(void)hipSetDevice(prev);
	for (int r = 0; r < n_dev; ++r) { if (res[r] != MSCOMP_OK) { return res[r]; } }
	return MSCOMP_OK;
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

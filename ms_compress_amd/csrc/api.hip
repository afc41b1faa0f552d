// api.hip -- C-ABI of libmscomp_amd.so (include/mscomp_amd.h): context, plan, batch execution and the
// drop-in one-shot entry points. Host orchestration only; every output byte comes from the HIP kernels.
//
// Reference boundary mirrored here: ms_compress / ms_max_compressed_size (/root/reference/src/mscomp.cpp:96-117),
// lznt1_compress (/root/reference/src/lznt1_compress.cpp:233), xpress_compress (/root/reference/src/xpress_compress.cpp:240),
// xpress_huff_compress (/root/reference/src/xpress_huff_compress.cpp:247).
#include "../../include/mscomp_amd.h"
#include "kernels.h"
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <condition_variable>
#include <hip/hip_runtime.h>
#include <new>
#include <string>
#include <vector>
#include <cstring>

using namespace msc;

namespace {

// A captured hipGraph of a plan holds raw scratch pointers and a kernel choice. Two counters say when it is stale:
// the context's own epoch (bumped whenever one of ITS device buffers moves; a context is used by one host thread at a
// time, so a plain counter) and a process-wide one for the test hooks that switch kernels (atomic: any thread may call them
// while other threads execute plans).
static std::atomic<uint64_t> g_mode_epoch{1};
static std::atomic<int> g_finder_mode{1};              // 1 = default (Xpress units up to 64 KiB: the lazy finder, xpress_lazy.hip), 2 = Find for every position everywhere
static int lznt1_sa_env() { const char* e = getenv("MSCOMP_AMD_LZNT1_SA_DICT"); return (e && *e && *e != '0') ? 1 : 0; }
static std::atomic<int> g_lznt1_sa{lznt1_sa_env()};    // 1 = LZNT1 compresses with the suffix-array dictionary flavour (lznt1_sa.hip; the reference's MSCOMP_WITH_LZNT1_SA_DICT build)
static std::atomic<int> g_one_zero_copy{[] { const char* e = getenv("MSCOMP_AMD_ONE_ZEROCOPY"); return (e && *e == '0') ? 0 : 1; }()};   // large host-pointer LZNT1 calls: 1 = caller buffers mapped, one launch (default), 0 = slices on three streams
static std::atomic<int> g_xpd_mode{0};                 // Xpress decompression: 0 = tokens a flag word at a time + copy kernels (default; large streams by segments), 1 = xpd_kernel (a token at a time, bytes in the same wave), 2 = as 0 without the segments
struct DevBuf {
	void* p = nullptr; size_t cap = 0;
	uint64_t* epoch = nullptr;                         // the owning context's epoch (null for plan-owned tables)
	bool reserve(size_t n)
	{
		if (n <= cap) { return true; }
		if (epoch) { ++*epoch; }
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		const size_t want = n + n / 8 + 256;
		if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return false; }
		cap = want; return true;
	}
	void release() { if (p) { (void)hipFree(p); if (epoch) { ++*epoch; } } p = nullptr; cap = 0; }
};

struct ProfRec { const char* name; hipEvent_t a, b; };
bool lznt1_sa_for(const mscomp_amd_ctx* c);              // the flavour a plan created in `c` right now would get
const uint32_t XH_FB_BLOCKS = 256;                     // persistent blocks of the fallback kernel (one per CU: its package pool fills the LDS)

} // namespace

struct mscomp_amd_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	DevBuf slots, slot_size, prefix, tile_sums;        // chunk scratch (grow-only, shared by all plans of the ctx)
	DevBuf links, lasthead, mlen3, moff;               // Xpress-family match finder scratch (per 64 KiB link chunk)
	DevBuf wtok, wmat, wfar;                           // Xpress parse records per 64-position window (token mask, match mask, far length)
	DevBuf wrec, sbrec;                                // ... state / counts / prefixes per window (6 x u32), per super-block (tot 4 x u32, pre 3 x u64, seams)
	DevBuf tokbits, counts, extra, lens, codes, fb_list, fbflag;   // Xpress+Huffman per-chunk scratch
	DevBuf dz_cin, dz_csize, dz_unit;                  // LZNT1 decompression: header offset / decoded size per chunk slot, per-unit records
	DevBuf dz_scr;                                     // Xpress+Huffman decompression: token scratch of the candidates of multi-chunk buffers
	DevBuf dz_tok, dz_ntok, dz_xhc;                    // Xpress+Huffman decompression: 32-bit tokens of every unit, token counts, candidate chunk records
	DevBuf lzg_bsum, lzg_dir, lzg_words;               // tokens -> bytes of large units by all CUs (lzglobal.hip): token block sums, tile directory, a word per output byte + pass counters
	DevBuf xps_buf;                                    // large Xpress streams by segments: segment records | mode per stream | done per unit
	DevBuf cp_tab;                                     // compaction: out_off (u64) | tile_prefix (u32) of the batch being packed
	DevBuf one_in, one_out, one_meta;                  // staging of the host-pointer one-shot path
	void* h_tab = nullptr; size_t h_tab_cap = 0;       // pinned staging of a plan's tables: they go up stream-ordered, plan_create does not wait for the stream
	hipEvent_t h_tab_ev = nullptr; bool h_tab_busy = false;   // (a stream that shares a hardware queue with a busy one would make that wait as long as the other's kernels)
	std::vector<DevBuf> table_pool;                    // table buffers of destroyed plans, reused by the next plan (hipFree waits for the whole device: it would stall pipelines that create a plan per batch)
	uint64_t epoch = 1;                                // bumped when one of the buffers above moves (captured graphs are stale then)
	int lznt1_sa = -1;                                 // LZNT1 dictionary flavour of the plans this context creates: -1 = the process default at plan creation, 0 / 1 = set for this context
	bool profiling = false;
	std::vector<ProfRec> recs;
	std::vector<hipEvent_t> free_events;
	std::vector<DevBuf*> bufs()
	{
		return { &slots, &slot_size, &prefix, &tile_sums, &links, &lasthead, &mlen3, &moff, &wtok, &wmat, &wfar, &wrec, &sbrec,
		         &tokbits, &counts, &extra, &lens, &codes, &fb_list, &fbflag, &dz_cin, &dz_csize, &dz_unit, &dz_tok, &dz_ntok, &dz_xhc, &dz_scr,
		         &lzg_bsum, &lzg_dir, &lzg_words, &xps_buf, &cp_tab, &one_in, &one_out, &one_meta };
	}
	mscomp_amd_ctx() { for (DevBuf* b : bufs()) { b->epoch = &epoch; } }
};

struct mscomp_amd_plan {
	mscomp_amd_ctx* ctx = nullptr;
	MSCompFormat format = MSCOMP_NONE;
	bool decompress = false;
	uint32_t n_units = 0, n_chunks = 0;
	uint64_t total_in = 0, max_unit = 0;
	bool lznt1_sa = false;                             // LZNT1: the suffix-array dictionary flavour -- fixed when the plan is created: a plan never changes its bytes under a running caller
	bool matches_ready = false;                        // the caller has run the links + find kernels of this execution itself, range by range (the pipelined one-shot call): plan_launch skips them once
	bool no_graph = false;                             // one-shot plans run with changing buffer addresses: a captured graph would be re-captured every time
	DevBuf tables;                                     // in_off | out_off | chunk_prefix
	DevBuf tokpre;                                     // decompression by tokens: first token slot | first candidate slot of every unit (2 x (n_units + 1) u64)
	uint32_t xhc_slots = 0;                            // candidate chunk slots of the batch
	uint64_t xhc_scr = 0;                              // token-scratch slots (candidates of multi-chunk buffers), 0 = none
	DevBuf lzg_tab;                                    // lzglobal.hip: unit (u32 x n_big, padded) | tb_prefix | tile_prefix | word_prefix (u64 x (n_big + 1) each)
	uint32_t lzg_big = 0, lzg_tb = 0, lzg_tiles = 0;   // units taken by that path (0: not used), their token blocks and tiles
	uint64_t lzg_words = 0;
	DevBuf xps_tab;                                    // xps_*: unit (u32 x n_big, padded) | seg_prefix (u64 x (n_big + 1))
	uint32_t xps_big = 0, xps_seg = 0, xps_seg_bytes = 0, xps_warm_bytes = 0;
	BatchTables bt{};
	// the launch sequence of plan_execute as a hipGraph: captured on the plan's second execution, replayed while the
	// arguments and the scratch buffers stay where they were
	hipGraphExec_t gexec = nullptr;
	const void* g_args[4] = { nullptr, nullptr, nullptr, nullptr };
	uint64_t g_epoch = 0, g_mode = 0;
	uint32_t executions = 0;
};

namespace {

// The mscomp_amd_debug_set_* hooks pick between bit-identical kernels for the tests. They are process-wide, so the product ignores them unless the
// process asked for them BEFORE the library was loaded (MSCOMP_AMD_TEST_HOOKS=1: tests/conftest.py and the tools do): a deployment cannot have its
// kernels switched under it by a stray call.
bool test_hooks_on() { static const bool on = [] { const char* e = getenv("MSCOMP_AMD_TEST_HOOKS"); return e && *e == '1'; }(); return on; }
bool lznt1_sa_for(const mscomp_amd_ctx* c) { return (c->lznt1_sa >= 0 ? c->lznt1_sa : g_lznt1_sa.load(std::memory_order_relaxed)) != 0; }

struct DeviceGuard {
	int prev = -1; bool ok = true;
	explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; } if (prev != dev) { ok = hipSetDevice(dev) == hipSuccess; } }
	~DeviceGuard() { int cur = -1; if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) { (void)hipSetDevice(prev); } }
};

hipEvent_t get_event(mscomp_amd_ctx* c)
{
	if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
	hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
struct KernelTimer {
	mscomp_amd_ctx* c; ProfRec r;
	KernelTimer(mscomp_amd_ctx* ctx, const char* name) : c(ctx), r{name, nullptr, nullptr}
	{ if (c->profiling) { r.a = get_event(c); r.b = get_event(c); (void)hipEventRecord(r.a, c->stream); } }
	~KernelTimer() { if (c->profiling) { (void)hipEventRecord(r.b, c->stream); c->recs.push_back(r); } }
};

// Bytes an OPTIONAL fast-path scratch buffer may take: the environment's cap, and never more than half of what the device has free
// right now (what the buffer already holds counts as free: reserve() reuses it). The fast paths have fallbacks that need no scratch,
// so a plan never fails for them (the constants above assume a 288 GB device that is otherwise empty; a caller may hold most of it).
uint64_t optional_scratch_budget(uint64_t env_cap, size_t already_held)
{
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return env_cap; }
	const uint64_t half = ((uint64_t)free_b + already_held) / 2;
	return env_cap < half ? env_cap : half;
}

uint32_t chunks_of(MSCompFormat f, bool decompress, uint64_t n)
{
	if (decompress) {                                  // LZNT1: segments of the header walk; Xpress+Huffman: tiles of the candidate search
		if (f == MSCOMP_LZNT1) { return (uint32_t)(n ? (n + LZD_SEG - 1u) / LZD_SEG : 1u); }
		if (f == MSCOMP_XPRESS_HUFF) { return (uint32_t)(n ? (n + XHC_TILE_BYTES - 1u) / XHC_TILE_BYTES : 1u); }
		return 1u;
	}
	switch (f) {
	case MSCOMP_LZNT1:       return (uint32_t)((n + 4095u) / 4096u);
	case MSCOMP_XPRESS_HUFF: return (uint32_t)((n + 65535u) / 65536u);   // n==0 -> 0 chunks, 0 bytes of output
	default:                 return (uint32_t)((n + 65535u) / 65536u);   // Xpress: 64 KiB link chunks (one stream per unit)
	}
}

} // namespace

extern "C" {

#ifdef MSCOMP_AMD_DEV
#define MSCOMP_AMD_FLAVOUR "; DEVELOPMENT flavour: + the measurement-mode Xpress+Huffman finders"
#else
#define MSCOMP_AMD_FLAVOUR ""
#endif
const char* mscomp_amd_version(void) { return "mscomp_amd 0.6 (gfx950, HIP; LZNT1 / Xpress / Xpress+Huffman compressors and decompressors, LZNT1 streaming, host batches over several GPUs" MSCOMP_AMD_FLAVOUR ")"; }

size_t lznt1_max_compressed_size(size_t n)       { return n + 3 + 2 * ((n + 4095) / 4096); }
size_t xpress_max_compressed_size(size_t n)      { return n + 4 + 4 * (n / 32); }
size_t xpress_huff_max_compressed_size(size_t n) { return n + 34 + 258 + 258 * (n / 65536); }
#ifndef MSCOMP_AMD_NO_FACADE   /* define when src/mscomp.cpp of the reference is linked too */
size_t ms_max_compressed_size(MSCompFormat f, size_t n)
{
	switch ((int)f) {
	case MSCOMP_NONE:        return n;
	case MSCOMP_LZNT1:       return lznt1_max_compressed_size(n);
	case MSCOMP_XPRESS:      return xpress_max_compressed_size(n);
	case MSCOMP_XPRESS_HUFF: return xpress_huff_max_compressed_size(n);
	default:                 return (size_t)-1;                          // mscomp.cpp:98
	}
}
#endif

// The LZNT1 bucket sort and the Xpress chain links are only bit-exact when the returning same-address LDS atomics of one
// wave instruction are served in lane order (util.hip: lds_lane_order_kernel). That is how gfx950 behaves, but it is not an
// architectural promise, so the library checks the device it is about to use, once per device and process. A device that serves them in
// another order gets the order-independent form of the two kernels (one lane at a time: kernels.h set_serial_atomics -- the same bytes,
// a slower sort); a device on which the check cannot RUN is refused (MSCOMP_ERRNO): wrong bytes must never leave silently.
static int lane_order_verdict(int device, hipStream_t st)
{
	static std::mutex mu;
	static int verdict[64];                            // 0 = unknown, 1 = in order, -1 = out of order / check failed
	if (device >= 64) { device = 63; }
	std::lock_guard<std::mutex> lk(mu);
	if (verdict[device] == 0) {
		uint32_t* d_bad = nullptr;
		uint32_t bad = 0xFFFFFFFFu;
		if (hipMalloc(reinterpret_cast<void**>(&d_bad), 64) == hipSuccess) {
			bad = run_lds_lane_order_check(st, 0x5EEDu + (uint32_t)device, 64, 64, 97, d_bad);     // 64 blocks x 64 rounds x 64 lanes x {add, exchange}
			if (bad == 0) { bad = run_lds_lane_order_check(st, 0xC0FFEEu, 64, 64, 2048, d_bad); }
			(void)hipFree(d_bad);
		}
		verdict[device] = bad == 0 ? 1 : (bad == 0xFFFFFFFFu ? -1 : 2);             // 2 = runs, out of order: the serial form
		if (verdict[device] == 2) { set_serial_atomics(device, 1); }
	}
	return verdict[device];
}

MSCompStatus mscomp_amd_ctx_create(int device, void* hip_stream, mscomp_amd_ctx** out)
{
	if (!out) { return MSCOMP_ARG_ERROR; }
	*out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { return MSCOMP_ERRNO; }
	DeviceGuard g(device);
	if (!g.ok) { return MSCOMP_ERRNO; }
	if (lane_order_verdict(device, (hipStream_t)hip_stream) < 1) { return MSCOMP_ERRNO; }
	mscomp_amd_ctx* c = new (std::nothrow) mscomp_amd_ctx();
	if (!c) { return MSCOMP_MEM_ERROR; }
	c->device = device; c->stream = (hipStream_t)hip_stream;
	*out = c;
	return MSCOMP_OK;
}

void mscomp_amd_ctx_destroy(mscomp_amd_ctx* c)
{
	if (!c) { return; }
	DeviceGuard g(c->device);
	(void)hipStreamSynchronize(c->stream);
	for (auto& r : c->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
	for (auto e : c->free_events) { (void)hipEventDestroy(e); }
	for (DevBuf* b : c->bufs()) { b->release(); }
	for (DevBuf& b : c->table_pool) { b.release(); }
	if (c->h_tab) { (void)hipHostFree(c->h_tab); }
	if (c->h_tab_ev) { (void)hipEventDestroy(c->h_tab_ev); }
	delete c;
}

void mscomp_amd_profile_enable(mscomp_amd_ctx* c, int on) { if (c) { c->profiling = on != 0; } }

int mscomp_amd_profile_read(mscomp_amd_ctx* c, const char** names, double* ms, uint64_t* launches, int cap)
{
	if (!c) { return 0; }
	DeviceGuard g(c->device);
	(void)hipStreamSynchronize(c->stream);
	int n = 0;
	for (auto& r : c->recs) {
		float t = 0.f; (void)hipEventElapsedTime(&t, r.a, r.b);
		int k = 0;
		for (; k < n; ++k) { if (names[k] == r.name) { break; } }
		if (k == n) { if (n >= cap) { continue; } names[n] = r.name; ms[n] = 0; launches[n] = 0; ++n; }
		ms[k] += t; launches[k] += 1;
		c->free_events.push_back(r.a); c->free_events.push_back(r.b);
	}
	c->recs.clear();
	return n;
}

static MSCompStatus plan_create_impl(mscomp_amd_ctx* c, MSCompFormat format, bool decompress, size_t n_units,
                                     const uint64_t* in_off, const uint64_t* in_len,
                                     const uint64_t* out_off, const uint64_t* out_cap, mscomp_amd_plan** out)
{
	if (!out) { return MSCOMP_ARG_ERROR; }
	*out = nullptr;
	if (!c || (n_units && (!in_off || !in_len || !out_off || !out_cap)) || n_units > 0x7FFFFFF0u) { return MSCOMP_ARG_ERROR; }
	if (format != MSCOMP_LZNT1 && format != MSCOMP_XPRESS && format != MSCOMP_XPRESS_HUFF) { return MSCOMP_ARG_ERROR; }
	DeviceGuard g(c->device);
	if (!g.ok) { return MSCOMP_ERRNO; }
	mscomp_amd_plan* p = new (std::nothrow) mscomp_amd_plan();
	if (!p) { return MSCOMP_MEM_ERROR; }
	p->ctx = c; p->format = format; p->decompress = decompress; p->n_units = (uint32_t)n_units;
	p->lznt1_sa = !decompress && format == MSCOMP_LZNT1 && lznt1_sa_for(c);

	const size_t host_words = n_units * 4 + (n_units + 2) / 2 + 1;
	std::vector<uint64_t> host;
	uint64_t* h = nullptr;
	if (!decompress) {                                   // compress plans: pinned staging, no stream wait (the decompress plans below upload more tables and wait as before)
		if (c->h_tab_busy) { (void)hipEventSynchronize(c->h_tab_ev); c->h_tab_busy = false; }   // (the previous plan's copy: long done)
		if (c->h_tab_cap < host_words * 8) {
			if (c->h_tab) { (void)hipHostFree(c->h_tab); c->h_tab = nullptr; c->h_tab_cap = 0; }
			const size_t want = host_words * 8 + host_words + 4096;
			if (hipHostMalloc(&c->h_tab, want, hipHostMallocDefault) == hipSuccess) { c->h_tab_cap = want; } else { c->h_tab = nullptr; (void)hipGetLastError(); }
		}
		if (c->h_tab && !c->h_tab_ev && hipEventCreateWithFlags(&c->h_tab_ev, hipEventDisableTiming) != hipSuccess) { c->h_tab_ev = nullptr; (void)hipGetLastError(); }
		if (c->h_tab && c->h_tab_ev) { h = static_cast<uint64_t*>(c->h_tab); }
	}
	const bool pinned = h != nullptr;
	if (!pinned) { host.resize(host_words); h = host.data(); }
	uint32_t* h_cp = reinterpret_cast<uint32_t*>(h + 4 * n_units);
	uint64_t chunks = 0, total = 0;
	for (size_t i = 0; i < n_units; ++i) {
		h[i] = in_off[i]; h[n_units + i] = in_len[i]; h[2 * n_units + i] = out_off[i]; h[3 * n_units + i] = out_cap[i];
		h_cp[i] = (uint32_t)chunks;
		if (decompress && in_len[i] > 0xFFFFF000u) { delete p; return MSCOMP_ARG_ERROR; }   // offsets inside a unit are 32-bit (4 GiB - 4096 at most)
		chunks += chunks_of(format, decompress, in_len[i]);
		total += in_len[i];
		if (in_len[i] > p->max_unit) { p->max_unit = in_len[i]; }
		if (chunks > 0x7FFFFFF0u) { delete p; return MSCOMP_ARG_ERROR; }
	}
	h_cp[n_units] = (uint32_t)chunks;
	p->n_chunks = (uint32_t)chunks;
	p->total_in = total;
	const size_t bytes = host_words * sizeof(uint64_t);
	{	// the smallest pooled buffer that is large enough, else a new one
		size_t best = c->table_pool.size();
		for (size_t i = 0; i < c->table_pool.size(); ++i) { if (c->table_pool[i].cap >= bytes && (best == c->table_pool.size() || c->table_pool[i].cap < c->table_pool[best].cap)) { best = i; } }
		if (best < c->table_pool.size()) { p->tables = c->table_pool[best]; c->table_pool.erase(c->table_pool.begin() + (long)best); }
	}
	if (!p->tables.reserve(bytes)) { delete p; return MSCOMP_MEM_ERROR; }
	if (hipMemcpyAsync(p->tables.p, h, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) { p->tables.release(); delete p; return MSCOMP_ERRNO; }
	if (pinned) { if (hipEventRecord(c->h_tab_ev, c->stream) != hipSuccess) { p->tables.release(); delete p; return MSCOMP_ERRNO; } c->h_tab_busy = true; }
	else if (hipStreamSynchronize(c->stream) != hipSuccess) { p->tables.release(); delete p; return MSCOMP_ERRNO; }
	uint64_t* d = static_cast<uint64_t*>(p->tables.p);
	p->bt.in_off = d; p->bt.in_len = d + n_units; p->bt.out_off = d + 2 * n_units; p->bt.out_cap = d + 3 * n_units;
	p->bt.chunk_prefix = reinterpret_cast<const uint32_t*>(d + 4 * n_units);
	p->bt.n_units = p->n_units; p->bt.n_chunks = p->n_chunks;

	// size the ctx scratch now so that execute() never allocates
	if (decompress) {
		bool okd = c->prefix.reserve(((size_t)p->n_chunks + 2) * sizeof(uint64_t)) && c->tile_sums.reserve(((size_t)p->n_chunks / 1024 + 4) * sizeof(uint64_t));
		if (okd && format == MSCOMP_LZNT1) {
			okd = c->dz_cin.reserve((size_t)p->n_chunks * LZD_SLOTS * 4 + 64) && c->dz_csize.reserve((size_t)p->n_chunks * LZD_SLOTS * 2 + 64) &&
			      c->dz_unit.reserve(((size_t)p->n_chunks * (5 * LZD_K + 2) + (size_t)n_units * 2 + 8) * 4);
		}
		if (okd && format == MSCOMP_XPRESS_HUFF) {
			// a symbol gives one token, a match one more per 32766 bytes; no token without an output byte
			std::vector<uint64_t> tp(3 * (n_units + 1));                  // first token slot | first candidate slot | first token-scratch slot of every unit
			uint64_t slots = 0, cands = 0, scr = 0;
			for (size_t i = 0; i < n_units; ++i) {
				tp[i] = slots; tp[n_units + 1 + i] = cands; tp[2 * (n_units + 1) + i] = scr;
				const uint64_t by_in = 8 * in_len[i] + out_cap[i] / 32766u + 1, cnt = out_cap[i] < by_in ? out_cap[i] : by_in;
				slots += cnt + 64;
				// candidate chunk starts: a chunk gives 65536 bytes and takes at least 260; a quarter more for windows that only look like a table
				const uint64_t by_out = out_cap[i] / 65536u + 2, by_len = in_len[i] / 260u + 1, most = by_out < by_len ? by_out : by_len;
				cands += most + most / 4 + 2;
				if (out_cap[i] > 65536u) { scr += most + most / 4 + 2; }   // a buffer of several chunks: its candidates keep their tokens (no second walk)
			}
			tp[n_units] = slots; tp[2 * n_units + 1] = cands; tp[3 * n_units + 2] = scr;
			if (cands > 0x7FFFFFF0u) { p->tables.release(); delete p; return MSCOMP_ARG_ERROR; }
			p->xhc_slots = (uint32_t)cands;
			okd = p->tokpre.reserve(tp.size() * 8) && c->dz_tok.reserve(slots * 4 + 256) && c->dz_ntok.reserve((n_units + 1) * 8) &&
			      c->dz_xhc.reserve(((size_t)n_units + 1) * 8 + (size_t)cands * (4 * 4 + 3 * 8) + 64);
			{	// ... if that scratch is affordable: at most MSCOMP_AMD_XHC_SCR_MAX_MB (default 32 GiB; 0 = always walk twice) and at most half of
				// what the device has free right now; a reservation that fails anyway switches the path off (the second walk needs no scratch)
				static const uint64_t scr_env = [] { const char* e = getenv("MSCOMP_AMD_XHC_SCR_MAX_MB"); const long long v = e ? atoll(e) : 32768; return (uint64_t)(v < 0 ? 0 : v) << 20; }();
				const uint64_t need = scr * XHC_SCR * 4 + 64, scr_budget = optional_scratch_budget(scr_env, c->dz_scr.cap);
				if (scr && (need > scr_budget || !okd || !c->dz_scr.reserve(need))) { for (size_t i = 0; i <= n_units; ++i) { tp[2 * (n_units + 1) + i] = 0; } scr = 0; }
				p->xhc_scr = scr;
			}
			if (okd && (hipMemcpyAsync(p->tokpre.p, tp.data(), tp.size() * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
			            hipStreamSynchronize(c->stream) != hipSuccess)) { p->tokpre.release(); p->tables.release(); delete p; return MSCOMP_ERRNO; }
		}
		if (okd && format == MSCOMP_XPRESS) {
			// a token takes at least one input byte and gives at least one output byte; a match one more token per 32766 bytes
			std::vector<uint64_t> tp(n_units + 1);
			uint64_t slots = 0;
			for (size_t i = 0; i < n_units; ++i) {
				tp[i] = slots;
				const uint64_t by_in = in_len[i] + out_cap[i] / 32766u + 1, cnt = out_cap[i] < by_in ? out_cap[i] : by_in;
				slots += cnt + 64;
			}
			tp[n_units] = slots;
			okd = p->tokpre.reserve(tp.size() * 8) && c->dz_tok.reserve(slots * 4 + 256) && c->dz_ntok.reserve((n_units + 1) * 8);
			if (okd && (hipMemcpyAsync(p->tokpre.p, tp.data(), tp.size() * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
			            hipStreamSynchronize(c->stream) != hipSuccess)) { p->tokpre.release(); p->tables.release(); delete p; return MSCOMP_ERRNO; }
		}
		if (okd && (format == MSCOMP_XPRESS || format == MSCOMP_XPRESS_HUFF)) {
			// units with room for LZG_MIN_CAP bytes or more get their bytes from all CUs (lzglobal.hip): 4 bytes of scratch per byte of capacity;
			// when that is more than the budget (MSCOMP_AMD_LZG_MAX_MB, default 64 GiB of the 288; 0 switches the path off) the block-per-unit kernel takes them
			static const uint64_t budget_env = [] { const char* e = getenv("MSCOMP_AMD_LZG_MAX_MB"); const long long v = e ? atoll(e) : 65536; return (uint64_t)(v < 0 ? 0 : v) << 20; }();
			const uint64_t budget = optional_scratch_budget(budget_env, c->lzg_words.cap);   // ... and never more than half of the free HBM: the block kernel needs none of it
			std::vector<uint32_t> big;
			uint64_t words = 0;
			bool too_large = false;                                       // (32-bit word indices: a unit with room for 4 GiB keeps the whole plan on the block kernel, which every such unit then takes)
			for (size_t i = 0; i < n_units; ++i) { if (out_cap[i] >= 0xFFFFFF00ull) { too_large = true; } else if (out_cap[i] >= LZG_MIN_CAP) { big.push_back((uint32_t)i); words += out_cap[i] + 64; } }
			if (too_large) { big.clear(); }
			// ... and when it pays: the all-CU stage costs about 22 ms per GB of output whatever the units are (74 ms for 192 files, 3.39 GB), the
			// block-per-unit kernel about 1 ms per MB of the LARGEST unit as long as there are no more large units than CUs (51 ms for the same 192
			// files, whose largest is 51 MB; 60 ms for 12 of them, where the all-CU stage takes 5 ms)
			bool pays = false;
			{
				uint64_t tot = 0, mx = 0;
				for (uint32_t i : big) { tot += out_cap[i]; mx = out_cap[i] > mx ? out_cap[i] : mx; }
				const double mb = 1.0 / (1 << 20), c_all = 0.022 * (double)tot * mb, per_cu = (double)tot * mb / 256.0, c_blk = 1.0 * ((double)mx * mb > per_cu ? (double)mx * mb : per_cu);
				pays = c_all < c_blk;
			}
			if (!big.empty() && pays && words * 4 <= budget) {
				const size_t nb = big.size(), upad = (nb + 1) / 2;              // the unit list in whole u64 slots
				std::vector<uint64_t> tab(upad + 3 * (nb + 1));
				memcpy(tab.data(), big.data(), nb * 4);
				uint64_t* tbp = tab.data() + upad; uint64_t* tlp = tbp + nb + 1; uint64_t* wdp = tlp + nb + 1;
				uint64_t tb = 0, tl = 0, wd = 0;
				for (size_t k = 0; k < nb; ++k) {
					const size_t i = big[k];
					const uint64_t by_in = (format == MSCOMP_XPRESS ? 1 : 8) * in_len[i] + out_cap[i] / 32766u + 1, cnt = (out_cap[i] < by_in ? out_cap[i] : by_in) + 64;   // (the unit's token slots, as above)
					tbp[k] = tb; tlp[k] = tl; wdp[k] = wd;
					tb += (cnt + 8191) / 8192; tl += (out_cap[i] + (1u << LZG_TILE_SHIFT) - 1) >> LZG_TILE_SHIFT; wd += out_cap[i] + 64;
				}
				tbp[nb] = tb; tlp[nb] = tl; wdp[nb] = wd;
				if (tb < 0x7FFFFFF0ull && tl < 0x7FFFFFF0ull) {
					// an allocation that fails here only switches this (optional) path off: the block-per-unit kernel takes the units instead
					const bool got = p->lzg_tab.reserve(tab.size() * 8) && c->lzg_bsum.reserve(tb * 8 + 64) && c->lzg_dir.reserve(tl * 8 + 64) && c->lzg_words.reserve(wd * 4 + LZG_PASSES * 4 + tl + 64);
					if (got && (hipMemcpyAsync(p->lzg_tab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
					            hipStreamSynchronize(c->stream) != hipSuccess)) { p->lzg_tab.release(); p->tokpre.release(); p->tables.release(); delete p; return MSCOMP_ERRNO; }
					if (got) { p->lzg_big = (uint32_t)nb; p->lzg_tb = (uint32_t)tb; p->lzg_tiles = (uint32_t)tl; p->lzg_words = wd; }
					else { p->lzg_tab.release(); (void)hipGetLastError(); }
				}
			}
		}
		if (okd && format == MSCOMP_XPRESS) {
			std::vector<uint32_t> big;
			for (size_t i = 0; i < n_units; ++i) { if (in_len[i] >= XPS_MIN_IN) { big.push_back((uint32_t)i); } }
			if (!big.empty()) {
				const size_t nb = big.size(), upad = (nb + 1) / 2;
				std::vector<uint64_t> tab(upad + nb + 1);
				memcpy(tab.data(), big.data(), nb * 4);
				// segment size / warm-up (MSCOMP_AMD_XPS_SEG_KB / _WARM_KB override the defaults, for measurements)
				static const uint32_t seg_b = [] { const char* e = getenv("MSCOMP_AMD_XPS_SEG_KB"); const long v = e ? atol(e) : 0; return (uint32_t)(v >= 4 && v <= 65536 ? v << 10 : XPS_SEG); }();
				static const uint32_t warm_b = [] { const char* e = getenv("MSCOMP_AMD_XPS_WARM_KB"); const long v = e ? atol(e) : 0; const uint32_t w = (uint32_t)(v >= 1 && v <= 65536 ? v << 10 : XPS_WARM); return w < seg_b ? w : seg_b; }();
				p->xps_seg_bytes = seg_b; p->xps_warm_bytes = warm_b;
				uint64_t sg = 0;
				for (size_t k = 0; k < nb; ++k) { tab[upad + k] = sg; sg += (in_len[big[k]] + seg_b - 1) / seg_b; }
				tab[upad + nb] = sg;
				if (sg < 0x7FFFFFF0ull) {
					okd = p->xps_tab.reserve(tab.size() * 8) && c->xps_buf.reserve(sg * XPS_SEG_BYTES + nb * 4 + n_units * 4 + 64);
					if (okd && (hipMemcpyAsync(p->xps_tab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
					            hipStreamSynchronize(c->stream) != hipSuccess)) { p->xps_tab.release(); p->lzg_tab.release(); p->tokpre.release(); p->tables.release(); delete p; return MSCOMP_ERRNO; }
					if (okd) { p->xps_big = (uint32_t)nb; p->xps_seg = (uint32_t)sg; }
				}
			}
		}
		if (!okd) { p->xps_tab.release(); p->lzg_tab.release(); p->tokpre.release(); p->tables.release(); delete p; return MSCOMP_MEM_ERROR; }
		*out = p;
		return MSCOMP_OK;
	}
	bool ok = c->slot_size.reserve(((size_t)p->n_chunks + 1) * sizeof(uint32_t)) &&
	          c->prefix.reserve(((size_t)p->n_chunks + 2) * sizeof(uint64_t)) &&
	          c->tile_sums.reserve(((size_t)p->n_chunks / 1024 + 4) * sizeof(uint64_t));
	if (ok && format == MSCOMP_LZNT1) { ok = c->slots.reserve((size_t)p->n_chunks * LZNT1_SLOT + 64); }
	if (ok && format != MSCOMP_LZNT1) {
		const size_t per = (size_t)p->n_chunks * 65536u * sizeof(uint16_t) + 64;
		// (mlen3: 4 bytes per position, both halves of the match word)
#ifdef MSCOMP_AMD_DEV
		// (development flavour: Xpress+Huffman's sorted-span finder keeps its padded position array in `links` and its bucket starts, XS_STARTS_STRIDE
		// words per chunk, in `lasthead` -- reserved only when that measurement mode is switched on)
		static const bool xs_on = [] { const char* e = getenv("MSCOMP_AMD_XH_SORT"); return e && atoi(e) != 0; }();
		const size_t lh = (xs_on && format == MSCOMP_XPRESS_HUFF) ? (size_t)p->n_chunks * XS_STARTS_STRIDE * sizeof(uint32_t) + 64 : per / 2 + 64;
		const size_t lpad = xs_on ? 2 * XS_FRONT_PAD : 0;
#else
		const size_t lh = per / 2 + 64, lpad = 0;
#endif
		ok = c->links.reserve(per + lpad) && c->mlen3.reserve(2 * per) && c->lasthead.reserve(lh);
		if (ok && format == MSCOMP_XPRESS) {
			const size_t nw = (size_t)p->n_chunks * 1024u + 64;
			ok = c->wtok.reserve(nw * 8) && c->wmat.reserve(nw * 8) && c->wfar.reserve(nw * 4);
			if (ok && xpress_emit_mode_for(p->n_units, p->n_chunks) == 4) {    // the block-per-super-block kernels
				const size_t ns = (size_t)p->n_chunks + 1;
				ok = c->wrec.reserve(nw * 6 * 4) && c->sbrec.reserve(ns * (16 + 24 + 16 * 8 * 4 + 16 * 2 * 8 + 8));
			}
		}
	}
	if (ok && format == MSCOMP_XPRESS_HUFF) {
		const size_t nc = (size_t)p->n_chunks + 1;
		ok = c->tokbits.reserve(nc * 1024 * 8) && c->counts.reserve(nc * 512 * 4) && c->extra.reserve(nc * 4) &&
		     c->lens.reserve(nc * 512) && c->codes.reserve(nc * 1024) && c->fb_list.reserve(nc * 4 + 64) && c->fbflag.reserve(nc * 4);
	}
	if (!ok) { p->tables.release(); delete p; return MSCOMP_MEM_ERROR; }
	*out = p;
	return MSCOMP_OK;
}

MSCompStatus mscomp_amd_plan_create(mscomp_amd_ctx* c, MSCompFormat format, size_t n_units, const uint64_t* in_off, const uint64_t* in_len,
                                    const uint64_t* out_off, const uint64_t* out_cap, mscomp_amd_plan** out)
{
	return plan_create_impl(c, format, false, n_units, in_off, in_len, out_off, out_cap, out);
}
MSCompStatus mscomp_amd_plan_create_decompress(mscomp_amd_ctx* c, MSCompFormat format, size_t n_units, const uint64_t* in_off, const uint64_t* in_len,
                                               const uint64_t* out_off, const uint64_t* out_cap, mscomp_amd_plan** out)
{
	return plan_create_impl(c, format, true, n_units, in_off, in_len, out_off, out_cap, out);
}

void mscomp_amd_plan_destroy(mscomp_amd_plan* p)
{
	if (!p) { return; }
	DeviceGuard g(p->ctx->device);
	(void)hipStreamSynchronize(p->ctx->stream);
	if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); }
	if (p->tables.p) {
		// the pool keeps the 8 LARGEST buffers it has seen: a full pool gives up its smallest one for a larger one (a pool full of
		// small buffers made every larger plan hipFree on destroy -- a device-wide wait)
		std::vector<DevBuf>& pool = p->ctx->table_pool;
		if (pool.size() < 8) { pool.push_back(p->tables); p->tables.p = nullptr; p->tables.cap = 0; }
		else {
			size_t least = 0;
			for (size_t i = 1; i < pool.size(); ++i) { if (pool[i].cap < pool[least].cap) { least = i; } }
			if (pool[least].cap < p->tables.cap) { DevBuf t = pool[least]; pool[least] = p->tables; p->tables = t; }   // (the smaller one is released below)
		}
	}
	p->tables.release(); p->tokpre.release(); p->lzg_tab.release(); p->xps_tab.release();
	delete p;
}

static LzgTables lzg_tables(const mscomp_amd_plan* p, mscomp_amd_ctx* c)
{
	LzgTables g = {};
	g.n_big = p->lzg_big; g.n_tb = p->lzg_tb; g.n_tiles = p->lzg_tiles;
	if (!g.n_big) { return g; }
	const size_t nb = g.n_big, upad = (nb + 1) / 2;
	const uint64_t* t = static_cast<const uint64_t*>(p->lzg_tab.p);
	g.unit = reinterpret_cast<const uint32_t*>(t); g.tb_prefix = t + upad; g.tile_prefix = g.tb_prefix + nb + 1; g.word_prefix = g.tile_prefix + nb + 1;
	g.bsum = static_cast<u64*>(c->lzg_bsum.p);
	g.dir_tok = static_cast<uint32_t*>(c->lzg_dir.p); g.dir_pos = g.dir_tok + p->lzg_tiles;
	g.words = static_cast<uint32_t*>(c->lzg_words.p); g.open = g.words + p->lzg_words; g.tile_pass = reinterpret_cast<uint8_t*>(g.open + LZG_PASSES);
	return g;
}
// the last stage of Xpress / Xpress+Huffman decompression: tokens -> bytes (a wave per small unit, a block per middle one, all CUs for the large ones)
static void run_lz_copy_global(mscomp_amd_ctx* c, const mscomp_amd_plan* p, hipStream_t st, const u64* tp, const uint32_t* tok, const u64* ntok,
                               const u64* d_out_len, const int32_t* d_status, uint8_t* d_out)
{
	if (!p->lzg_big) { return; }
	const LzgTables g = lzg_tables(p, c);
	static const char* const names[3] = {"lzg_dir_kernels", "lzg_expand_kernel", "lzg_jump_kernel"};
	for (int ph = 0; ph < 3; ++ph) { KernelTimer t(c, names[ph]); launch_lz_copy_global(st, g, p->bt, tp, tok, ntok, d_out_len, d_status, d_out, ph); }
}

static XpressWinBufs xpress_win_bufs(mscomp_amd_ctx* c, uint32_t n_chunks)
{
	XpressWinBufs b = {};
	b.wtok = static_cast<u64*>(c->wtok.p); b.wmat = static_cast<u64*>(c->wmat.p); b.wfar = static_cast<uint32_t*>(c->wfar.p);
	const size_t nw = (size_t)n_chunks * 1024u + 64, ns = (size_t)n_chunks + 1;
	if (c->wrec.p && c->sbrec.p) {
		uint32_t* w = static_cast<uint32_t*>(c->wrec.p);
		b.wecur = w; b.weF = w + nw; b.wsum = w + 2 * nw; b.wnr = w + 3 * nw; b.ws0 = w + 4 * nw; b.ws1 = w + 5 * nw;
		uint8_t* q = static_cast<uint8_t*>(c->sbrec.p);
		b.sbpre = reinterpret_cast<u64*>(q); q += ns * 24;
		b.seampos = reinterpret_cast<u64*>(q); q += ns * 16 * 2 * 8;
		b.sbtot = reinterpret_cast<uint32_t*>(q); q += ns * 16;
		b.seam = reinterpret_cast<uint32_t*>(q); q += ns * 16 * 8 * 4;
		b.used = reinterpret_cast<uint32_t*>(q);
	}
	return b;
}

static MSCompStatus plan_launch(mscomp_amd_plan* p, const uint8_t* d_in, uint8_t* d_out, uint64_t* d_out_len, int32_t* d_status)
{
	mscomp_amd_ctx* c = p->ctx;
	hipStream_t st = c->stream;
	uint32_t* slot_size = static_cast<uint32_t*>(c->slot_size.p);
	u64* prefix = static_cast<u64*>(c->prefix.p);
	u64* tile_sums = static_cast<u64*>(c->tile_sums.p);
	if (p->decompress) {
		switch (p->format) {
		case MSCOMP_LZNT1: {
			LzdBufs b;
			b.cin = static_cast<uint32_t*>(c->dz_cin.p); b.csize = static_cast<uint16_t*>(c->dz_csize.p);
			const size_t nk = (size_t)p->n_chunks * LZD_K;
			b.segL = static_cast<uint32_t*>(c->dz_unit.p); b.segE = b.segL + nk; b.segcnt = b.segE + nk; b.segstop = b.segcnt + nk; b.segoff = b.segstop + nk;
			b.selcnt = b.segoff + nk; b.seloff = b.selcnt + p->n_chunks;
			b.stop = b.seloff + p->n_chunks; b.irregular = b.stop + p->n_units + 1u;
			b.flat = prefix;
			{ KernelTimer t(c, "lzd_seg_kernel"); launch_lzd_segments(st, d_in, p->bt, b); }
			{ KernelTimer t(c, "lzd_verify_kernel"); launch_lzd_verify(st, d_in, p->bt, b); }
			{ KernelTimer t(c, "scan_sizes"); launch_scan_sizes(st, b.selcnt, prefix, p->n_chunks, tile_sums); }
			{ KernelTimer t(c, "lzd_chunk_kernel"); launch_lzd_chunks(st, d_in, p->bt, b, d_out, 0); }
			{ KernelTimer t(c, "lzd_finalize_kernel"); launch_lzd_finalize(st, p->bt, b, d_out_len, d_status); }
			{ KernelTimer t(c, "lzd_replace_kernel"); launch_lzd_chunks(st, d_in, p->bt, b, d_out, 1); }
			return MSCOMP_OK;
		}
		case MSCOMP_XPRESS: {
			if (g_xpd_mode.load(std::memory_order_relaxed) == 1) { KernelTimer t(c, "xpd_kernel"); launch_xpress_decompress(st, d_in, p->bt, d_out, d_out_len, d_status); return MSCOMP_OK; }
			const u64* tp = static_cast<const u64*>(p->tokpre.p); uint32_t* tok = static_cast<uint32_t*>(c->dz_tok.p); u64* ntok = static_cast<u64*>(c->dz_ntok.p);
			static const char* const names[3] = {"xpt_parse_kernel", "lz_copy_kernel", "lz_copy_block_kernel"};
			const u64 gmin = p->lzg_big ? (u64)LZG_MIN_CAP : ~(u64)0;
			XpsTables x = {};
			if (p->xps_big && g_xpd_mode.load(std::memory_order_relaxed) != 2) {
				const size_t nb = p->xps_big, upad = (nb + 1) / 2;
				const uint64_t* t = static_cast<const uint64_t*>(p->xps_tab.p);
				x.unit = reinterpret_cast<const uint32_t*>(t); x.seg_prefix = t + upad; x.n_big = p->xps_big; x.n_seg = p->xps_seg; x.seg_bytes = p->xps_seg_bytes; x.warm_bytes = p->xps_warm_bytes;
				x.seg = c->xps_buf.p; x.mode = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(c->xps_buf.p) + (size_t)p->xps_seg * XPS_SEG_BYTES); x.done = x.mode + nb;
				{ KernelTimer t2(c, "xps_walk_kernel"); launch_xpress_decompress_tokens(st, d_in, p->bt, tp, tok, ntok, d_out, d_out_len, d_status, -1, gmin, x); }
				static const uint32_t rounds = [] { const char* e = getenv("MSCOMP_AMD_XPS_ROUNDS"); const long v = e ? atol(e) : 0; return (uint32_t)(v >= 1 && v <= 256 ? v : XPS_ROUNDS); }();
				for (uint32_t r = 0; r < rounds; ++r) { KernelTimer t2(c, "xps_redo_kernels"); launch_xpress_decompress_tokens(st, d_in, p->bt, tp, tok, ntok, d_out, d_out_len, d_status, -2, gmin, x); }
				{ KernelTimer t2(c, "xps_emit_kernel"); launch_xpress_decompress_tokens(st, d_in, p->bt, tp, tok, ntok, d_out, d_out_len, d_status, -3, gmin, x); }
			}
			for (int ph = 0; ph < 3; ++ph) { KernelTimer t(c, names[ph]); launch_xpress_decompress_tokens(st, d_in, p->bt, tp, tok, ntok, d_out, d_out_len, d_status, ph, gmin, x); }
			run_lz_copy_global(c, p, st, tp, tok, ntok, d_out_len, d_status, d_out);
			return MSCOMP_OK;
		}
		case MSCOMP_XPRESS_HUFF: {
			const u64* tp = static_cast<const u64*>(p->tokpre.p); uint32_t* tok = static_cast<uint32_t*>(c->dz_tok.p); u64* ntok = static_cast<u64*>(c->dz_ntok.p);
			const u64* cp = tp + (p->n_units + 1u);
			XhcBufs xb = {};
			if (p->xhc_scr) { xb.scr_prefix = tp + 2 * (p->n_units + 1u); xb.scr_tok = static_cast<uint32_t*>(c->dz_scr.p); }
			{
				const size_t nu = (size_t)p->n_units + 1, ns = p->xhc_slots;
				uint8_t* q = static_cast<uint8_t*>(c->dz_xhc.p);
				xb.res_prod = reinterpret_cast<u64*>(q); q += ns * 8; xb.res_ntok = reinterpret_cast<u64*>(q); q += ns * 8; xb.tok_off = reinterpret_cast<u64*>(q); q += ns * 8;
				xb.cand_pos = reinterpret_cast<uint32_t*>(q); q += ns * 4; xb.res_end = reinterpret_cast<uint32_t*>(q); q += ns * 4;
				xb.res_reach = reinterpret_cast<uint32_t*>(q); q += ns * 4; xb.res_state = reinterpret_cast<uint32_t*>(q); q += ns * 4;
				xb.cand_cnt = reinterpret_cast<uint32_t*>(q); q += nu * 4; xb.mode = reinterpret_cast<uint32_t*>(q);
			}
			static const char* const names[6] = { "xhc_mark_kernel", "xhc_parse_kernel", "xhc_chain_kernel", "xhc_parse2_kernel", "xhd_parse_kernel", "lz_copy_kernel" };
			for (int ph = 0; ph < 6; ++ph) {
				KernelTimer t(c, names[ph]);
				launch_xpress_huff_decompress(st, d_in, p->bt, tp, tok, ntok, cp, p->xhc_slots, xb, d_out, d_out_len, d_status, ph, p->lzg_big ? (u64)LZG_MIN_CAP : ~(u64)0);
			}
			run_lz_copy_global(c, p, st, tp, tok, ntok, d_out_len, d_status, d_out);
			return MSCOMP_OK;
		}
		default:
			return MSCOMP_ARG_ERROR;
		}
	}
	switch (p->format) {
	case MSCOMP_LZNT1: {
		uint8_t* slots = static_cast<uint8_t*>(c->slots.p);
		if (p->lznt1_sa) { KernelTimer t(c, "lznt1_sa_chunk_kernel"); launch_lznt1_sa_chunks(st, d_in, p->bt, slots, slot_size); }
		else { KernelTimer t(c, "lznt1_chunk_kernel"); launch_lznt1_chunks(st, d_in, p->bt, slots, slot_size); }
		{ KernelTimer t(c, "scan_sizes"); launch_scan_sizes(st, slot_size, prefix, p->n_chunks, tile_sums); }
		{ KernelTimer t(c, "concat_slots_kernel"); launch_concat_slots(st, slots, LZNT1_SLOT, slot_size, prefix, p->bt, d_out); }
		{ KernelTimer t(c, "finalize_units_kernel"); launch_finalize_units(st, prefix, p->bt, d_out, d_out_len, d_status, 1); }
		break;
	}
	case MSCOMP_XPRESS: {
		uint16_t* links = static_cast<uint16_t*>(c->links.p); uint16_t* lasthead = static_cast<uint16_t*>(c->lasthead.p);
		uint16_t* mlen3 = static_cast<uint16_t*>(c->mlen3.p); uint16_t* moff = mlen3 + 1;   /* one word per position: length - 3 | offset << 16 (common.h S16) */
		if (p->matches_ready) { p->matches_ready = false; }
		else {
			{ KernelTimer t(c, "xp_links_kernel"); launch_xp_links(st, d_in, p->bt, links, lasthead); }
			// units of one link chunk: Find only where a greedy parse can start a token, window and links in LDS; longer streams: every position
			if (g_finder_mode.load(std::memory_order_relaxed) != 2 && p->max_unit <= 65536u) { KernelTimer t(c, "xp_lazy2_kernel"); launch_xp_lazy2(st, d_in, p->bt, links, mlen3, moff); }
			else { KernelTimer t(c, "xp_find_kernel"); launch_xp_find(st, d_in, p->bt, links, lasthead, mlen3, moff, 0x2000u, 0); }
		}
		{ KernelTimer t(c, "xpress_emit_kernel"); launch_xpress_emit(st, d_in, p->bt, mlen3, moff, xpress_win_bufs(c, p->n_chunks), d_out, d_out_len, d_status); }
		break;
	}
	case MSCOMP_XPRESS_HUFF: {
		uint16_t* links = static_cast<uint16_t*>(c->links.p); uint16_t* lasthead = static_cast<uint16_t*>(c->lasthead.p);
		uint16_t* mlen3 = static_cast<uint16_t*>(c->mlen3.p); uint16_t* moff = mlen3 + 1;   /* one word per position: length - 3 | offset << 16 (common.h S16) */
		u64* tokbits = static_cast<u64*>(c->tokbits.p); uint32_t* counts = static_cast<uint32_t*>(c->counts.p);
		uint32_t* extra = static_cast<uint32_t*>(c->extra.p); uint8_t* lens = static_cast<uint8_t*>(c->lens.p);
		uint16_t* codes = static_cast<uint16_t*>(c->codes.p); uint32_t* fbflag = static_cast<uint32_t*>(c->fbflag.p);
		uint32_t* fb_count = static_cast<uint32_t*>(c->fb_list.p); uint32_t* fb_list = fb_count + 16;
#ifdef MSCOMP_AMD_DEV
		static const bool xh_lazy = [] { const char* e = getenv("MSCOMP_AMD_XH_LAZY"); return e && atoi(e) != 0; }();   // measurement switch: the lazy finder of xhuff_lazy.hip
		static const int xh_sort = [] { const char* e = getenv("MSCOMP_AMD_XH_SORT"); return e ? atoi(e) : 0; }();      // the finder of xpress_sort.hip (sorted spans instead of the chain walk)
		if (xh_sort && !xh_lazy) {
			{ KernelTimer t(c, "xp_sort_kernel"); launch_xp_sort(st, d_in, p->bt, links + XS_FRONT_PAD, reinterpret_cast<uint32_t*>(mlen3), reinterpret_cast<uint32_t*>(lasthead)); }
			{ KernelTimer t(c, "xp_find2_kernel"); launch_xp_find2(st, d_in, p->bt, links + XS_FRONT_PAD, reinterpret_cast<const uint32_t*>(lasthead), reinterpret_cast<uint32_t*>(mlen3), 0xFFFFu, 1); }
		} else if (xh_lazy) {
			{ KernelTimer t(c, "xp_links_kernel"); launch_xp_links(st, d_in, p->bt, links, lasthead); }
			{ KernelTimer t(c, "xh_lazy_kernel"); launch_xh_lazy(st, d_in, p->bt, links, lasthead, mlen3); }
		} else
#endif
		if (p->matches_ready) { p->matches_ready = false; }
		else {
			{ KernelTimer t(c, "xp_links_kernel"); launch_xp_links(st, d_in, p->bt, links, lasthead); }
			{ KernelTimer t(c, "xp_find_kernel"); launch_xp_find(st, d_in, p->bt, links, lasthead, mlen3, moff, 0xFFFFu, 1); }
		}
		{ KernelTimer t(c, "xh_parse_kernel"); launch_xh_parse(st, d_in, p->bt, mlen3, moff, tokbits, counts, extra); }
		{ KernelTimer t(c, "xh_huff_kernel"); launch_xh_huff(st, p->bt, counts, extra, lens, codes, slot_size, fb_list, fb_count, fbflag); }
		{ KernelTimer t(c, "xh_fallback_kernel"); launch_xh_fallback(st, d_in, p->bt, fb_list, fb_count, XH_FB_BLOCKS, tokbits, lens, codes, slot_size); }
		{ KernelTimer t(c, "scan_sizes"); launch_scan_sizes(st, slot_size, prefix, p->n_chunks, tile_sums); }
		{ KernelTimer t(c, "xh_encode_kernel"); launch_xh_encode(st, d_in, p->bt, mlen3, moff, tokbits, lens, codes, fbflag, prefix, d_out); }
		{ KernelTimer t(c, "finalize_units_kernel"); launch_finalize_units(st, prefix, p->bt, d_out, d_out_len, d_status, 0); }
		break;
	}
	default:
		return MSCOMP_ARG_ERROR;
	}
	return MSCOMP_OK;
}

MSCompStatus mscomp_amd_plan_execute(mscomp_amd_plan* p, const uint8_t* d_in, uint8_t* d_out, uint64_t* d_out_len, int32_t* d_status)
{
	if (!p || (p->n_units && (!d_out_len || !d_status)) || (p->total_in && !d_in)) { return MSCOMP_ARG_ERROR; }
	mscomp_amd_ctx* c = p->ctx;
	DeviceGuard g(c->device);
	if (!g.ok) { return MSCOMP_ERRNO; }
	static const bool no_graph = getenv("MSCOMP_AMD_NO_GRAPH") != nullptr;
	++p->executions;
	// A plan that is executed repeatedly replays its 4-9 launches as one hipGraph (the gaps between the launches are
	// ~3 % of an LZNT1 pass). Not while profiling (the per-kernel events are not part of the graph), not on the first
	// execution (one-time function attributes are set there).
	if (!no_graph && !p->no_graph && !c->profiling && p->executions >= 2 && p->n_units) {
		const uint64_t mode_now = g_mode_epoch.load(std::memory_order_acquire);
		const bool same = p->gexec && p->g_epoch == c->epoch && p->g_mode == mode_now && p->g_args[0] == d_in && p->g_args[1] == d_out &&
		                  p->g_args[2] == d_out_len && p->g_args[3] == d_status;
		if (!same) {
			if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
			hipGraph_t graph = nullptr;
			if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
				const MSCompStatus ls = plan_launch(p, d_in, d_out, d_out_len, d_status);
				const hipError_t ee = hipStreamEndCapture(c->stream, &graph);
				if (ls == MSCOMP_OK && ee == hipSuccess && graph && hipGraphInstantiate(&p->gexec, graph, nullptr, nullptr, 0) == hipSuccess) {
					p->g_epoch = c->epoch; p->g_mode = mode_now; p->g_args[0] = d_in; p->g_args[1] = d_out; p->g_args[2] = d_out_len; p->g_args[3] = d_status;
				} else { p->gexec = nullptr; }
				if (graph) { (void)hipGraphDestroy(graph); }
			}
			(void)hipGetLastError();
		}
		if (p->gexec) { return hipGraphLaunch(p->gexec, c->stream) == hipSuccess ? MSCOMP_OK : MSCOMP_ERRNO; }
	}
	const MSCompStatus ls = plan_launch(p, d_in, d_out, d_out_len, d_status);
	if (ls != MSCOMP_OK) { return ls; }
	return hipGetLastError() == hipSuccess ? MSCOMP_OK : MSCOMP_ERRNO;
}

MSCompStatus mscomp_amd_compress_batch(mscomp_amd_ctx* c, MSCompFormat format, size_t n_units,
                                       const uint8_t* d_in, const uint64_t* in_off, const uint64_t* in_len,
                                       uint8_t* d_out, const uint64_t* out_off, const uint64_t* out_cap,
                                       uint64_t* d_out_len, int32_t* d_status)
{
	mscomp_amd_plan* p = nullptr;
	MSCompStatus s = mscomp_amd_plan_create(c, format, n_units, in_off, in_len, out_off, out_cap, &p);
	if (s != MSCOMP_OK) { return s; }
	s = mscomp_amd_plan_execute(p, d_in, d_out, d_out_len, d_status);
	DeviceGuard g(c->device);
	if (hipStreamSynchronize(c->stream) != hipSuccess && s == MSCOMP_OK) { s = MSCOMP_ERRNO; }
	mscomp_amd_plan_destroy(p);
	return s;
}

MSCompStatus mscomp_amd_decompress_batch(mscomp_amd_ctx* c, MSCompFormat format, size_t n_units,
                                       const uint8_t* d_in, const uint64_t* in_off, const uint64_t* in_len,
                                       uint8_t* d_out, const uint64_t* out_off, const uint64_t* out_cap,
                                       uint64_t* d_out_len, int32_t* d_status)
{
	mscomp_amd_plan* p = nullptr;
	MSCompStatus s = mscomp_amd_plan_create_decompress(c, format, n_units, in_off, in_len, out_off, out_cap, &p);
	if (s != MSCOMP_OK) { return s; }
	s = mscomp_amd_plan_execute(p, d_in, d_out, d_out_len, d_status);
	DeviceGuard g(c->device);
	if (hipStreamSynchronize(c->stream) != hipSuccess && s == MSCOMP_OK) { s = MSCOMP_ERRNO; }
	mscomp_amd_plan_destroy(p);
	return s;
}

// Stage-level test hook: per-position (len-3 capped at 45, offset) of ONE unit as found by the HIP match finder.
// h_len3/h_off: host arrays of in_len u16. max_off 0x2000 (Xpress) or 0xFFFF (Xpress+Huffman, clip to 64 KiB chunks).
// ---- batch helpers: capacity planning and device-side compaction ----
uint64_t mscomp_amd_plan_layout(MSCompFormat format, size_t n_units, const uint64_t* in_len, uint64_t align, uint64_t* out_off, uint64_t* out_cap)
{
	if (!align) { align = 1; }
	uint64_t pos = 0;
	for (size_t i = 0; i < n_units; ++i) {
		size_t cap;
		switch ((int)format) {
		case MSCOMP_LZNT1:       cap = lznt1_max_compressed_size((size_t)in_len[i]) + 2; break;   // room for the uncounted End_of_buffer
		case MSCOMP_XPRESS:      cap = xpress_max_compressed_size((size_t)in_len[i]); break;
		case MSCOMP_XPRESS_HUFF: cap = xpress_huff_max_compressed_size((size_t)in_len[i]); break;
		case MSCOMP_NONE:        cap = (size_t)in_len[i]; break;
		default:                 return (uint64_t)-1;
		}
		if (out_off) { out_off[i] = pos; }
		if (out_cap) { out_cap[i] = cap; }
		pos += (cap + align - 1) / align * align;
	}
	return pos;
}

MSCompStatus mscomp_amd_compact_batch(mscomp_amd_ctx* c, size_t n_units, const uint8_t* d_out, const uint64_t* out_off, const uint64_t* out_cap,
                                      const uint64_t* d_out_len, uint8_t* d_packed, uint64_t* d_packed_off)
{
	if (!c || !d_packed_off || (n_units && (!d_out || !out_off || !out_cap || !d_out_len || !d_packed)) || n_units > 0x7FFFFFF0u) { return MSCOMP_ARG_ERROR; }
	DeviceGuard g(c->device);
	if (!g.ok) { return MSCOMP_ERRNO; }
	if (n_units == 0) { return hipMemsetAsync(d_packed_off, 0, 8, c->stream) == hipSuccess ? MSCOMP_OK : MSCOMP_ERRNO; }
	std::vector<uint64_t> host(n_units + (n_units + 2) / 2 + 1);
	uint32_t* tp = reinterpret_cast<uint32_t*>(host.data() + n_units);
	uint64_t tiles = 0;
	for (size_t i = 0; i < n_units; ++i) {
		host[i] = out_off[i]; tp[i] = (uint32_t)tiles;
		tiles += (out_cap[i] + 65535u) / 65536u;
		if (tiles > 0x7FFFFFF0u) { return MSCOMP_ARG_ERROR; }
	}
	tp[n_units] = (uint32_t)tiles;
	const size_t bytes = host.size() * sizeof(uint64_t);
	if (!c->cp_tab.reserve(bytes)) { return MSCOMP_MEM_ERROR; }
	if (hipMemcpyAsync(c->cp_tab.p, host.data(), bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { return MSCOMP_ERRNO; }
	const u64* d_off = static_cast<const u64*>(c->cp_tab.p);
	launch_compact(c->stream, d_out, d_off, reinterpret_cast<const uint32_t*>(d_off + n_units), (uint32_t)n_units, (uint32_t)tiles, d_out_len, d_packed_off, d_packed);
	return hipGetLastError() == hipSuccess ? MSCOMP_OK : MSCOMP_ERRNO;
}

MSCompStatus mscomp_amd_debug_xpress_matches(mscomp_amd_ctx* c, const uint8_t* d_in, size_t in_len, uint32_t max_off, int clip,
                                             uint16_t* h_len3, uint16_t* h_off)
{
	if (!c) { return MSCOMP_ARG_ERROR; }
	const uint64_t z = 0, len = in_len, cap = 0;
	mscomp_amd_plan* p = nullptr;
	MSCompStatus s = mscomp_amd_plan_create(c, MSCOMP_XPRESS, 1, &z, &len, &z, &cap, &p);
	if (s != MSCOMP_OK) { return s; }
	DeviceGuard g(c->device);
	uint16_t* links = static_cast<uint16_t*>(c->links.p); uint16_t* lasthead = static_cast<uint16_t*>(c->lasthead.p);
	uint16_t* mlen3 = static_cast<uint16_t*>(c->mlen3.p); uint16_t* moff = mlen3 + 1;   /* one word per position: length - 3 | offset << 16 (common.h S16) */
	launch_xp_links(c->stream, d_in, p->bt, links, lasthead);
	launch_xp_find(c->stream, d_in, p->bt, links, lasthead, mlen3, moff, max_off, clip);
	bool ok = hipStreamSynchronize(c->stream) == hipSuccess;
	std::vector<uint32_t> words(in_len + 1);
	ok = ok && hipMemcpy(words.data(), mlen3, in_len * 4, hipMemcpyDeviceToHost) == hipSuccess;
	for (size_t i = 0; ok && i < in_len; ++i) { h_len3[i] = (uint16_t)words[i]; h_off[i] = (uint16_t)(words[i] >> 16); }
	mscomp_amd_plan_destroy(p);
	return ok ? MSCOMP_OK : MSCOMP_ERRNO;
}

// Stage-level test hook: HuffmanEncoder<15,512>::CreateCodes (HuffmanEncoder.h:58-107) for n histograms of 512 counts (host arrays)
MSCompStatus mscomp_amd_debug_huff_lengths(mscomp_amd_ctx* c, const uint32_t* h_counts, size_t n, uint8_t* h_lens)
{
	if (!c || (n && (!h_counts || !h_lens)) || n > 0x100000u) { return MSCOMP_ARG_ERROR; }
	if (!n) { return MSCOMP_OK; }
	DeviceGuard g(c->device);
	if (!g.ok || !c->counts.reserve(n * 2048) || !c->lens.reserve(n * 512)) { return MSCOMP_MEM_ERROR; }
	bool ok = hipMemcpyAsync(c->counts.p, h_counts, n * 2048, hipMemcpyHostToDevice, c->stream) == hipSuccess;
	if (ok) { launch_xh_huff_debug(c->stream, static_cast<const uint32_t*>(c->counts.p), static_cast<uint8_t*>(c->lens.p), (uint32_t)n); }
	ok = ok && hipMemcpyAsync(h_lens, c->lens.p, n * 512, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
	return ok ? MSCOMP_OK : MSCOMP_ERRNO;
}

int mscomp_amd_debug_hooks_enabled(void) { return test_hooks_on() ? 1 : 0; }
// Test hook: which Xpress parse/emit kernel runs (0 = chosen by batch size, 1 = one wave per unit, 2 = four waves per unit).
void mscomp_amd_debug_set_xpress_emit(int mode) { if (!test_hooks_on()) { return; } set_xpress_emit_mode(mode); g_mode_epoch.fetch_add(1, std::memory_order_acq_rel); }
// ... and which LZNT1 chunk kernel (0 = default, 1 = one wave per chunk, 2 = four waves per chunk).
uint32_t mscomp_amd_debug_lzd_walked(mscomp_amd_ctx* c)
{
	if (!c) { return 0xFFFFFFFFu; }
	DeviceGuard g(c->device);
	if (!g.ok || hipStreamSynchronize(c->stream) != hipSuccess) { return 0xFFFFFFFFu; }
	return lzd_read_walked();
}

// The LZNT1 dictionary flavour is a property of a PLAN, fixed when the plan is created (the reference fixes it when the library is built,
// config.h:83-88): from its context's setting (mscomp_amd_ctx_set_lznt1_sa_dict) or, when the context has none, from the process default below.
// Changing either never touches a plan that exists, so callers that run plans concurrently keep their bytes. The one-shot entries (ms_compress,
// ms_deflate: no context argument) and the host-batch entry create their plans per call and follow the process default of that moment.
void mscomp_amd_set_lznt1_sa_dict(int on) { g_lznt1_sa.store(on ? 1 : 0, std::memory_order_relaxed); }
int  mscomp_amd_get_lznt1_sa_dict(void) { return g_lznt1_sa.load(std::memory_order_relaxed); }
MSCompStatus mscomp_amd_ctx_set_lznt1_sa_dict(mscomp_amd_ctx* c, int on)
{
	if (!c || on < -1 || on > 1) { return MSCOMP_ARG_ERROR; }
	c->lznt1_sa = on;
	return MSCOMP_OK;
}
int mscomp_amd_debug_lzg_open(mscomp_amd_ctx* c, uint64_t words, uint32_t* out)
{	// test hook: the open-word counters of the last lzglobal.hip run (`words` = the plan's word count: sum of capacity + 64 over the units taken)
	if (!c || !out || !c->lzg_words.p) { return -1; }
	DeviceGuard g(c->device);
	if (!g.ok || hipStreamSynchronize(c->stream) != hipSuccess) { return -1; }
	return hipMemcpy(out, static_cast<uint32_t*>(c->lzg_words.p) + words, LZG_PASSES * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
void mscomp_amd_debug_set_xpress_decoder(int mode) { if (!test_hooks_on()) { return; } g_xpd_mode.store(mode, std::memory_order_relaxed); g_mode_epoch.fetch_add(1, std::memory_order_acq_rel); }
void mscomp_amd_debug_set_finder(int mode) { if (!test_hooks_on()) { return; } g_finder_mode.store(mode, std::memory_order_relaxed); g_mode_epoch.fetch_add(1, std::memory_order_acq_rel); }
void mscomp_amd_debug_set_one_shot(int mode) { if (!test_hooks_on()) { return; } g_one_zero_copy.store(mode == 1 ? 0 : 1, std::memory_order_relaxed); }
void mscomp_amd_debug_set_lznt1(int mode) { if (!test_hooks_on()) { return; } set_lznt1_mode(mode); g_mode_epoch.fetch_add(1, std::memory_order_acq_rel); }
// Test hook: 1 = the order-independent form of the LZNT1 bucket sort and the Xpress chain links on every device (what a device that fails the
// lane-order self-check gets), 0 = back to one atomic per 64 positions
void mscomp_amd_debug_set_serial_atomics(int on) { if (!test_hooks_on()) { return; } set_serial_atomics(-1, on); g_mode_epoch.fetch_add(1, std::memory_order_acq_rel); }

// Hardware self-check (see util.hip): lanes whose returning LDS atomic was NOT served in lane order, summed over
// blocks x rounds x 64 lanes x {add, exchange}; 0 on gfx950. 0xFFFFFFFF = the check could not run.
uint32_t mscomp_amd_debug_lds_lane_order(mscomp_amd_ctx* c, uint32_t seed, uint32_t blocks, uint32_t rounds, uint32_t nkeys)
{
	if (!c || nkeys == 0 || nkeys > 2048u) { return 0xFFFFFFFFu; }
	DeviceGuard g(c->device);
	if (!c->slot_size.reserve(64)) { return 0xFFFFFFFFu; }
	return run_lds_lane_order_check(c->stream, seed, blocks, rounds, nkeys, static_cast<uint32_t*>(c->slot_size.p));
}

// ---- drop-in one-shot path (host pointers): H2D, one-unit batch on the GPU, D2H. No CPU encoder exists here. ----
namespace {

// One context per calling host thread (the reference is reentrant: SURVEY.md 8b "Threading"), destroyed with its thread. The
// plans of recent calls are kept: a caller that compresses buffers of one size (pages, records, fixed blocks) pays the table
// upload once and gets the captured hipGraph from its second call on.
struct OneShotTls {
	mscomp_amd_ctx* ctx = nullptr;
	hipStream_t exec = nullptr, h2d = nullptr, d2h = nullptr;   // non-blocking streams: kernels, uploads, downloads
	std::vector<hipEvent_t> ev;                         // two per slice of a pipelined call
	uint64_t* h_len = nullptr; int32_t* h_st = nullptr; size_t h_cap = 0;   // pinned: per-slice results
	DevBuf d_meta;
	struct Entry { MSCompFormat format; bool decompress; uint64_t in_len, out_cap; mscomp_amd_plan* plan; uint64_t used; };
	std::vector<Entry> plans;
	uint64_t tick = 0;
	void drop()
	{
		for (auto& e : plans) { mscomp_amd_plan_destroy(e.plan); }
		plans.clear();
		if (ctx) { mscomp_amd_ctx_destroy(ctx); ctx = nullptr; }
		for (auto e : ev) { (void)hipEventDestroy(e); }
		ev.clear();
		if (h_len) { (void)hipHostFree(h_len); h_len = nullptr; h_st = nullptr; h_cap = 0; }
		d_meta.release();
		if (exec) { (void)hipStreamDestroy(exec); exec = nullptr; }
		if (h2d) { (void)hipStreamDestroy(h2d); h2d = nullptr; }
		if (d2h) { (void)hipStreamDestroy(d2h); d2h = nullptr; }
	}
	bool slices(size_t n)                               // events, pinned result slots and device result slots for n slices
	{
		while (ev.size() < 2 * n) { hipEvent_t e = nullptr; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { return false; } ev.push_back(e); }
		if (h_cap < n) {
			if (h_len) { (void)hipHostFree(h_len); h_len = nullptr; h_cap = 0; }
			void* q = nullptr;
			if (hipHostMalloc(&q, n * 16, hipHostMallocDefault) != hipSuccess) { return false; }
			h_len = static_cast<uint64_t*>(q); h_st = reinterpret_cast<int32_t*>(h_len + n); h_cap = n;
		}
		return d_meta.reserve(n * 16);
	}
	~OneShotTls() { drop(); }
	MSCompStatus plan_for(MSCompFormat f, bool dec, uint64_t in_len, uint64_t out_cap, mscomp_amd_plan** out)
	{
		for (auto& e : plans) {
			if (e.format == f && e.decompress == dec && e.in_len == in_len && e.out_cap == out_cap && e.plan->lznt1_sa == (!dec && f == MSCOMP_LZNT1 && lznt1_sa_for(ctx))) { e.used = ++tick; *out = e.plan; return MSCOMP_OK; }
		}
		const uint64_t z = 0;
		mscomp_amd_plan* p = nullptr;
		const MSCompStatus s = plan_create_impl(ctx, f, dec, 1, &z, &in_len, &z, &out_cap, &p);
		if (s != MSCOMP_OK) { return s; }
		p->no_graph = true;
		if (plans.size() >= 8) {
			size_t lru = 0;
			for (size_t i = 1; i < plans.size(); ++i) { if (plans[i].used < plans[lru].used) { lru = i; } }
			mscomp_amd_plan_destroy(plans[lru].plan);
			plans.erase(plans.begin() + (long)lru);
		}
		plans.push_back(Entry{ f, dec, in_len, out_cap, p, ++tick });
		*out = p;
		return MSCOMP_OK;
	}
};

} // namespace

// Page-locking of caller buffers for the large host-pointer calls. hipHostRegister refuses a range that overlaps a registered one, and a range
// one thread unregisters while another thread's copy of the same pages is in flight would pull the pages from under that copy (two threads may
// legally compress the SAME input at once: the reference only forbids aliasing in / out). So registrations go through this table: the same range
// is shared (counted), an overlapping but different range waits until the other call has released its own, and a range that cannot be registered
// at all (locked-memory limits) is simply not pinned -- the copies then go through the runtime's staging buffers.
struct HostPins {
	struct Ent { const uint8_t* b; size_t n; int refs; std::vector<std::thread::id> owners; };
	std::mutex mu; std::condition_variable cv; std::vector<Ent> ents;
	bool pin(const void* ptr, size_t n)                       // true: the range is page-locked (and mapped) until unpin
	{
		if (!ptr || !n) { return false; }
		const uint8_t* b = static_cast<const uint8_t*>(ptr);
		const std::thread::id me = std::this_thread::get_id();
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			bool overlap = false;
			for (auto& e : ents) {
				if (e.b == b && e.n == n) { ++e.refs; e.owners.push_back(me); return true; }
				if (b < e.b + e.n && e.b < b + n) {
					// a different range that overlaps one THIS thread holds (a call whose in and out overlap): waiting would wait for ourselves --
					// the range is simply not pinned and its copies go through the runtime's staging buffers
					for (auto& o : e.owners) { if (o == me) { return false; } }
					overlap = true;
				}
			}
			if (!overlap) { break; }
			cv.wait(lk);
		}
		// (hipHostRegister of tens of megabytes takes milliseconds and runs under the table's lock: large one-shot calls of different threads
		// register one after the other; their copies and kernels still overlap)
		if (hipHostRegister(const_cast<uint8_t*>(b), n, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return false; }
		ents.push_back(Ent{ b, n, 1, { me } });
		return true;
	}
	// A call that does NOT pin its input (the staged paths) but reads a range another thread's call has registered: the runtime then copies
	// from it as from pinned memory (asynchronously), so the registration must outlive that copy. hold() takes a reference on every entry that
	// overlaps the range (the entries are returned and given back with release() once the copy is done).
	std::vector<std::pair<const uint8_t*, size_t>> hold(const void* ptr, size_t n)
	{
		std::vector<std::pair<const uint8_t*, size_t>> held;
		if (!ptr || !n) { return held; }
		const uint8_t* b = static_cast<const uint8_t*>(ptr);
		std::unique_lock<std::mutex> lk(mu);
		for (auto& e : ents) { if (b < e.b + e.n && e.b < b + n) { ++e.refs; held.push_back({ e.b, e.n }); } }
		return held;
	}
	void release(const std::vector<std::pair<const uint8_t*, size_t>>& held)
	{
		std::unique_lock<std::mutex> lk(mu);
		for (auto& h : held) {
			for (size_t i = 0; i < ents.size(); ++i) {
				if (ents[i].b == h.first && ents[i].n == h.second) {
					if (--ents[i].refs == 0) { (void)hipHostUnregister(const_cast<uint8_t*>(ents[i].b)); ents.erase(ents.begin() + (long)i); cv.notify_all(); }
					break;
				}
			}
		}
	}
	void unpin(const void* ptr, size_t n)
	{
		const uint8_t* b = static_cast<const uint8_t*>(ptr);
		const std::thread::id me = std::this_thread::get_id();
		std::unique_lock<std::mutex> lk(mu);
		for (size_t i = 0; i < ents.size(); ++i) {
			if (ents[i].b == b && ents[i].n == n) {
				for (size_t k = 0; k < ents[i].owners.size(); ++k) { if (ents[i].owners[k] == me) { ents[i].owners.erase(ents[i].owners.begin() + (long)k); break; } }
				if (--ents[i].refs == 0) { (void)hipHostUnregister(const_cast<uint8_t*>(b)); ents.erase(ents.begin() + (long)i); cv.notify_all(); }
				return;
			}
		}
	}
};
static HostPins g_pins;
} // extern "C" (the two functions below are C++: they belong to the library's other translation units)
namespace msc {
// for the other host paths of the library (hostbatch.hip): keep every registration that overlaps [ptr, ptr + n) alive until the returned
// token is given back (nullptr: nothing overlapped)
void* host_pins_hold(const void* ptr, size_t n)
{
	auto v = g_pins.hold(ptr, n);
	if (v.empty()) { return nullptr; }
	return new (std::nothrow) std::vector<std::pair<const uint8_t*, size_t>>(std::move(v));
}
void host_pins_release(void* token)
{
	if (!token) { return; }
	auto* v = static_cast<std::vector<std::pair<const uint8_t*, size_t>>*>(token);
	g_pins.release(*v);
	delete v;
}
}
extern "C" {

// one large host buffer through the GPU with the link busy in both directions while the kernels run. LZNT1 chunks
// are independent and the output is their concatenation (lznt1_compress.cpp:262), so the buffer is cut into slices of 1024 chunks:
// slice k + 1 is on its way up (stream h2d) and slice k - 1 on its way down (stream d2h) while slice k is compressed (the
// context's stream); a slice's place in the caller's output is known once the sizes of the slices before it are back (16 bytes per
// slice into pinned memory). The caller's buffers are page-locked for the duration of the call when the runtime allows it
// (hipHostRegister: uploads and downloads then really overlap; measured on the MI355X box: 85 MB both ways at once 0.92 ms pinned,
// 1.7 ms pageable); without it the copies still work.
// Slices of about 24 MiB (6 144 chunks: three rounds of the chunk kernel's 2 048 resident blocks; with 4 MiB slices the kernels ran at a
// third of their batch rate and the call took 4.9 instead of 2.7 ms), equal in size, at least two.
static const size_t ONE_RANGED_MIN = [] { const char* e = getenv("MSCOMP_AMD_ONE_RANGED_MIN_MB"); const long v = e ? atol(e) : 8; return (size_t)(v >= 1 && v <= 65536 ? v : 8) << 20; }();   // Xpress / Xpress+Huffman one-shot calls from this size on go up in ranges (one_shot)
static const uint32_t ONE_RANGE_CHUNKS = [] { const char* e = getenv("MSCOMP_AMD_ONE_RANGE_CHUNKS"); const long v = e ? atol(e) : 256; return (uint32_t)(v >= 1 && v <= 65536 ? v : 256); }();   // 64 KiB chunks per range (256 = 16 MiB; 51 MB, ms per call Xpress / Xpress+Huffman: 32: 5.22 / 5.85, 64: 4.38 / 4.99, 128: 4.04 / 4.65, 256: 3.91 / 4.55, one copy: 4.23 / 4.92)
static const size_t ONE_SLICE = [] { const char* e = getenv("MSCOMP_AMD_ONE_SLICE_MB"); const long v = e ? atol(e) : 24; return (size_t)(v >= 1 && v <= 4096 ? v : 24) << 20; }();   // (a value that is no number of MiB in 1..4096 is ignored)
static MSCompStatus lznt1_compress_pipelined(OneShotTls& tls, const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len)
{
	mscomp_amd_ctx* c = tls.ctx;
	const size_t cap = *out_len;
	const size_t ns = (in_len + ONE_SLICE / 2) / ONE_SLICE < 2 ? 2 : (in_len + ONE_SLICE / 2) / ONE_SLICE;
	const size_t SL = ((in_len + ns - 1) / ns + 4095) & ~(size_t)4095;          // bytes per slice: whole chunks
	const size_t scap = (lznt1_max_compressed_size(SL) + 2 + 15) & ~(size_t)15;
	if (!tls.slices(ns) || !c->one_in.reserve(in_len + 64) || !c->one_out.reserve(ns * scap + 64)) { return MSCOMP_MEM_ERROR; }
	uint8_t* d_in = static_cast<uint8_t*>(c->one_in.p); uint8_t* d_out = static_cast<uint8_t*>(c->one_out.p);
	uint64_t* d_len = static_cast<uint64_t*>(tls.d_meta.p); int32_t* d_st = reinterpret_cast<int32_t*>(d_len + ns);
	const size_t most = lznt1_max_compressed_size(in_len) + 2, out_span = cap < most ? cap : most;
	const bool reg_in = g_pins.pin(in, in_len);
	const bool reg_out = out_span && g_pins.pin(out, out_span);
	MSCompStatus rs = MSCOMP_OK;
	for (size_t k = 0; k < ns && rs == MSCOMP_OK; ++k) {
		const size_t o = k * SL, len = in_len - o < SL ? in_len - o : SL;
		if (hipMemcpyAsync(d_in + o, in + o, len, hipMemcpyHostToDevice, tls.h2d) != hipSuccess || hipEventRecord(tls.ev[2 * k], tls.h2d) != hipSuccess) { rs = MSCOMP_ERRNO; }
	}
	for (size_t k = 0; k < ns && rs == MSCOMP_OK; ++k) {
		const size_t o = k * SL, len = in_len - o < SL ? in_len - o : SL;
		mscomp_amd_plan* p = nullptr;
		rs = tls.plan_for(MSCOMP_LZNT1, false, len, scap, &p);
		if (rs != MSCOMP_OK) { break; }
		if (hipStreamWaitEvent(tls.exec, tls.ev[2 * k], 0) != hipSuccess) { rs = MSCOMP_ERRNO; break; }
		rs = mscomp_amd_plan_execute(p, d_in + o, d_out + k * scap, d_len + k, d_st + k);
		if (rs != MSCOMP_OK) { break; }
		if (hipMemcpyAsync(tls.h_len + k, d_len + k, 8, hipMemcpyDeviceToHost, tls.exec) != hipSuccess ||
		    hipMemcpyAsync(tls.h_st + k, d_st + k, 4, hipMemcpyDeviceToHost, tls.exec) != hipSuccess ||
		    hipEventRecord(tls.ev[2 * k + 1], tls.exec) != hipSuccess) { rs = MSCOMP_ERRNO; }
	}
	size_t total = 0;
	bool fits = true;
	for (size_t k = 0; k < ns && rs == MSCOMP_OK; ++k) {
		if (hipEventSynchronize(tls.ev[2 * k + 1]) != hipSuccess) { rs = MSCOMP_ERRNO; break; }
		if (tls.h_st[k] != MSCOMP_OK) { rs = (MSCompStatus)tls.h_st[k]; break; }
		const size_t len = (size_t)tls.h_len[k];
		if (fits && total + len <= cap) {
			if (len && hipMemcpyAsync(out + total, d_out + k * scap, len, hipMemcpyDeviceToHost, tls.d2h) != hipSuccess) { rs = MSCOMP_ERRNO; }
		} else { fits = false; }                                    // MSCOMP_BUF_ERROR (lznt1_compress.cpp:251,267): nothing behind the capacity is written
		total += len;
	}
	(void)hipStreamSynchronize(tls.h2d); (void)hipStreamSynchronize(tls.exec);
	if (hipStreamSynchronize(tls.d2h) != hipSuccess && rs == MSCOMP_OK) { rs = MSCOMP_ERRNO; }
	if (reg_in) { g_pins.unpin(in, in_len); }
	if (reg_out) { g_pins.unpin(out, out_span); }
	if (rs != MSCOMP_OK) { return rs; }
	if (!fits) { return MSCOMP_BUF_ERROR; }
	if (cap - total >= 2) { out[total] = 0; out[total + 1] = 0; }   // the uncounted End_of_buffer bytes (lznt1_compress.cpp:270)
	*out_len = total;
	return MSCOMP_OK;
}

// The same call without staging copies (51 MB: 2.28 -> 2.08 ms; MSCOMP_AMD_ONE_ZEROCOPY=0 switches it off): the caller's buffers are page-locked AND mapped, the chunk kernel
// reads its 4 KiB chunks straight from host memory over PCIe (every input byte once, 16 bytes per lane) and the placement kernel writes the
// images straight into the caller's output. One launch for the whole buffer: no slices, no copies in HBM. -100 = could not map (the caller falls
// back to the sliced path).
// (decompress = true works too, but is slower than the staged path for 51 MB: 2.26 instead of 2.01 ms -- the decoder reads its input twice and
// writes its output in 64-byte rows; not used)
static MSCompStatus lznt1_zero_copy(OneShotTls& tls, bool decompress, const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len)
{
	mscomp_amd_ctx* c = tls.ctx;
	const size_t cap = *out_len;
	const size_t most = decompress ? (in_len / 3 + 1) * 4096 : lznt1_max_compressed_size(in_len) + 2, out_span = cap < most ? cap : most;   // (what the format can produce: see one_shot)
	if (!out_span || !c->one_meta.reserve(64)) { return (MSCompStatus)-100; }
	if (!g_pins.pin(in, in_len)) { return (MSCompStatus)-100; }
	if (!g_pins.pin(out, out_span)) { g_pins.unpin(in, in_len); return (MSCompStatus)-100; }
	void* din = nullptr; void* dout = nullptr;
	MSCompStatus rs = MSCOMP_OK;
	struct { uint64_t len; int32_t st; int32_t pad; } meta = { 0, MSCOMP_ERRNO, 0 };
	if (hipHostGetDevicePointer(&din, const_cast<uint8_t*>(in), 0) != hipSuccess || hipHostGetDevicePointer(&dout, out, 0) != hipSuccess) { (void)hipGetLastError(); rs = (MSCompStatus)-100; }
	if (rs == MSCOMP_OK) {
		uint64_t* d_len = static_cast<uint64_t*>(c->one_meta.p); int32_t* d_st = reinterpret_cast<int32_t*>(d_len + 1);
		mscomp_amd_plan* p = nullptr;
		rs = tls.plan_for(MSCOMP_LZNT1, decompress, in_len, out_span, &p);
		if (rs == MSCOMP_OK) { rs = mscomp_amd_plan_execute(p, static_cast<const uint8_t*>(din), static_cast<uint8_t*>(dout), d_len, d_st); }
		if (rs == MSCOMP_OK && (hipMemcpyAsync(&meta, d_len, sizeof meta, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) { rs = MSCOMP_ERRNO; }
	}
	g_pins.unpin(in, in_len); g_pins.unpin(out, out_span);
	if (rs != MSCOMP_OK) { return rs; }
	if (meta.st != MSCOMP_OK) { return (MSCompStatus)meta.st; }
	*out_len = (size_t)meta.len;
	return MSCOMP_OK;
}

static MSCompStatus one_shot(MSCompFormat format, bool decompress, const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len)
{
	if (!out_len || (in_len && !in) || (*out_len && !out)) { return MSCOMP_ARG_ERROR; }
	thread_local OneShotTls tls;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { return MSCOMP_ERRNO; }   // no GPU / no HIP runtime: fail loudly, never fall back
	if (tls.ctx && tls.ctx->device != dev) { tls.drop(); }
	if (!tls.ctx) {
		if (hipStreamCreateWithFlags(&tls.exec, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&tls.h2d, hipStreamNonBlocking) != hipSuccess ||
		    hipStreamCreateWithFlags(&tls.d2h, hipStreamNonBlocking) != hipSuccess) { tls.drop(); return MSCOMP_ERRNO; }
		MSCompStatus s = mscomp_amd_ctx_create(dev, tls.exec, &tls.ctx); if (s != MSCOMP_OK) { tls.drop(); return s; }
	}
	mscomp_amd_ctx* c = tls.ctx;
	DeviceGuard g(c->device);
	const size_t cap = *out_len;
	if (!decompress && format == MSCOMP_LZNT1 && in_len >= ONE_SLICE + ONE_SLICE / 2) {
		// (MSCOMP_AMD_ONE_ZEROCOPY=0 / mscomp_amd_debug_set_one_shot(1): always the sliced path below)
		if (g_one_zero_copy.load(std::memory_order_relaxed)) { const MSCompStatus z = lznt1_zero_copy(tls, false, in, in_len, out, out_len); if ((int)z != -100) { return z; } }
		return lznt1_compress_pipelined(tls, in, in_len, out, out_len);
	}
	// The device copy of the output is sized by what the format can PRODUCE, never by a generous caller capacity (legal in the
	// reference: *out_len = 1 << 40 must not become a hipMalloc of a terabyte). Compression: at most ms_max_compressed_size (+ the
	// two uncounted LZNT1 End_of_buffer bytes), and a unit that fits that bound gets the status and bytes it would get with any larger
	// capacity. LZNT1 decompression: a chunk takes at least 3 bytes and gives at most 4096. Xpress / Xpress+Huffman decompression
	// has no such bound (a 10-byte match token can stand for 4 GiB), so the device capacity starts at 16 x the input and grows only
	// when the decoder says MSCOMP_BUF_ERROR: OK and DATA_ERROR are final at any capacity, because everything the decoders of
	// xpress_decompress.cpp:405-462 / xpress_huff_decompress.cpp:39-162 test before the output runs full is independent of where it ends.
	size_t dev_cap = cap;
	if (!decompress) {
		size_t most = format == MSCOMP_LZNT1 ? lznt1_max_compressed_size(in_len) + 2 : format == MSCOMP_XPRESS ? xpress_max_compressed_size(in_len) : xpress_huff_max_compressed_size(in_len);
		if (most < dev_cap) { dev_cap = most; }
	} else if (format == MSCOMP_LZNT1) {
		const size_t most = (in_len / 3 + 1) * 4096; if (most < dev_cap) { dev_cap = most; }
	} else {
		const size_t first = in_len * 16 + (1u << 20); if (first < dev_cap) { dev_cap = first; }
	}
	if (!c->one_in.reserve(in_len + 64) || !c->one_meta.reserve(64)) { return MSCOMP_MEM_ERROR; }
	uint8_t* d_in = static_cast<uint8_t*>(c->one_in.p);
	uint64_t* d_len = static_cast<uint64_t*>(c->one_meta.p); int32_t* d_st = reinterpret_cast<int32_t*>(d_len + 1);
	// (another thread's large LZNT1 call may have page-locked this very input: keep its registration alive until our copy is through)
	struct Held { std::vector<std::pair<const uint8_t*, size_t>> v; ~Held() { if (!v.empty()) { g_pins.release(v); } } } held{ g_pins.hold(in, in_len) };
	// Xpress / Xpress+Huffman compression of ONE large buffer (round 6): the buffer goes up in ranges of ONE_RANGE_CHUNKS chunks
	// on the upload stream and the match finding of a range -- chain links and Find, the two kernels that can run on part of a unit: a chunk's links
	// and matches depend on the chunks in front of it only -- starts as soon as the range (and the 64 bytes behind it that the last positions compare
	// into) has landed, under the upload of the next one; parse / Huffman / encode (or the Xpress emit kernels) follow over the whole buffer as
	// before: one stream's flag words, and the place of an Xpress+Huffman chunk in the output, couple all chunks. A copy from pageable memory holds
	// the calling thread, not the GPU, so no page-locking is needed for the overlap.
	const bool ranged = !decompress && (format == MSCOMP_XPRESS || format == MSCOMP_XPRESS_HUFF) && in_len >= ONE_RANGED_MIN &&
	                    g_one_zero_copy.load(std::memory_order_relaxed);      // (mscomp_amd_debug_set_one_shot(1) / MSCOMP_AMD_ONE_ZEROCOPY=0: the one-copy path, for A/B runs and tests)
	if (!ranged && in_len && hipMemcpyAsync(d_in, in, in_len, hipMemcpyHostToDevice, c->stream) != hipSuccess) { return MSCOMP_ERRNO; }
	struct { uint64_t len; int32_t st; int32_t pad; } meta;
	for (bool first = true;; first = false) {
		if (!c->one_out.reserve(dev_cap + 64)) { return MSCOMP_MEM_ERROR; }
		mscomp_amd_plan* p = nullptr;
		MSCompStatus s = tls.plan_for(format, decompress, in_len, dev_cap, &p);
		if (s != MSCOMP_OK) { return s; }
		if (ranged && first) {
			const uint32_t nch = p->n_chunks, nr = (nch + ONE_RANGE_CHUNKS - 1u) / ONE_RANGE_CHUNKS;
			if (!tls.slices(nr)) { return MSCOMP_MEM_ERROR; }
			uint16_t* links = static_cast<uint16_t*>(c->links.p); uint16_t* lasthead = static_cast<uint16_t*>(c->lasthead.p);
			uint16_t* mlen3 = static_cast<uint16_t*>(c->mlen3.p); uint16_t* moff = mlen3 + 1;
			const uint32_t max_off = format == MSCOMP_XPRESS ? 0x2000u : 0xFFFFu; const int clip = format == MSCOMP_XPRESS ? 0 : 1;
			if (hipEventRecord(tls.ev[0], c->stream) != hipSuccess || hipStreamWaitEvent(tls.h2d, tls.ev[0], 0) != hipSuccess) { return MSCOMP_ERRNO; }   // (the staging buffer's last reader was on the execution stream)
			for (uint32_t r = 0; r < nr; ++r) {
				const uint32_t c0 = r * ONE_RANGE_CHUNKS, cnt = nch - c0 < ONE_RANGE_CHUNKS ? nch - c0 : ONE_RANGE_CHUNKS;
				const size_t b0 = (size_t)c0 * 65536u + (r ? 64u : 0u);
				size_t b1 = (size_t)(c0 + cnt) * 65536u + 64u; if (b1 > in_len) { b1 = in_len; }
				if (b1 > b0 && hipMemcpyAsync(d_in + b0, in + b0, b1 - b0, hipMemcpyHostToDevice, tls.h2d) != hipSuccess) { return MSCOMP_ERRNO; }
				if (hipEventRecord(tls.ev[1 + r], tls.h2d) != hipSuccess || hipStreamWaitEvent(c->stream, tls.ev[1 + r], 0) != hipSuccess) { return MSCOMP_ERRNO; }
				{ KernelTimer t(c, "xp_links_kernel"); launch_xp_links_range(c->stream, d_in, p->bt, links, lasthead, c0, cnt); }
				{ KernelTimer t(c, "xp_find_kernel"); launch_xp_find_range(c->stream, d_in, p->bt, links, lasthead, mlen3, moff, max_off, clip, c0, cnt); }
			}
			p->matches_ready = true;
		}
		s = mscomp_amd_plan_execute(p, d_in, static_cast<uint8_t*>(c->one_out.p), d_len, d_st);
		p->matches_ready = false;                            // (a cached plan: whatever happened, its next execution finds its matches itself)
		if (s != MSCOMP_OK) { return s; }
		if (hipMemcpyAsync(&meta, d_len, sizeof meta, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
		    hipStreamSynchronize(c->stream) != hipSuccess) { return MSCOMP_ERRNO; }
		if (meta.st == MSCOMP_BUF_ERROR && dev_cap < cap) {          // only the staging was too small: decode again with room
			dev_cap = dev_cap > cap / 16 ? cap : dev_cap * 16;
			continue;
		}
		break;
	}
	if (meta.st != MSCOMP_OK) { return (MSCompStatus)meta.st; }
	size_t copy = (size_t)meta.len;
	if (!decompress && format == MSCOMP_LZNT1 && cap - copy >= 2) { copy += 2; }    // the uncounted End_of_buffer bytes (lznt1_compress.cpp:270)
	if (copy && hipMemcpy(out, c->one_out.p, copy, hipMemcpyDeviceToHost) != hipSuccess) { return MSCOMP_ERRNO; }
	*out_len = (size_t)meta.len;
	return MSCOMP_OK;
}

MSCompStatus lznt1_compress(const uint8_t* in, size_t n, uint8_t* out, size_t* out_len)       { return one_shot(MSCOMP_LZNT1, false, in, n, out, out_len); }
MSCompStatus xpress_compress(const uint8_t* in, size_t n, uint8_t* out, size_t* out_len)      { return one_shot(MSCOMP_XPRESS, false, in, n, out, out_len); }
MSCompStatus xpress_huff_compress(const uint8_t* in, size_t n, uint8_t* out, size_t* out_len) { return one_shot(MSCOMP_XPRESS_HUFF, false, in, n, out, out_len); }

// the decompressors (lznt1_decompress.cpp:293 via internal.h:616-630, ...): same staging, the decoding happens on the GPU
MSCompStatus lznt1_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t* out_len)       { return one_shot(MSCOMP_LZNT1, true, in, n, out, out_len); }
MSCompStatus xpress_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t* out_len)      { return one_shot(MSCOMP_XPRESS, true, in, n, out, out_len); }
MSCompStatus xpress_huff_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t* out_len) { return one_shot(MSCOMP_XPRESS_HUFF, true, in, n, out, out_len); }

#ifndef MSCOMP_AMD_NO_FACADE
MSCompStatus ms_decompress(MSCompFormat format, const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len)
{
	switch ((int)format) {
	case MSCOMP_LZNT1: case MSCOMP_XPRESS: case MSCOMP_XPRESS_HUFF: return one_shot(format, true, in, in_len, out, out_len);
	case MSCOMP_NONE:                                               // mscomp.cpp:26-32
		if (!out_len || in_len > *out_len) { return out_len ? MSCOMP_BUF_ERROR : MSCOMP_ARG_ERROR; }
		if (in_len) { memcpy(out, in, in_len); }
		*out_len = in_len; return MSCOMP_OK;
	default: return MSCOMP_ARG_ERROR;                                // mscomp.cpp:131
	}
}
MSCompStatus ms_compress(MSCompFormat format, const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len)
{
	switch ((int)format) {
	case MSCOMP_LZNT1: case MSCOMP_XPRESS: case MSCOMP_XPRESS_HUFF: return one_shot(format, false, in, in_len, out, out_len);
	case MSCOMP_NONE:                                               // the reference's "copy" codec (mscomp.cpp:26-32): host memcpy
		if (!out_len || in_len > *out_len) { return out_len ? MSCOMP_BUF_ERROR : MSCOMP_ARG_ERROR; }
		if (in_len) { memcpy(out, in, in_len); }
		*out_len = in_len; return MSCOMP_OK;
	default: return MSCOMP_ARG_ERROR;                                // mscomp.cpp:115
	}
}
#endif

} // extern "C"

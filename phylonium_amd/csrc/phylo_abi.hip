// phylo_abi.hip — the C ABI (include/phylonium_amd.h): context, device memory,
// orchestration of the phase-A and phase-B kernels, and the host-side steps
// that stay on the CPU cores (suffix array, sort + chain filter).
//
// Mirrors process() of /root/reference/src/process.cxx:408-556.  There is no
// CPU compute fallback: every entry point that does the path's arithmetic
// launches HIP kernels and fails if no device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/phylonium_amd.h"
#include "../host/fasta_reader.hpp"
#include "hostlogic.hpp"
#include "kernels.h"

using namespace phy;

namespace {

thread_local std::string g_last_error;

double now_ms()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

template <class T> struct DevBuf {
	T *p = nullptr;
	size_t cap = 0; // elements
	hipError_t ensure(size_t n)
	{
		if (n <= cap) return hipSuccess;
		if (p) (void)hipFree(p);
		p = nullptr;
		cap = 0;
		size_t want = n + n / 8 + 16;
		hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
		if (e == hipSuccess) cap = want;
		return e;
	}
	void release()
	{
		if (p) (void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
};

struct TimedSpan {
	std::string name;
	hipEvent_t a, b;
};

// page-locked host staging buffer (grow-only): async copies to/from it do not
// bounce through the runtime's own staging area
template <class T> struct PinBuf {
	T *p = nullptr;
	size_t cap = 0;
	hipError_t ensure(size_t n)
	{
		if (n <= cap) return hipSuccess;
		if (p) (void)hipHostFree(p);
		p = nullptr;
		cap = 0;
		size_t want = n + n / 4 + 64;
		hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
		if (e == hipSuccess) cap = want;
		return e;
	}
	void release()
	{
		if (p) (void)hipHostFree(p);
		p = nullptr;
		cap = 0;
	}
};

// Persistent worker threads for the per-query host steps (std::sort + chain
// filter); replaces the reference's `#pragma omp parallel for` at process.cxx:433.
class WorkerPool
{
	std::vector<std::thread> threads;
	std::mutex m;
	std::condition_variable cv_work, cv_done;
	std::function<void(size_t)> job;
	std::atomic<size_t> next{0};
	size_t total = 0, generation = 0, running = 0;
	bool stop = false;

	void loop()
	{
		size_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> lk(m);
				cv_work.wait(lk, [&] { return stop || generation != seen; });
				if (stop) return;
				seen = generation;
			}
			for (;;) {
				size_t i = next.fetch_add(1);
				if (i >= total) break;
				job(i);
			}
			std::unique_lock<std::mutex> lk(m);
			if (--running == 0) cv_done.notify_all();
		}
	}

  public:
	explicit WorkerPool(size_t n)
	{
		for (size_t t = 0; t < n; t++) threads.emplace_back([this] { loop(); });
	}
	~WorkerPool()
	{
		{
			std::unique_lock<std::mutex> lk(m);
			stop = true;
		}
		cv_work.notify_all();
		for (auto &t : threads) t.join();
	}
	size_t size() const { return threads.size(); }
	// f(i) for i in [0, n) on the pool; `meanwhile`, if given, runs on the calling thread
	// while the pool works (it is the one thread that talks to the GPU).
	// The threads sleep between jobs and are not spun up ahead of one: the GPU boxes this
	// runs on give a process a CPU-time quota (cgroup cpu.max, 16 CPUs' worth here), and
	// 48 spinning threads run into it within a few milliseconds.
	void run(size_t n, std::function<void(size_t)> f, const std::function<void()> &meanwhile = nullptr)
	{
		if (n == 0) return;
		if (threads.empty() || n < 8) { // waking the pool costs more than a handful of lists
			for (size_t i = 0; i < n; i++) f(i);
			if (meanwhile) meanwhile();
			return;
		}
		std::unique_lock<std::mutex> lk(m);
		job = std::move(f);
		total = n;
		next = 0;
		running = threads.size();
		generation++;
		cv_work.notify_all();
		if (meanwhile) {
			lk.unlock();
			meanwhile();
			lk.lock();
		}
		cv_done.wait(lk, [&] { return running == 0; });
	}
};

} // namespace

struct phylo_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t copy_stream = nullptr; // uploads that run beside kernels of `stream` (ordered by events)
	std::vector<hipEvent_t> copy_events;
	// Phase A's per-query tail — bridges, fold, sort + filter, projection — is issued per group of queries,
	// every group on its own stream, so that one group's fold / filter / projection run under the other
	// groups' bridge tails (a bridge kernel ends with a few long dependent chains and an idle device)
	static const int TAIL_GROUPS = 3;
	hipStream_t tail_stream[TAIL_GROUPS - 1] = {nullptr, nullptr};
	hipEvent_t tail_event[TAIL_GROUPS] = {nullptr, nullptr, nullptr};
	// Phase A in groups of queries, one behind the other (option "pipeline_groups"): a group's bridges, fold, filter and
	// projection run on tail_stream[0] while the next group's speculative chains run on the context's stream.
	static const int PIPE_GROUPS = 8;
	int opt_pipeline_groups = 1;           // 0: the library chooses, n: that many (as far as the queries allow)
	hipEvent_t pipe_event[PIPE_GROUPS + 1] = {}; // group g's speculative chains are done; [PIPE_GROUPS]: the last tail is
	std::vector<uint32_t> plan_gb;         // the plan's groups: queries [plan_gb[g], plan_gb[g+1]) ...
	std::vector<uint32_t> plan_item0;      // ... and their items [plan_item0[g], plan_item0[g+1]) of the work order
	int opt_filter_kernel = 0; // option "filter_kernel": 0 stretch-wise chain filter (then the general kernel for what it hands over), 1 general only
	int opt_sa_builder = 1; // option "sa_builder": who builds the suffix array when the caller brings none — 1 the device, 0 the host cores
	uint32_t opt_pairs_wchunk = 0; // option "pairs_wchunk": windows per chunk of the pair kernel (0: chosen from the L2 size)
	uint32_t opt_lean_batch = 0; // option "lean_batch": the chain kernels' rarer phases on every n-th trip only (0, 1: every trip)
	uint32_t opt_fold_blocks = 0; // option "fold_blocks": blocks per query of the fold kernel (0: chosen from the number of queries)
	int opt_pairs_kernel = 0; // option "pairs_kernel": 0 the matrix-core kernel when no projected position holds '!' (default), 1 the vector-ALU kernel always
	int opt_tail_groups = 1; // option "tail_groups": streams the tail is spread over (default 1: measured, the groups run in lockstep and nothing is hidden — DESIGN.md)
	std::string err;
	int n_cu = 256;

	// options
	uint32_t opt_chunk = 0, opt_chunk_tail = 0, opt_kmer = 0;
	int plan_spec_per_cu = 4; // blocks of the speculative chain kernel per CU the plan was made for
	bool profile = false;
	int backend = 0;
	int host_threads = 0;

	// genomes
	size_t n = 0;
	std::vector<uint64_t> goff, glen;
	uint8_t *d_genomes = nullptr;
	bool own_genomes = false;
	DevBuf<uint8_t> genomes_store;
	DevBuf<uint64_t> d_goff;
	DevBuf<uint32_t> d_glen;

	// reference index
	bool have_ref = false;
	size_t ref_idx = 0;
	uint32_t L = 0, ns = 0, k = 0, threshold = 0;
	DevBuf<uint8_t> d_S;
	DevBuf<U4> d_SAX, d_SLOT;
	DevBuf<uint32_t> d_SA;
	DevBuf<uint32_t> d_LCP, d_T;
	// 2-bit packed companions for the lean chain kernels (lean_core.h): genomes and S, 16 bases per
	// dword, and the sorted positions of their non-ACGT bytes
	DevBuf<uint32_t> d_Q2, d_QBAD, d_qbad_off, d_S2, d_SBAD, d_badscr;
	DevBuf<uint64_t> d_badoff;
	uint32_t nsb = 0, sb_first = 0;
	bool cache_quirk = false; // the reference's 6-mer cache over-reports matches on this subject (hostlogic.hpp: esa_cache_quirks)
	DevBuf<uint8_t> d_ABS;    // one byte per k-mer: the step of a window whose k-mer does not occur in S (lean_core.h: LeanIndex::absent)
	int opt_absent_table = 0; // option "absent_table": 1 the chain kernels take such steps from the table, 0 every step fetches its slot (default: measured faster)
	bool abs_built = false;   // d_ABS holds the table of the installed subject
	DevBuf<U4> d_quirk;       // its over-deep entries {prefix, k | depth << 8, lo, hi} for the chains' slow resolver (lean_core.h)
	uint32_t nquirk = 0;
	int opt_cache_quirk = 1; // option "cache_quirk": 1 reproduce what the reference answers on such a subject (default), 0 the true longest matches
	int anchor_kernel = 1; // option "anchor_kernel": 1 lean 2-bit chains (default), 0 the general byte-wise chains
	int lean_force_slow = 0;

	// phase A scratch
	DevBuf<uint64_t> a_qoff;
	DevBuf<uint32_t> a_qlen, a_qchunk0, a_qnb, a_qanc0, a_items, a_chunk_query, a_spec_cnt, a_visited, a_misc;
	DevBuf<Anchor> a_spec_anchors;
	DevBuf<SpecExit> a_spec_exit;
	DevBuf<BridgeRec> a_bridge;
	DevBuf<PoolBlock> a_pool;
	DevBuf<RawHom> a_raw, a_raw_compact;
	DevBuf<uint64_t> a_out_base, a_cmp_base;
	DevBuf<uint32_t> a_out_cap, a_out_cnt;

	// homologies (host, ctx-owned)
	std::vector<std::vector<phylo_homology>> homs;

	// phase B scratch
	DevBuf<uint32_t> b_planes, b_hom_rng, b_tiles, b_flag, b_first;
	// phase A over all genomes leaves the filtered lists on the device already (see phylo_anchor)
	int filter_mode = 0; // option "filter": 0 device sort + filter for 128 queries or more, host below; 1 host; 2 device
	DevBuf<uint32_t> a_flt; // [0] kept total, [1..nq] per-query flags of the device sort + filter
	DevBuf<uint8_t> a_long; // scratch slots of the long-list filter kernel (allocated when a query is long enough to need it)
	bool anchor_pending = false; // a deferred phase A is queued: its flags (h_rng) have not been read yet
	double pend_t0 = 0, pend_t1 = 0, pend_t2 = 0, pend_total = 0;
	uint32_t pend_nch = 0, pend_C = 0;
	bool homs_staged = false;
	// ... and has projected them for the whole reference (part 0 of 1); with five planes or three
	bool eager_valid = false, eager_five = false;
	PinBuf<uint32_t> h_rng;
	// lists attached by phylo_attach_packed_device: borrowed device records + per-genome ranges
	const DevHom *att_homs = nullptr;
	std::vector<uint64_t> att_begin, att_count;
	std::vector<uint8_t> host_stale; // [n] 1: the host list of this genome must be fetched from att_homs first
	bool att_rng_on_device = false;  // att_begin / att_count have not been read back yet: they are b_hom_rng (phylo_attach_blocks_device)
	bool att_unchecked = false;      // ... and their validity flags (b_flag[1..2]) have not been looked at yet
	hipStream_t own_stream = nullptr; // the stream this context created (phylo_ctx_set_stream may lend it another)
	bool pileup_five = false; // the last projection met '!': start with five planes next time
	DevBuf<DevHom> b_homs;
	DevBuf<unsigned long long> b_subst, b_homologs;
	DevBuf<uint32_t> b_sym32; // both result matrices as symmetric u32, on their way to the host
	DevBuf<uint32_t> b_bang;  // the projected '!' of the three-plane projection: {genome | reverse << 31, position} each
	uint32_t bang_cap = 0;    // as many as the genomes hold separators (a separator is projected at most once)
	DevBuf<Segment> s_segs;
	DevBuf<uint64_t> s_out;

	// host staging and workers
	PinBuf<uint32_t> h_cnt;
	PinBuf<RawHom> h_raw;
	PinBuf<DevHom> h_devhom;
	PinBuf<uint64_t> h_mat;
	std::unique_ptr<WorkerPool> pool;
	// cached phase-A plan
	bool plan_valid = false;
	size_t plan_qb = 0, plan_qe = 0;
	ChunkPlan plan;
	std::vector<uint64_t> plan_out_base;
	uint64_t plan_raw_total = 0;

	// stats
	std::map<std::string, double> stats;
	std::vector<TimedSpan> spans;
	std::vector<hipEvent_t> event_pool;

	int fail(const char *fmt, ...)
	{
		char buf[1024];
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(buf, sizeof buf, fmt, ap);
		va_end(ap);
		err = buf;
		g_last_error = buf;
		return 1;
	}
};

#define HIPOK(ctx, call)                                                                                         \
	do {                                                                                                         \
		hipError_t e__ = (call);                                                                                 \
		if (e__ != hipSuccess) return (ctx)->fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
	} while (0)

namespace {

hipEvent_t get_event(phylo_ctx *c)
{
	if (!c->event_pool.empty()) {
		hipEvent_t e = c->event_pool.back();
		c->event_pool.pop_back();
		return e;
	}
	hipEvent_t e;
	(void)hipEventCreate(&e);
	return e;
}

// Times one kernel launch with HIP events on the context's stream.
struct KernelSpan {
	phylo_ctx *c;
	hipEvent_t a = nullptr, b = nullptr;
	const char *name;
	hipStream_t st;
	KernelSpan(phylo_ctx *ctx, const char *nm, hipStream_t on = nullptr) : c(ctx), name(nm), st(on ? on : ctx->stream)
	{
		if (c->profile) {
			a = get_event(c);
			b = get_event(c);
			(void)hipEventRecord(a, st);
		}
	}
	~KernelSpan()
	{
		if (c->profile) {
			(void)hipEventRecord(b, st);
			c->spans.push_back(TimedSpan{name, a, b});
		}
	}
};

// Call after the stream has been synchronised.
void harvest_spans(phylo_ctx *c)
{
	for (TimedSpan &s : c->spans) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
			c->stats["ms:" + s.name] += ms;
			c->stats["n:" + s.name] += 1;
		}
		c->event_pool.push_back(s.a);
		c->event_pool.push_back(s.b);
	}
	c->spans.clear();
}

int sync_stream(phylo_ctx *c)
{
	HIPOK(c, hipStreamSynchronize(c->stream));
	harvest_spans(c);
	return 0;
}

static QuerySrc query_src(const phylo_ctx *c) { return QuerySrc{c->d_Q2.p, c->d_goff.p, c->d_QBAD.p, c->d_qbad_off.p}; }

WorkerPool &workers(phylo_ctx *c)
{
	if (!c->pool) {
		size_t n;
		if (c->host_threads > 0) {
			n = (size_t)c->host_threads;
		} else {
			unsigned h = std::thread::hardware_concurrency();
			n = h ? std::min(h, 48u) : 1;
		}
		c->pool.reset(new WorkerPool(n <= 1 ? 0 : n));
	}
	return *c->pool;
}

} // namespace

extern "C" {

const char *phylo_version(void) { return "phylonium_amd 0.1 (gfx950)"; }

const char *phylo_last_error(const phylo_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int phylo_ctx_create(phylo_ctx **out, int device)
{
	if (!out) return 1;
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		g_last_error = std::string("no usable HIP device: ") + hipGetErrorString(e);
		return 2;
	}
	if (device < 0 || device >= count) {
		g_last_error = "device ordinal out of range";
		return 3;
	}
	phylo_ctx *c = new phylo_ctx();
	c->device = device;
	if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&c->stream)) != hipSuccess ||
		(e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) != hipSuccess) {
		g_last_error = std::string("cannot initialise device: ") + hipGetErrorString(e);
		delete c;
		return 4;
	}
	c->own_stream = c->stream;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
	*out = c;
	return 0;
}

int phylo_ctx_set_stream(phylo_ctx *c, void *stream)
{
	if (!c) return 1;
	HIPOK(c, hipSetDevice(c->device));
	HIPOK(c, hipStreamSynchronize(c->stream));
	harvest_spans(c);
	c->stream = stream ? (hipStream_t)stream : c->own_stream;
	return 0;
}

int phylo_ctx_device(const phylo_ctx *c) { return c ? c->device : -1; }

void phylo_ctx_destroy(phylo_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	c->pool.reset();
	c->h_cnt.release();
	c->h_rng.release();
	c->h_raw.release();
	c->h_devhom.release();
	c->h_mat.release();
	c->genomes_store.release();
	c->d_goff.release();
	c->d_glen.release();
	c->d_S.release();
	c->d_SAX.release();
	c->d_SLOT.release();
	c->d_SA.release();
	c->d_LCP.release();
	c->d_T.release();
	c->d_Q2.release();
	c->d_QBAD.release();
	c->d_qbad_off.release();
	c->d_S2.release();
	c->d_SBAD.release();
	c->d_badscr.release();
	c->d_badoff.release();
	c->d_quirk.release();
	c->d_ABS.release();
	c->a_flt.release();
	c->a_long.release();
	c->a_qoff.release();
	c->a_qlen.release();
	c->a_qchunk0.release();
	c->a_qnb.release();
	c->a_qanc0.release();
	c->a_items.release();
	c->a_chunk_query.release();
	c->a_spec_cnt.release();
	c->a_visited.release();
	c->a_misc.release();
	c->a_spec_anchors.release();
	c->a_spec_exit.release();
	c->a_bridge.release();
	c->a_pool.release();
	c->a_raw.release();
	c->a_raw_compact.release();
	c->a_out_base.release();
	c->a_cmp_base.release();
	c->a_out_cap.release();
	c->a_out_cnt.release();
	c->b_planes.release();
	c->b_hom_rng.release();
	c->b_tiles.release();
	c->b_flag.release();
	c->b_first.release();
	c->b_homs.release();
	c->b_subst.release();
	c->b_homologs.release();
	c->b_sym32.release();
	c->b_bang.release();
	c->s_segs.release();
	c->s_out.release();
	for (TimedSpan &s : c->spans) {
		(void)hipEventDestroy(s.a);
		(void)hipEventDestroy(s.b);
	}
	for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
	for (hipEvent_t e : c->copy_events) (void)hipEventDestroy(e);
	for (hipStream_t ts : c->tail_stream)
		if (ts) (void)hipStreamDestroy(ts);
	for (hipEvent_t e : c->pipe_event)
		if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : c->tail_event)
		if (e) (void)hipEventDestroy(e);
	if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
	(void)hipStreamDestroy(c->own_stream);
	delete c;
}

int phylo_set_option(phylo_ctx *c, const char *key, long value)
{
	if (!c || !key) return 1;
	std::string k = key;
	if (k == "chunk") {
		if (value != 0 && (value < 64 || value % 64 || value > 65536)) return c->fail("chunk must be 0 or a multiple of 64 in 64..65536");
		c->opt_chunk = (uint32_t)value;
		c->plan_valid = false;
		c->homs_staged = false;
	c->homs_staged = false;
	} else if (k == "chunk_tail") {
		if (value != 0 && (value < 64 || value % 64 || value > 65536)) return c->fail("chunk_tail must be 0 or a multiple of 64 in 64..65536");
		c->opt_chunk_tail = (uint32_t)value;
		c->plan_valid = false;
	} else if (k == "kmer") {
		if (value < 0 || value > 14) return c->fail("kmer must be in 0..14");
		c->opt_kmer = (uint32_t)value;
		c->have_ref = false;
	} else if (k == "anchor_kernel") {
		if (value != 0 && value != 1) return c->fail("anchor_kernel must be 1 (lean 2-bit chains) or 0 (general byte-wise chains)");
		c->anchor_kernel = (int)value;
		c->plan_valid = false;
	} else if (k == "filter_kernel") {
		if (value != 0 && value != 1) return c->fail("filter_kernel must be 0 (stretch-wise) or 1 (general)");
		c->opt_filter_kernel = (int)value;
	} else if (k == "pipeline_groups") {
		if (value < 0 || value > phylo_ctx::PIPE_GROUPS) return c->fail("pipeline_groups must be in 0..%d", phylo_ctx::PIPE_GROUPS);
		c->opt_pipeline_groups = (int)value;
		c->plan_valid = false;
	} else if (k == "tail_groups") {
		if (value < 1 || value > phylo_ctx::TAIL_GROUPS) return c->fail("tail_groups must be in 1..%d", phylo_ctx::TAIL_GROUPS);
		c->opt_tail_groups = (int)value;
	} else if (k == "sa_builder") {
		if (value != 0 && value != 1) return c->fail("sa_builder must be 1 (device) or 0 (host cores)");
		c->opt_sa_builder = (int)value;
	} else if (k == "pairs_wchunk") {
		if (value < 0 || value > (1 << 20)) return c->fail("pairs_wchunk must be in 0..2^20");
		c->opt_pairs_wchunk = (uint32_t)value;
	} else if (k == "cache_quirk") {
		if (value != 0 && value != 1) return c->fail("cache_quirk must be 1 (as the reference answers) or 0 (true longest matches)");
		c->opt_cache_quirk = (int)value;
		c->plan_valid = false;
		c->homs_staged = false;
	} else if (k == "absent_table") {
		if (value != 0 && value != 1) return c->fail("absent_table must be 1 (steps of absent k-mers from the table) or 0 (every step fetches its slot)");
		c->opt_absent_table = (int)value;
	} else if (k == "lean_batch") {
		if (value < 0 || value > 16) return c->fail("lean_batch must be in 0..16");
		c->opt_lean_batch = (uint32_t)value;
	} else if (k == "fold_blocks") {
		if (value < 0 || value > 64) return c->fail("fold_blocks must be in 0..64");
		c->opt_fold_blocks = (uint32_t)value;
	} else if (k == "pairs_kernel") {
		if (value != 0 && value != 1) return c->fail("pairs_kernel must be 0 (matrix cores unless '!' is projected) or 1 (vector ALUs)");
		c->opt_pairs_kernel = (int)value;
	} else if (k == "lean_force_slow") {
		c->lean_force_slow = value != 0;
	} else if (k == "profile") {
		c->profile = value != 0;
	} else if (k == "filter") {
		if (value < 0 || value > 2) return c->fail("filter must be 0 (auto), 1 (host) or 2 (device)");
		c->filter_mode = (int)value;
	} else if (k == "compare_backend") {
		if (value != 0 && value != 1) return c->fail("compare_backend must be 0 (pileup) or 1 (segments)");
		c->backend = (int)value;
	} else if (k == "host_threads") {
		c->host_threads = (int)value;
		c->pool.reset();
	} else {
		return c->fail("unknown option '%s'", key);
	}
	return 0;
}

int phylo_get_stat(phylo_ctx *c, const char *key, double *out)
{
	if (!c || !key || !out) return 1;
	auto it = c->stats.find(key);
	if (it == c->stats.end()) return 1;
	*out = it->second;
	return 0;
}

int phylo_reset_stats(phylo_ctx *c)
{
	if (!c) return 1;
	c->stats.clear();
	return 0;
}

size_t phylo_stat_keys(phylo_ctx *c, char *buf, size_t cap)
{
	if (!c) return 0;
	size_t need = 1;
	for (auto &kv : c->stats) need += kv.first.size() + 1;
	if (buf && cap >= need) {
		char *w = buf;
		for (auto &kv : c->stats) {
			memcpy(w, kv.first.c_str(), kv.first.size() + 1);
			w += kv.first.size() + 1;
		}
		*w = '\0';
	}
	return need;
}

// ───────────────────────── genomes ─────────────────────────

// Sorted non-ACGT positions of `nseq` sequences (base + off[j], len[j] bytes; device arrays) into
// `out` (grown as needed), list j at [list_off[j], list_off[j+1]); `extra` more slots are left
// after the last list.  Two passes of bad_positions_kernel over segments of 1 MiB.
static int bad_lists(phylo_ctx *c, const uint8_t *base, const uint64_t *d_off, const uint32_t *d_len,
					 const std::vector<uint64_t> &len, DevBuf<uint32_t> &out, std::vector<uint32_t> &list_off, size_t extra)
{
	const size_t nseq = len.size();
	const uint64_t SEG = bad_segment_bytes();
	std::vector<uint32_t> seg_seq, seg_first(nseq + 1);
	for (size_t j = 0; j < nseq; j++) {
		seg_first[j] = (uint32_t)seg_seq.size();
		const uint64_t ns = std::max<uint64_t>(1, (len[j] + SEG - 1) / SEG);
		for (uint64_t t = 0; t < ns; t++) seg_seq.push_back((uint32_t)j);
	}
	seg_first[nseq] = (uint32_t)seg_seq.size();
	const size_t nseg = seg_seq.size();
	list_off.assign(nseq + 1, 0);
	if (!nseg) return c->d_badscr.ensure(4) == hipSuccess && out.ensure(extra + 1) == hipSuccess ? 0 : c->fail("out of device memory");
	// scratch: seg_seq | seg_first | seg_cnt | seg_off
	HIPOK(c, c->d_badscr.ensure(3 * nseg + nseq + 1));
	uint32_t *d_seq = c->d_badscr.p, *d_first = d_seq + nseg, *d_cnt = d_first + nseq + 1, *d_soff = d_cnt + nseg;
	hipStream_t st = c->stream;
	HIPOK(c, hipMemcpyAsync(d_seq, seg_seq.data(), nseg * 4, hipMemcpyHostToDevice, st));
	HIPOK(c, hipMemcpyAsync(d_first, seg_first.data(), (nseq + 1) * 4, hipMemcpyHostToDevice, st));
	launch_bad_positions(base, d_off, d_len, d_seq, d_first, (uint32_t)nseg, d_cnt, nullptr, nullptr, st);
	std::vector<uint32_t> cnt(nseg), soff(nseg + 1, 0);
	HIPOK(c, hipMemcpyAsync(cnt.data(), d_cnt, nseg * 4, hipMemcpyDeviceToHost, st));
	HIPOK(c, hipStreamSynchronize(st));
	uint64_t tot = 0;
	for (size_t g = 0; g < nseg; g++) {
		soff[g] = (uint32_t)tot;
		tot += cnt[g];
	}
	if (tot + extra >= 0xffffffffull) return c->fail("more than 2^32 non-ACGT positions");
	soff[nseg] = (uint32_t)tot;
	for (size_t j = 0; j <= nseq; j++) list_off[j] = soff[seg_first[j]];
	HIPOK(c, out.ensure(tot + extra + 1));
	if (tot) {
		HIPOK(c, hipMemcpyAsync(d_soff, soff.data(), nseg * 4, hipMemcpyHostToDevice, st));
		launch_bad_positions(base, d_off, d_len, d_seq, d_first, (uint32_t)nseg, d_cnt, d_soff, out.p, st);
	}
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipStreamSynchronize(st));
	return 0;
}

// Q2 + QBAD of the installed genomes (d_goff / d_glen are in place)
static int pack_genomes(phylo_ctx *c)
{
	const size_t n = c->n;
	double t0 = now_ms();
	uint64_t extent = 64;
	for (size_t j = 0; j < n; j++) extent = std::max<uint64_t>(extent, c->goff[j] + (c->glen[j] + 63) / 64 * 64 + 64);
	if (extent / 16 + 64 >= 0xffffffffull) return c->fail("genome buffer too large for 32-bit word offsets");
	const size_t words = (size_t)(extent / 16);
	HIPOK(c, c->d_Q2.ensure(words + 64));
	HIPOK(c, hipMemsetAsync(c->d_Q2.p, 0, (words + 64) * 4, c->stream));
	if (n) launch_pack2(c->d_genomes, (uint64_t)words * 16, c->d_Q2.p, c->stream);
	std::vector<uint32_t> off;
	if (bad_lists(c, c->d_genomes, c->d_goff.p, c->d_glen.p, c->glen, c->d_QBAD, off, 1)) return 1;
	HIPOK(c, c->d_qbad_off.ensure(n + 2));
	HIPOK(c, hipMemcpy(c->d_qbad_off.p, off.data(), (n + 1) * 4, hipMemcpyHostToDevice));
	c->stats["ms:pack_genomes"] += now_ms() - t0;
	c->stats["count:genome_non_acgt"] = off[n];
	c->pileup_five = off[n] > 0;
	c->bang_cap = off[n];
	return 0;
}

static int install_layout(phylo_ctx *c, bool pack = true)
{
	size_t n = c->n;
	std::vector<uint32_t> l32(n);
	for (size_t j = 0; j < n; j++) {
		if (c->glen[j] >= 0xfff00000ull) return c->fail("genome %zu is too long (%llu >= 2^32-2^20)", j, (unsigned long long)c->glen[j]);
		l32[j] = (uint32_t)c->glen[j];
	}
	HIPOK(c, c->d_goff.ensure(n + 1));
	HIPOK(c, c->d_glen.ensure(n + 1));
	HIPOK(c, hipMemcpyAsync(c->d_goff.p, c->goff.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipMemcpyAsync(c->d_glen.p, l32.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	c->homs.assign(n, {});
	c->have_ref = false;
	c->plan_valid = false;
	c->homs_staged = false;
	c->att_homs = nullptr; // lists that only lived in an attached buffer are gone
	c->att_rng_on_device = false;
	c->host_stale.clear();
	// Phase B goes ahead on a guess of whether a projected position will hold '!' (compare_pileup).  Genomes without any
	// separator cannot project one; genomes in several contigs nearly always do (a homology that ends at a contig join
	// carries it).  The guess is re-seeded from the new genomes' '!' count where that count becomes known, so the first
	// call after an install starts with the right kernels instead of repeating the work.
	c->pileup_five = false;
	return pack ? pack_genomes(c) : 0;
}

// goff / glen / arena size for n genomes of the given lengths: 64 bytes in front, every genome padded to a
// multiple of 64 and followed by 64 zero bytes, 256 behind the last (kernels prefetch whole 128-byte windows)
static uint64_t layout_genomes(phylo_ctx *c, size_t n, const size_t *len)
{
	c->n = n;
	c->goff.assign(n, 0);
	c->glen.assign(n, 0);
	uint64_t tot = 64;
	for (size_t j = 0; j < n; j++) {
		c->goff[j] = tot;
		c->glen[j] = len[j];
		tot += ((len[j] + 63) / 64) * 64 + 64;
	}
	return tot + 256;
}

int phylo_set_genomes(phylo_ctx *c, size_t n, const char *const *seq, const size_t *len)
{
	if (!c) return 1;
	if (n && (!seq || !len)) return c->fail("null genome arrays");
	HIPOK(c, hipSetDevice(c->device));
	const uint64_t tot = layout_genomes(c, n, len);
	double t0 = now_ms();
	HIPOK(c, c->genomes_store.ensure(tot));
	HIPOK(c, hipMemsetAsync(c->genomes_store.p, 0, tot, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	double t1 = now_ms();
	// one thread, one stream: copies from pageable memory issued from several threads at once run at a
	// fraction of this rate (measured: 38 GB/s against 6-20 GB/s from 4-16 threads)
	for (size_t j = 0; j < n; j++)
		if (len[j])
			HIPOK(c, hipMemcpyAsync(c->genomes_store.p + c->goff[j], seq[j], len[j], hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	c->d_genomes = c->genomes_store.p;
	c->own_genomes = true;
	double t2 = now_ms();
	int rc = install_layout(c);
	c->stats["ms:genomes_alloc"] += t1 - t0;
	c->stats["ms:genomes_copy"] += t2 - t1;
	c->stats["ms:genomes_install"] += now_ms() - t2;
	return rc;
}

// Genomes that arrive as 2-bit codes + separator positions (what phylo_host_read_fasta_packed makes): a quarter
// of the bytes cross PCIe, Q2 is copied straight into place and the byte arena is written by the device.
int phylo_set_genomes_packed(phylo_ctx *c, size_t n, const uint32_t *const *q2, const size_t *len, const uint32_t *const *bad,
							 const size_t *nbad)
{
	if (!c) return 1;
	if (n && (!q2 || !len || !bad || !nbad)) return c->fail("null genome arrays");
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	std::vector<uint32_t> boff(n + 1, 0), blist;
	{
		uint64_t tb = 0;
		for (size_t j = 0; j < n; j++) {
			if (len[j] && !q2[j]) return c->fail("genome %zu: null code array", j);
			if (nbad[j] && !bad[j]) return c->fail("genome %zu: null position list", j);
			for (size_t k = 0; k < nbad[j]; k++)
				if (bad[j][k] >= len[j] || (k && bad[j][k] <= bad[j][k - 1]))
					return c->fail("genome %zu: separator positions must ascend and lie inside the genome", j);
			tb += nbad[j];
			if (tb + 1 >= 0xffffffffull) return c->fail("more than 2^32 non-ACGT positions");
			boff[j + 1] = (uint32_t)tb;
		}
		blist.reserve(tb);
		for (size_t j = 0; j < n; j++) blist.insert(blist.end(), bad[j], bad[j] + nbad[j]);
	}
	const uint64_t tot = layout_genomes(c, n, len);
	if (tot / 16 + 64 >= 0xffffffffull) return c->fail("genome buffer too large for 32-bit word offsets");
	for (size_t j = 0; j < n; j++)
		if (len[j] >= 0xfff00000ull) return c->fail("genome %zu is too long (%llu >= 2^32-2^20)", j, (unsigned long long)len[j]);
	const size_t words = (size_t)(tot / 16);
	HIPOK(c, c->genomes_store.ensure(tot));
	HIPOK(c, c->d_Q2.ensure(words + 64));
	HIPOK(c, hipMemsetAsync(c->d_Q2.p, 0, (words + 64) * 4, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	double t1 = now_ms();
	for (size_t j = 0; j < n; j++)
		if (len[j])
			HIPOK(c, hipMemcpyAsync(c->d_Q2.p + c->goff[j] / 16, q2[j], (len[j] + 15) / 16 * 4, hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	double t2 = now_ms();
	c->d_genomes = c->genomes_store.p;
	c->own_genomes = true;
	if (install_layout(c, false)) return 1; // d_goff / d_glen in place
	HIPOK(c, c->d_QBAD.ensure(blist.size() + 2));
	HIPOK(c, c->d_qbad_off.ensure(n + 2));
	if (!blist.empty()) HIPOK(c, hipMemcpyAsync(c->d_QBAD.p, blist.data(), blist.size() * 4, hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipMemcpyAsync(c->d_qbad_off.p, boff.data(), (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
	launch_unpack2(c->d_Q2.p, c->d_goff.p, c->d_glen.p, (uint32_t)n, tot, c->genomes_store.p, c->d_QBAD.p, c->d_qbad_off.p,
				   (uint32_t)blist.size(), c->stream);
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipStreamSynchronize(c->stream));
	c->stats["ms:genomes_alloc"] += t1 - t0;
	c->stats["ms:genomes_copy"] += t2 - t1;
	c->stats["ms:genomes_install"] += now_ms() - t2;
	c->stats["count:genome_non_acgt"] = (double)blist.size();
	c->pileup_five = !blist.empty();
	c->bang_cap = (uint32_t)blist.size();
	return 0;
}

// The packed genomes already in device memory, laid out as the arena's Q2: word w of dev_q2 holds the codes of
// arena bytes [16w, 16w + 16), genome j at byte offset offsets[j] (the rules of phylo_set_genomes_device), codes
// outside the genomes and at the separator positions 0.  A rank of a multi-GPU run gathers exactly this.
int phylo_set_genomes_packed_device(phylo_ctx *c, size_t n, const void *dev_q2, const uint64_t *offsets, const uint64_t *lens,
									const uint32_t *const *bad, const size_t *nbad)
{
	if (!c) return 1;
	if (n && (!dev_q2 || !offsets || !lens || !bad || !nbad)) return c->fail("null genome arrays");
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	std::vector<uint32_t> boff(n + 1, 0), blist;
	uint64_t tot = 64;
	for (size_t j = 0; j < n; j++) {
		if (offsets[j] % 64 || offsets[j] < 64)
			return c->fail("genome %zu: device offset must be a multiple of 64 and >= 64", j);
		if (j && offsets[j] < offsets[j - 1] + (lens[j - 1] + 63) / 64 * 64 + 64)
			return c->fail("genome %zu: device offsets must ascend, each genome followed by at least 64 bytes of padding", j);
		if (lens[j] >= 0xfff00000ull) return c->fail("genome %zu is too long (%llu >= 2^32-2^20)", j, (unsigned long long)lens[j]);
		if (nbad[j] && !bad[j]) return c->fail("genome %zu: null position list", j);
		for (size_t k = 0; k < nbad[j]; k++)
			if (bad[j][k] >= lens[j] || (k && bad[j][k] <= bad[j][k - 1]))
				return c->fail("genome %zu: separator positions must ascend and lie inside the genome", j);
		if (blist.size() + nbad[j] + 1 >= 0xffffffffull) return c->fail("more than 2^32 non-ACGT positions");
		blist.insert(blist.end(), bad[j], bad[j] + nbad[j]);
		boff[j + 1] = (uint32_t)blist.size();
		tot = offsets[j] + (lens[j] + 63) / 64 * 64 + 64;
	}
	// the caller's buffer reaches 16 bytes' worth of words past the last genome's padding (the header's promise): that
	// much is copied, the rest of this context's Q2 — the 256 bytes the kernels may prefetch behind it — is cleared here
	const size_t src_words = (size_t)(tot / 16) + 1;
	tot += 256;
	if (tot / 16 + 64 >= 0xffffffffull) return c->fail("genome buffer too large for 32-bit word offsets");
	const size_t words = (size_t)(tot / 16);
	c->n = n;
	c->goff.assign(offsets, offsets + n);
	c->glen.assign(lens, lens + n);
	HIPOK(c, c->genomes_store.ensure(tot));
	HIPOK(c, c->d_Q2.ensure(words + 64));
	HIPOK(c, hipMemsetAsync(c->d_Q2.p + src_words, 0, (words + 64 - src_words) * 4, c->stream));
	HIPOK(c, hipMemcpyAsync(c->d_Q2.p, dev_q2, src_words * 4, hipMemcpyDeviceToDevice, c->stream));
	c->d_genomes = c->genomes_store.p;
	c->own_genomes = true;
	if (install_layout(c, false)) return 1;
	HIPOK(c, c->d_QBAD.ensure(blist.size() + 2));
	HIPOK(c, c->d_qbad_off.ensure(n + 2));
	if (!blist.empty()) HIPOK(c, hipMemcpyAsync(c->d_QBAD.p, blist.data(), blist.size() * 4, hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipMemcpyAsync(c->d_qbad_off.p, boff.data(), (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
	launch_unpack2(c->d_Q2.p, c->d_goff.p, c->d_glen.p, (uint32_t)n, tot, c->genomes_store.p, c->d_QBAD.p, c->d_qbad_off.p,
				   (uint32_t)blist.size(), c->stream);
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipStreamSynchronize(c->stream));
	c->stats["ms:genomes_install"] += now_ms() - t0;
	c->stats["count:genome_non_acgt"] = (double)blist.size();
	c->pileup_five = !blist.empty();
	c->bang_cap = (uint32_t)blist.size();
	return 0;
}

int phylo_get_genome(phylo_ctx *c, size_t i, char *buf)
{
	if (!c) return 1;
	if (i >= c->n) return c->fail("genome index %zu out of range (n=%zu)", i, c->n);
	if (!buf && c->glen[i]) return c->fail("null buffer");
	HIPOK(c, hipSetDevice(c->device));
	if (c->glen[i]) HIPOK(c, hipMemcpy(buf, c->d_genomes + c->goff[i], c->glen[i], hipMemcpyDeviceToHost));
	return 0;
}

int phylo_set_genomes_device(phylo_ctx *c, size_t n, const void *dev_base, const uint64_t *offsets,
							 const uint64_t *lens)
{
	if (!c) return 1;
	if (n && (!dev_base || !offsets || !lens)) return c->fail("null genome arrays");
	HIPOK(c, hipSetDevice(c->device));
	for (size_t j = 0; j < n; j++) {
		if (offsets[j] % 64 || offsets[j] < 64)
			return c->fail("genome %zu: device offset must be a multiple of 64 and >= 64 (kernels read up to 32 bytes before a genome)", j);
		// ascending and apart: phase A clears and addresses its visited bitmap by buffer offset, genome after genome
		if (j && offsets[j] < offsets[j - 1] + (lens[j - 1] + 63) / 64 * 64 + 64)
			return c->fail("genome %zu: device offsets must ascend, each genome followed by at least 64 bytes of zero padding "
						   "(rounded up to a multiple of 64) before the next one starts", j);
	}
	c->n = n;
	c->goff.assign(offsets, offsets + n);
	c->glen.assign(lens, lens + n);
	c->d_genomes = (uint8_t *)dev_base;
	c->own_genomes = false;
	return install_layout(c);
}

// ───────────────────────── reference index ─────────────────────────

// One 64-byte slot per k-mer (anchor_core.h: slot_pack): {T[c], T[c+1]} and the SAX
// records of ranks base..base+3, base = T[c] ? T[c]-1 : 0.  One thread per slot.
__global__ __launch_bounds__(256) void build_slots_kernel(const uint32_t *__restrict__ T, const U4 *__restrict__ sax,
														   uint32_t n, uint64_t codes, U4 *__restrict__ slot)
{
	const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= codes) return;
	const uint32_t lo = T[c], hi = T[c + 1];
	const uint32_t base = lo ? lo - 1 : 0;
	U4 rec[4];
	for (uint32_t i = 0; i < 4; i++) rec[i] = base + i < n ? sax[base + i] : U4{0, 0, 0, 0};
	U4 out[4];
	slot_pack(lo, hi, rec, out);
	for (uint32_t i = 0; i < 4; i++) slot[c * SLOT_RECS + i] = out[i];
}

int phylo_set_reference(phylo_ctx *c, size_t ref_idx, const int64_t *sa, size_t threshold)
{
	if (!c) return 1;
	if (ref_idx >= c->n) return c->fail("reference index %zu out of range (n=%zu)", ref_idx, c->n);
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	uint64_t L = c->glen[ref_idx];
	if (2 * L + 1 >= 0x7fffffffull) return c->fail("reference too long: 2L+1 must be < 2^31 (src/esa.cxx:374-375)");
	if (L == 0) return c->fail("reference genome is empty");
	uint32_t ns = (uint32_t)(2 * L + 1);
	// S = subject + '#' + reverse complement is made where the subject is — on the device — together with the GC
	// count the threshold needs; the host gets a copy only for the steps that walk it there (its own suffix sorters,
	// Kasai's LCP for repeats beyond the clip, the exact 6-mer-cache check)
	std::vector<uint8_t> S;
	std::vector<uint32_t> SA; // +4: tables are read 16 bytes at a time
	hipStream_t st = c->stream;
	HIPOK(c, c->d_S.ensure((size_t)ns + 64));
	HIPOK(c, c->d_SA.ensure((size_t)ns + 4));
	HIPOK(c, c->a_misc.ensure(16));
	DevBuf<uint32_t> d_next;
	HIPOK(c, d_next.ensure(next_bytes_entries() + 4));
	HIPOK(c, hipMemsetAsync(c->a_misc.p, 0, 64, st));
	HIPOK(c, hipMemsetAsync(d_next.p, 0, (next_bytes_entries() + 4) * 4, st));
	launch_build_subject(c->d_genomes + c->goff[ref_idx], (uint32_t)L, c->d_S.p, (unsigned long long *)(c->a_misc.p + 2), st);
	launch_next_bytes(c->d_S.p, ns, d_next.p, st);
	unsigned long long gc_count = 0;
	std::vector<uint32_t> next_masks(next_bytes_entries());
	HIPOK(c, hipMemcpyAsync(&gc_count, c->a_misc.p + 2, 8, hipMemcpyDeviceToHost, st));
	HIPOK(c, hipMemcpyAsync(next_masks.data(), d_next.p, next_masks.size() * 4, hipMemcpyDeviceToHost, st));
	HIPOK(c, hipStreamSynchronize(st));
	d_next.release();
	auto host_S = [&]() -> int {
		if (!S.empty()) return 0;
		S.assign((size_t)ns + 64, 0);
		HIPOK(c, hipMemcpy(S.data(), c->d_S.p, (size_t)ns, hipMemcpyDeviceToHost));
		return 0;
	};
	double t1 = now_ms();
	bool sa_on_device = false; // the array exists in d_SA only; `SA` is fetched if a host step asks for it
	uint32_t sa_rounds = 0;
	if (!sa && c->opt_sa_builder == 1) {
		// esa.cxx:74's divsufsort64, on the device (sa_kernels.hip)
		DevBuf<uint8_t> scratch;
		if (scratch.ensure(suffix_array_scratch_bytes(ns)) == hipSuccess) { // ~41 bytes per suffix
			HIPOK(c, hipMemsetAsync(c->d_SA.p + ns, 0, 16, st));
			const int rc = device_suffix_array(c->d_S.p, ns, c->d_SA.p, scratch.p, &sa_rounds, st);
			HIPOK(c, hipStreamSynchronize(st));
			scratch.release();
			if (rc == 2) { // a library primitive refused (e.g. its scratch requirement): the host builders take over
				c->stats["ref:sa_device_error"] = (double)hipGetLastError();
				(void)hipStreamSynchronize(st);
			}
			sa_on_device = rc == 0; // rc == 1: a byte outside ! # A C G T — the host builders order any bytes
		} else {
			(void)hipGetLastError(); // no room for the working set next to the genomes: the host cores sort
		}
	}
	auto host_sa = [&]() -> int { // the array on the host, for the steps that walk it there
		if (!SA.empty()) return 0;
		SA.assign((size_t)ns + 4, 0);
		HIPOK(c, hipMemcpy(SA.data(), c->d_SA.p, (size_t)ns * 4, hipMemcpyDeviceToHost));
		return 0;
	};
	if (!sa_on_device) {
		SA.assign((size_t)ns + 4, 0);
		if (sa) {
			for (uint32_t i = 0; i < ns; i++) {
				if (sa[i] < 0 || sa[i] >= (int64_t)ns) return c->fail("suffix array entry %u out of range", i);
				SA[i] = (uint32_t)sa[i];
			}
		} else {
			// host cores (north star); everything below is on the device
			if (host_S()) return 1;
			WorkerPool &pool = workers(c);
			auto par = [&](size_t nt, const std::function<void(size_t)> &f) { pool.run(nt, f); };
			suffix_array_u32_par(S.data(), ns, SA.data(), par, std::max<size_t>(1, pool.size()));
		}
		HIPOK(c, hipMemcpyAsync(c->d_SA.p, SA.data(), SA.size() * 4, hipMemcpyHostToDevice, st));
	}
	double t2 = now_ms();
	uint32_t k = c->opt_kmer ? c->opt_kmer : choose_k(ns);
	if (threshold == 0) threshold = min_anchor_length(0.025, (double)gc_count / (double)L, ns); // gc_content, sequence.cxx:152-165
	const uint64_t codes = (uint64_t)1 << (2 * k);
	{
		const double ta = now_ms();
		HIPOK(c, c->d_SAX.ensure((size_t)ns + 4));
		HIPOK(c, c->d_LCP.ensure((size_t)ns + 1 + 4));
		HIPOK(c, c->d_T.ensure(codes + 1 + 4 + kmer_table_scratch(k)));
		HIPOK(c, c->d_SLOT.ensure(codes * SLOT_RECS));
		c->stats["ms:ref_alloc"] += now_ms() - ta; // hipMalloc of tens of GB stalls when other processes have just released as much
	}
	HIPOK(c, hipMemsetAsync(c->a_misc.p, 0, 64, st));
	HIPOK(c, hipMemsetAsync(c->d_LCP.p, 0, ((size_t)ns + 1 + 4) * 4, st));
	// LCP by direct comparison of neighbouring suffixes, capped at the 16-bit clip of the
	// SAX records; only a repeat of >= 64 kbp needs the exact values (host, Kasai)
	launch_lcp(c->d_S.p, c->d_SA.p, ns, 0xffffu, c->d_LCP.p, c->a_misc.p, st);
	uint32_t capped = 0;
	HIPOK(c, hipMemcpyAsync(&capped, c->a_misc.p, 4, hipMemcpyDeviceToHost, st));
	HIPOK(c, hipStreamSynchronize(st));
	if (capped) {
		std::vector<uint32_t> LCP((size_t)ns + 1 + 4, 0);
		if (host_sa() || host_S()) return 1;
		lcp_kasai(S.data(), ns, SA.data(), LCP.data());
		HIPOK(c, hipMemcpyAsync(c->d_LCP.p, LCP.data(), LCP.size() * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, hipStreamSynchronize(st));
		c->stats["ref:lcp_from_host"] = 1;
	}
	double t2a = now_ms();
	launch_kmer_table(c->d_S.p, ns, k, c->d_T.p, c->d_T.p + codes + 1 + 4, st);
	launch_sax(c->d_S.p, c->d_SA.p, c->d_LCP.p, ns, c->d_SAX.p, st);
	{
		hipLaunchKernelGGL(build_slots_kernel, dim3((uint32_t)((codes + 255) / 256)), dim3(256), 0, st, c->d_T.p,
						   c->d_SAX.p, ns, codes, c->d_SLOT.p);
	}
	c->abs_built = false; // the absence table is built when a call first asks for it (option "absent_table")
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipStreamSynchronize(st));
	double t2b = now_ms();
	{ // the lean chain's view of S: 2-bit codes and the sorted non-ACGT positions, then n
		const size_t swords = ((size_t)ns + 64) / 16; // S's buffer is ns + 64 bytes
		HIPOK(c, c->d_S2.ensure(swords + 64));
		HIPOK(c, hipMemsetAsync(c->d_S2.p, 0, (swords + 64) * 4, st));
		launch_pack2(c->d_S.p, (uint64_t)swords * 16, c->d_S2.p, st);
		HIPOK(c, c->d_badoff.ensure(2));
		HIPOK(c, c->d_badscr.ensure(8));
		const uint64_t zero = 0;
		HIPOK(c, hipMemcpyAsync(c->d_badoff.p, &zero, 8, hipMemcpyHostToDevice, st));
		DevBuf<uint32_t> lenbuf;
		HIPOK(c, lenbuf.ensure(2));
		HIPOK(c, hipMemcpyAsync(lenbuf.p, &ns, 4, hipMemcpyHostToDevice, st));
		std::vector<uint32_t> off;
		int rc = bad_lists(c, c->d_S.p, c->d_badoff.p, lenbuf.p, std::vector<uint64_t>{ns}, c->d_SBAD, off, 1);
		lenbuf.release();
		if (rc) return 1;
		HIPOK(c, hipMemcpy(c->d_SBAD.p + off[1], &ns, 4, hipMemcpyHostToDevice)); // the end of S closes the list
		c->nsb = off[1] + 1;
		HIPOK(c, hipMemcpy(&c->sb_first, c->d_SBAD.p, 4, hipMemcpyDeviceToHost));
		// The reference's 6-mer cache bug (esa.cxx:174-199) needs two contig joins behind the same few
		// nucleotides: looked for only when S holds a '!' at all (nsb counts '#', the end and the '!'s).
		c->cache_quirk = false;
		c->stats["ms:ref_packed_view"] += now_ms() - t2b;
		// ... and a nucleotide string (up to 5 letters) all of whose occurrences go on with the same byte: when every
		// string that occurs is followed by two different bytes or more, the cache's walk never fast-forwards
		// (esa.cxx:150-155 is the only branch taken) and the exact check over the suffix array is not needed
		bool suspect = false;
		for (uint32_t m : next_masks) suspect = suspect || (m && !(m & (m - 1)));
		c->stats["ref:cache_quirk_exact_check"] = (c->nsb > 2 && suspect) ? 1 : 0;
		c->nquirk = 0;
		if (c->nsb > 2 && suspect) {
			if (host_sa() || host_S()) return 1;
			const std::vector<CacheQuirk> qs = esa_cache_quirks(S.data(), ns, SA.data());
			c->cache_quirk = !qs.empty();
			if (!qs.empty()) {
				std::vector<U4> tab;
				for (const CacheQuirk &e : qs) tab.push_back(U4{e.prefix, e.k | (e.depth << 8), e.lo, e.hi});
				HIPOK(c, c->d_quirk.ensure(tab.size()));
				HIPOK(c, hipMemcpy(c->d_quirk.p, tab.data(), tab.size() * sizeof(U4), hipMemcpyHostToDevice));
				c->nquirk = (uint32_t)tab.size();
			}
		}
		c->stats["ref:cache_quirk_entries"] = c->nquirk;
		c->stats["ref:cache_quirk"] = c->cache_quirk ? 1 : 0;
	}
	double t3 = now_ms();
	c->ref_idx = ref_idx;
	c->L = (uint32_t)L;
	c->ns = ns;
	c->k = k;
	c->threshold = (uint32_t)threshold;
	c->have_ref = true;
	c->plan_valid = false;
	c->homs_staged = false;
	c->att_homs = nullptr; // lists that only lived in an attached buffer are gone
	c->att_rng_on_device = false;
	c->host_stale.clear();
	c->stats["ms:ref_fetch"] += t1 - t0;
	c->stats["ms:ref_suffix_array"] += t2 - t1;
	c->stats["ms:ref_lcp_table"] += t3 - t2;
	c->stats["ms:ref_lcp"] += t2a - t2;
	c->stats["ms:ref_slots"] += t2b - t2a;
	c->stats["ms:ref_total"] += now_ms() - t0;
	c->stats["ref:sa_on_device"] = sa_on_device ? 1 : 0;
	c->stats["ref:sa_rounds"] = sa_rounds;
	c->stats["ref:k"] = k;
	c->stats["ref:threshold"] = (double)threshold;
	c->stats["ref:size"] = ns;
	return 0;
}

size_t phylo_threshold(const phylo_ctx *c) { return c ? c->threshold : 0; }

int phylo_reference_suffix_array(phylo_ctx *c, int64_t *sa)
{
	if (!c) return 1;
	if (!c->have_ref) return c->fail("phylo_reference_suffix_array: no reference set");
	if (!sa) return c->fail("null buffer");
	HIPOK(c, hipSetDevice(c->device));
	std::vector<uint32_t> tmp(c->ns);
	HIPOK(c, hipMemcpy(tmp.data(), c->d_SA.p, (size_t)c->ns * 4, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < c->ns; i++) sa[i] = tmp[i];
	return 0;
}

int phylo_reference_cache_quirk(const phylo_ctx *c) { return c && c->have_ref && c->cache_quirk ? 1 : 0; }

// ───────────────────────── phase A ─────────────────────────

__global__ void compact_raw_kernel(const RawHom *__restrict__ src, const uint64_t *__restrict__ src_base,
								   const uint32_t *__restrict__ cnt, const uint64_t *__restrict__ dst_base,
								   RawHom *__restrict__ dst)
{
	const uint32_t j = blockIdx.x;
	const RawHom *s = src + src_base[j];
	RawHom *d = dst + dst_base[j];
	for (uint32_t t = threadIdx.x; t < cnt[j]; t += blockDim.x) d[t] = s[t];
}

static int ensure_host_lists(phylo_ctx *c, size_t g0, size_t g1);

// The pileup of part `part` of `nparts`: a range of 64-window tiles of the reference.
// Every part projects and compares ALL genomes over its own range, so both kernels
// shrink with the number of parts and the partial tallies simply add up.
static int make_pileup(phylo_ctx *c, size_t part, size_t nparts, Pileup *out)
{
	Pileup P;
	P.N = (uint32_t)c->n;
	P.Npad = (uint32_t)((c->n + 63) / 64 * 64);
	P.L = c->L;
	uint32_t Wall = (c->L + 31) / 32;
	uint32_t ntile = (Wall + 63) / 64;
	uint32_t t0 = (uint32_t)((uint64_t)ntile * part / nparts), t1 = (uint32_t)((uint64_t)ntile * (part + 1) / nparts);
	P.w0 = t0 * 64;
	uint32_t wend = std::min<uint32_t>(Wall, t1 * 64);
	P.W = wend > P.w0 ? wend - P.w0 : 0;
	size_t plane_words = (size_t)P.W * P.Npad;
	HIPOK(c, c->b_planes.ensure(plane_words * 5));
	for (int p = 0; p < 5; p++) P.plane[p] = c->b_planes.p + plane_words * p;
	*out = P;
	return 0;
}

// defer: when this call covers every genome and leaves lists and projection on the device, do not wait for its flags —
// the caller queues phase B behind it and reads them with the result (phylo_anchor_compare); anchor_pending says so.
static int anchor_impl(phylo_ctx *c, size_t q_begin, size_t q_end, bool defer)
{
	if (!c) return 1;
	c->anchor_pending = false;
	if (!c->have_ref) return c->fail("phylo_anchor: no reference set");
	if (q_begin > q_end || q_end > c->n) return c->fail("phylo_anchor: bad query range");
	HIPOK(c, hipSetDevice(c->device));
	size_t nq = q_end - q_begin;
	if (nq == 0) return 0;
	// Lists of an earlier call that still live only in this context's device buffer (about to be
	// reused) and lie outside the range computed now are read back first.  Lists attached from a
	// caller's buffer are not: that buffer is borrowed only until this call.
	if (!c->host_stale.empty() && c->att_homs == c->b_homs.p) {
		if (ensure_host_lists(c, 0, q_begin) || ensure_host_lists(c, q_end, c->n)) return 1;
	}
	double t0 = now_ms();
	const bool quirk_mode = c->nquirk > 0 && c->opt_cache_quirk != 0; // (see below, where the chains' tables are set up)
	const bool lean_chains = c->anchor_kernel != 0 || quirk_mode;

	hipStream_t st = c->stream;
	if (!c->plan_valid || c->plan_qb != q_begin || c->plan_qe != q_end) {
		// chunk plan, output capacities and their device copies: rebuilt only when the
		// query range, the genomes or the reference change
		std::vector<uint32_t> qlen(nq);
		std::vector<uint64_t> qoff(nq);
		for (size_t j = 0; j < nq; j++) {
			qlen[j] = (uint32_t)c->glen[q_begin + j];
			qoff[j] = c->goff[q_begin + j];
			// The subject is one of the queries (src/phylonium.cxx:287).  Against itself the
			// chain is one lucky anchor of the whole length (process.cxx:227-242 at q = 0), so
			// its list is written directly below; as a GPU query it would make every
			// speculative chunk compare to the end of the genome.
			if (q_begin + j == c->ref_idx) qlen[j] = 0;
		}
		// Blocks of the chain kernels per CU, for the plan and for the launch alike.  The kernels are bound by the
		// instructions they issue, not by waiting, as long as the k-mer slot table is the 1-4 GB of k <= 13: a fourth
		// wavefront on a SIMD then only makes every trip of the other three longer, and since the speculative kernel
		// ends with its slowest chain, three blocks with longer chunks finish earlier than four with shorter ones
		// (C3: 3.51 -> 3.29 ms, C4: 13.06 -> 12.38, 128 x 20 Mbp: 8.49 -> 7.64, close or distant genomes alike).
		// The 17 GB table of k = 14 (C5's 100 Mbp subject) answers slowly enough for the fourth to pay: 21.9 against 24.1 ms.
		int per_cu_cap = c->k >= 14 ? 4 : 3;
		if (const char *e = getenv("PHY_SPEC_PER_CU")) per_cu_cap = std::max(1, atoi(e)); // experiments
		c->plan_spec_per_cu = per_cu_cap;
		const int resident = lean_chains ? std::min(lean_spec_resident_blocks(c->n_cu), per_cu_cap * c->n_cu) : spec_resident_blocks(c->n_cu);
		// groups of queries that go through phase A one behind the other: whole projection tiles, balanced by chunks
		const uint32_t tile_q = project_genomes_per_tile();
		int pg = c->opt_pipeline_groups;
		if (pg == 0) pg = 1;
		pg = (int)std::min<size_t>((size_t)pg, nq / (2 * tile_q));
		if (!lean_chains || c->filter_mode == 1 || pg < 1) pg = 1;
		c->plan = plan_chunks(qlen, c->threshold, c->opt_chunk, (uint32_t)resident * 256u, c->opt_chunk_tail,
							  lean_chains ? (uint32_t)c->n_cu * 256u : 0u, (uint32_t)pg);
		if (!c->plan.C) return c->fail("phase A: more than 2^32 anchor log slots");
		{
			c->plan_gb.assign(pg + 1, 0);
			c->plan_gb[pg] = (uint32_t)nq;
			for (int g = 1; g < pg; g++) {
				const uint32_t want = (uint32_t)((uint64_t)c->plan.nchunks * g / pg);
				uint32_t j = (uint32_t)(std::lower_bound(c->plan.qchunk0.begin(), c->plan.qchunk0.begin() + nq, want) - c->plan.qchunk0.begin());
				j = (j + tile_q / 2) / tile_q * tile_q;
				c->plan_gb[g] = std::min<uint32_t>(std::max(j, c->plan_gb[g - 1]), (uint32_t)nq);
			}
			c->plan_item0.clear();
			if (pg > 1) c->plan_item0 = group_items(c->plan, c->plan_gb);
		}
		const ChunkPlan &P = c->plan;
		// an emitted homology spans >= 2*threshold query positions
		c->plan_out_base.assign(nq + 1, 0);
		std::vector<uint32_t> out_cap(nq);
		uint64_t raw_total = 0;
		for (size_t j = 0; j < nq; j++) {
			c->plan_out_base[j] = raw_total;
			out_cap[j] = qlen[j] / (2 * c->threshold) + 2;
			raw_total += out_cap[j];
		}
		c->plan_out_base[nq] = raw_total;
		c->plan_raw_total = raw_total;
		uint32_t nchp = P.nchunks;
		HIPOK(c, c->a_qoff.ensure(nq));
		HIPOK(c, c->a_qlen.ensure(nq));
		HIPOK(c, c->a_qchunk0.ensure(nq + 1));
		HIPOK(c, c->a_items.ensure(nchp + 1));
		HIPOK(c, c->a_chunk_query.ensure(nchp + 1));
		HIPOK(c, c->a_spec_cnt.ensure(nchp + 1));
		// one visited bit per byte of the genome buffer (chains address it by buffer offset)
		HIPOK(c, c->a_visited.ensure((c->goff[c->n - 1] + c->glen[c->n - 1]) / 32 + 8));
		HIPOK(c, c->a_misc.ensure(32)); // [0..8): counters and flags, [8..16): the groups' bridge counters, [16..24): their chain counters
		HIPOK(c, c->a_spec_anchors.ensure(P.anchor_slots + 1));
		HIPOK(c, c->a_qnb.ensure(nq));
		HIPOK(c, c->a_qanc0.ensure(nq));
		HIPOK(c, c->a_spec_exit.ensure(nchp + 1));
		HIPOK(c, c->a_bridge.ensure(nchp + 1));
		HIPOK(c, c->a_pool.ensure(nchp / 4 + 4096));
		HIPOK(c, c->a_raw.ensure(raw_total + 1));
		HIPOK(c, c->a_out_base.ensure(nq + 1));
		HIPOK(c, c->a_cmp_base.ensure(nq + 1));
		HIPOK(c, c->a_out_cap.ensure(nq));
		HIPOK(c, c->a_out_cnt.ensure(nq));
		HIPOK(c, hipMemcpyAsync(c->a_qoff.p, qoff.data(), nq * 8, hipMemcpyHostToDevice, st));
		HIPOK(c, hipMemcpyAsync(c->a_qlen.p, qlen.data(), nq * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, hipMemcpyAsync(c->a_qchunk0.p, P.qchunk0.data(), (nq + 1) * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, hipMemcpyAsync(c->a_qnb.p, P.qnb.data(), nq * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, hipMemcpyAsync(c->a_qanc0.p, P.qanc0.data(), nq * 4, hipMemcpyHostToDevice, st));
		if (nchp) {
			HIPOK(c, hipMemcpyAsync(c->a_items.p, P.items.data(), (size_t)nchp * 4, hipMemcpyHostToDevice, st));
			HIPOK(c, hipMemcpyAsync(c->a_chunk_query.p, P.chunk_query.data(), (size_t)nchp * 4, hipMemcpyHostToDevice, st));
		}
		HIPOK(c, hipMemcpyAsync(c->a_out_base.p, c->plan_out_base.data(), (nq + 1) * 8, hipMemcpyHostToDevice, st));
		HIPOK(c, hipMemcpyAsync(c->a_out_cap.p, out_cap.data(), nq * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, hipStreamSynchronize(st)); // the host vectors above go out of scope
		c->plan_qb = q_begin;
		c->plan_qe = q_end;
		c->plan_valid = true;
	}
	const ChunkPlan &P = c->plan;
	const uint32_t nch = P.nchunks;
	const uint32_t pool_blocks = nch / 4 + 4096;
	uint64_t total = 0;
	for (size_t j = 0; j < nq; j++) total += c->glen[q_begin + j];
	HIPOK(c, hipMemsetAsync(c->a_misc.p, 0, 32 * 4, st));
	if (nch) { // the words of this call's queries (their genomes lie back to back in the buffer)
		const uint64_t w0 = c->goff[q_begin] / 32, w1 = (c->goff[q_end - 1] + c->glen[q_end - 1]) / 32 + 1;
		HIPOK(c, hipMemsetAsync(c->a_visited.p + w0, 0, (size_t)(w1 - w0) * 4, st));
	}

	PhaseA A;
	A.qbase = c->d_genomes;
	A.qoff = c->a_qoff.p;
	A.qlen = c->a_qlen.p;
	A.qchunk0 = c->a_qchunk0.p;
	A.items = c->a_items.p;
	A.chunk_query = c->a_chunk_query.p;
	A.nchunks = nch;
	A.C = P.C;
	A.Cs = P.Cs;
	A.cap = P.cap;
	A.caps = P.caps;
	A.qnb = c->a_qnb.p;
	A.qanc0 = c->a_qanc0.p;
	A.spec_anchors = c->a_spec_anchors.p;
	A.spec_cnt = c->a_spec_cnt.p;
	A.spec_exit = c->a_spec_exit.p;
	A.visited = c->a_visited.p;
	A.bridge = c->a_bridge.p;
	A.pool = c->a_pool.p;
	A.pool_blocks = pool_blocks;
	A.fetch = c->a_misc.p;          // [0] spec, [1] bridge
	A.pool_next = c->a_misc.p + 2;
	A.error = c->a_misc.p + 3;
	A.overrun = c->a_misc.p + 4;
	RefIndex R = {c->d_S.p, c->d_SAX.p, c->d_LCP.p, c->d_SLOT.p, c->ns, c->k, c->threshold};
	// A subject on which the reference's 6-mer cache holds over-deep intervals (esa.cxx:174-199): the reference's
	// answers there are reproduced by the lean chains' slow resolver, so every step goes through it (such subjects
	// are a few kbp in several contigs; option "cache_quirk" = 0 computes the true longest matches instead).
	if (c->opt_absent_table && !c->abs_built) { // the digest of every empty bucket, once per subject (it depends on the threshold: a match that long is an anchor)
		HIPOK(c, c->d_ABS.ensure(((size_t)1 << (2 * c->k)) + 16));
		launch_build_absent(R, c->d_ABS.p, st);
		HIPOK(c, hipGetLastError());
		c->abs_built = true;
	}
	LeanIndex X = {c->d_S2.p, c->d_SBAD.p, c->nsb, c->ns, c->sb_first, c->d_Q2.p, c->d_QBAD.p, c->d_qbad_off.p + q_begin,
				   (uint32_t)(c->lean_force_slow || quirk_mode), nullptr, quirk_mode ? c->d_quirk.p : nullptr, quirk_mode ? c->nquirk : 0u,
				   c->opt_lean_batch, c->opt_absent_table ? c->d_ABS.p : (const uint8_t *)nullptr};
#ifdef PHY_LEAN_TIMING
	static unsigned long long *dbg_buf = nullptr;
	const size_t dbg_words = 16 + 4 * 8192 + 64;
	if (!dbg_buf) (void)hipMalloc((void **)&dbg_buf, dbg_words * 8);
	(void)hipMemsetAsync(dbg_buf, 0, dbg_words * 8, st);
	X.dbg = dbg_buf;
#endif
	const bool lean = lean_chains;

	double t1 = now_ms();
	const bool dbg = getenv("PHY_DEBUG_SYNC") != nullptr; // name the kernel a hang is in
	if (dbg) {
		hipError_t e = hipStreamSynchronize(st);
		fprintf(stderr, "[phylonium_amd] phase A set up (%s): %zu queries, %u chunks of %u, cap %u, k %u, |S| %u, threshold %u\n",
				hipGetErrorString(e), nq, nch, P.C, P.cap, c->k, c->ns, c->threshold);
	}
	auto dbg_sync = [&](const char *what) {
		if (!dbg) return;
		hipError_t e = hipStreamSynchronize(st);
		fprintf(stderr, "[phylonium_amd] %s finished at +%.1f ms (%s), chunks %u of %u positions\n", what, now_ms() - t1,
				hipGetErrorString(e), nch, P.C);
	};
	const bool pipelined = lean && nch && c->plan_gb.size() > 2 && c->filter_mode != 1;
	if (nch && !pipelined) {
		{
			KernelSpan s(c, "anchor_spec");
			if (lean) launch_lean_spec(A, R, X, c->n_cu, st, c->plan_spec_per_cu * c->n_cu);
			else launch_spec(A, R, c->n_cu, st);
		}
		dbg_sync("anchor_spec");
		if (lean) {
			KernelSpan s(c, "anchor_overruns");
			launch_lean_overruns(A, R, (uint32_t)nq, st);
		}
	}
	// Sort + chain filter: on the device unless the host is asked for (option "filter" = 1).  Round 1 sent calls
	// with fewer than 128 queries to the host pool (the device's dependent scan took ~0.4 ms whatever the
	// number); stretch by stretch a list takes ~60 us there, and long lists have a kernel of their own.
	const bool device_filter = c->filter_mode != 1;
	const bool full = q_begin == 0 && q_end == c->n;
	const bool tail_eager = device_filter && full && c->backend == 0;
	const uint32_t ref_local = (c->ref_idx >= q_begin && c->ref_idx < q_end) ? (uint32_t)(c->ref_idx - q_begin) : 0xffffffffu;
	// groups of queries for the tail: whole projection tiles, balanced by chunks
	const uint32_t tsz_q = project_genomes_per_tile();
	int tgroups = 1;
	if (pipelined) {
		tgroups = (int)c->plan_gb.size() - 1;
	} else if (lean && device_filter && nch && c->opt_tail_groups > 1) {
		tgroups = c->opt_tail_groups;
		tgroups = (int)std::min<size_t>((size_t)tgroups, nq / (2 * tsz_q));
		if (tgroups < 1) tgroups = 1;
	}
	std::vector<uint32_t> gb(tgroups + 1, 0); // group g: queries [gb[g], gb[g+1])
	gb[tgroups] = (uint32_t)nq;
	if (pipelined) gb = c->plan_gb;
	for (int g = 1; g < tgroups && !pipelined; g++) {
		const uint32_t want = (uint32_t)((uint64_t)nch * g / tgroups);
		uint32_t j = (uint32_t)(std::lower_bound(P.qchunk0.begin(), P.qchunk0.begin() + nq, want) - P.qchunk0.begin());
		j = (j + tsz_q / 2) / tsz_q * tsz_q;
		gb[g] = std::min<uint32_t>(std::max(j, gb[g - 1]), (uint32_t)nq);
	}
	// queries of tens of Mbp leave more entries than a block's LDS holds (~330 per Mbp): their lists go through the
	// long-list kernel and its scratch slots; with shorter queries a list that long is an oddity and goes to the host
	bool long_lists = false;
	if (device_filter && c->opt_filter_kernel == 0) {
		uint64_t longest = 0;
		for (size_t j = 0; j < nq; j++)
			if (q_begin + j != c->ref_idx) longest = std::max<uint64_t>(longest, c->glen[q_begin + j]);
		long_lists = longest > 6000000;
		if (long_lists) HIPOK(c, c->a_long.ensure(long_filter_scratch_bytes()));
	}
	Pileup TP;
	if (device_filter) {
		HIPOK(c, c->b_homs.ensure(c->plan_raw_total + nq + 1));
		HIPOK(c, c->b_hom_rng.ensure(2 * std::max(nq, c->n)));
		HIPOK(c, c->a_flt.ensure(nq + 1));
		HIPOK(c, c->h_rng.ensure(3 * nq + 16));
		HIPOK(c, hipMemsetAsync(c->a_flt.p, 0, 4, st));
		if (tail_eager) {
			if (make_pileup(c, 0, 1, &TP)) return 1;
			HIPOK(c, c->b_flag.ensure(4));
			HIPOK(c, c->b_first.ensure(project_index_entries(TP) + 1));
			HIPOK(c, hipMemsetAsync(c->b_flag.p, 0, 16, st));
			c->eager_five = c->pileup_five && c->opt_pairs_kernel != 0; // (the matrix-core path lists the '!' instead: compare_pileup)
			HIPOK(c, c->b_bang.ensure(2 * (size_t)c->bang_cap + 2));
		}
	}
	if (pipelined) {
		if (!c->tail_stream[0]) HIPOK(c, hipStreamCreateWithFlags(&c->tail_stream[0], hipStreamNonBlocking));
		for (int g = 0; g <= phylo_ctx::PIPE_GROUPS; g++)
			if (!c->pipe_event[g]) HIPOK(c, hipEventCreateWithFlags(&c->pipe_event[g], hipEventDisableTiming));
	} else if (tgroups > 1) {
		for (int g = 0; g < tgroups; g++) {
			if (g && !c->tail_stream[g - 1]) HIPOK(c, hipStreamCreateWithFlags(&c->tail_stream[g - 1], hipStreamNonBlocking));
			if (!c->tail_event[g]) HIPOK(c, hipEventCreateWithFlags(&c->tail_event[g], hipEventDisableTiming));
		}
		HIPOK(c, hipEventRecord(c->tail_event[0], st)); // the speculative chains (and their overruns) are done
	}
	for (int g = 0; g < tgroups; g++) {
		hipStream_t sg = pipelined ? c->tail_stream[0] : g ? c->tail_stream[g - 1] : st;
		const uint32_t j0 = gb[g], j1 = gb[g + 1];
		if (pipelined) { // this group's speculative chains on the context's stream; its tail follows on the other one
			{
				KernelSpan s(c, "anchor_spec");
				launch_lean_spec_range(A, R, X, c->plan_item0[g], c->plan_item0[g + 1] - c->plan_item0[g], 16u + (uint32_t)g, c->n_cu, st,
									   c->plan_spec_per_cu * c->n_cu);
			}
			{
				KernelSpan s(c, "anchor_overruns");
				launch_lean_overruns_range(A, R, P.qchunk0[j0], P.qchunk0[j1], j0, j1, st);
			}
			HIPOK(c, hipEventRecord(c->pipe_event[g], st));
			HIPOK(c, hipStreamWaitEvent(sg, c->pipe_event[g], 0));
		} else if (g) {
			HIPOK(c, hipStreamWaitEvent(sg, c->tail_event[0], 0));
		}
		if (nch) {
			KernelSpan s(c, "anchor_bridge", sg);
			if (!lean) launch_bridge(A, R, c->n_cu, st);
			else if (tgroups == 1) launch_lean_bridge(A, R, X, c->n_cu, st);
			else launch_lean_bridge_range(A, R, X, P.qchunk0[j0], P.qchunk0[j1], 8u + (uint32_t)g, c->n_cu, sg);
		}
		{
			KernelSpan s(c, "anchor_fold", sg);
			// Blocks per query.  A block's time is its windows' walks (a latency chain every block of the query
			// repeats) plus its share of the anchors; with a block per CU or more there is nothing to gain from
			// splitting (measured, C3's 256 queries with two blocks each: 0.17 -> 0.21 ms), with a handful of queries
			// the idle CUs take a part each (c2like's 29 queries: 0.167 -> 0.100 ms; C5's 64: 2.65 -> 2.28 ms)
			uint32_t fold_nb = c->opt_fold_blocks;
			if (!fold_nb) fold_nb = 2 * (j1 - j0) <= (uint32_t)c->n_cu ? (uint32_t)std::min<size_t>(8, (size_t)c->n_cu / (j1 - j0)) : 1u;
			launch_fold(A, j0, j1, c->L, c->threshold, c->a_raw.p, c->a_out_base.p, c->a_out_cap.p, c->a_out_cnt.p, sg, fold_nb);
		}
		if (device_filter) {
			// reverseEh + sort + filter_overlaps_max on the device (filter_kernels.hip).  The lists stay
			// there in the 16-byte device form, the projection (phase B's first kernel, for the whole
			// reference = part 0 of 1) follows at once when this call covers all genomes, and the host
			// reads a list back only when somebody asks for it.  A query whose list has two entries
			// with the same projected start, or more entries than the kernel holds, is flagged: then
			// everything below runs on the host as it always did.
			{
				KernelSpan s(c, "anchor_filter", sg);
				launch_sort_filter(c->a_raw.p, c->a_out_base.p, c->a_out_cnt.p, j0, j1, c->L, c->threshold, ref_local, c->b_homs.p,
								   c->b_hom_rng.p, c->a_flt.p, c->a_flt.p + 1, sg, c->opt_filter_kernel, long_lists ? 1 : 0);
				if (long_lists)
					launch_sort_filter_long(c->a_raw.p, c->a_out_base.p, c->a_out_cnt.p, j0, j1, c->L, c->b_homs.p, c->b_hom_rng.p,
											c->a_flt.p, c->a_flt.p + 1, c->a_long.p, c->a_misc.p + 7, sg);
			}
			if (tail_eager && j1 > j0) {
				launch_tile_index(TP, query_src(c), c->b_homs.p, c->b_hom_rng.p, c->b_first.p, j0, j1, sg);
				KernelSpan s(c, c->eager_five ? "pileup_project5" : "pileup_project", sg);
				launch_project(TP, c->eager_five, query_src(c), c->b_homs.p, c->b_hom_rng.p, c->b_first.p, c->b_flag.p,
							   j0 / tsz_q, g + 1 == tgroups ? TP.Npad / tsz_q : j1 / tsz_q, sg, c->b_bang.p, c->bang_cap);
			}
		}
		if (pipelined) {
			if (g + 1 == tgroups) { // everything behind this call on the context's stream waits for the last tail
				HIPOK(c, hipEventRecord(c->pipe_event[phylo_ctx::PIPE_GROUPS], sg));
				HIPOK(c, hipStreamWaitEvent(st, c->pipe_event[phylo_ctx::PIPE_GROUPS], 0));
			}
		} else if (g) {
			HIPOK(c, hipEventRecord(c->tail_event[g], sg));
			HIPOK(c, hipStreamWaitEvent(st, c->tail_event[g], 0));
		}
	}
	dbg_sync("anchor tail (bridge, fold, filter, projection)");
#ifdef PHY_LEAN_TIMING
	{
		std::vector<unsigned long long> hv(dbg_words);
		unsigned long long *h = hv.data();
		(void)hipStreamSynchronize(st);
		(void)hipMemcpy(h, X.dbg, dbg_words * 8, hipMemcpyDeviceToHost);
		for (int m = 0; m < 2; m++) {
			const unsigned long long *pt = h + 16 + 4 * 8192 + m * 16, trips = h[16 + 4 * 8192 + 32 + m];
			fprintf(stderr, "[lean timing] mode %d trips %llu; share of trips with a lane in STEP %.3f SEARCH %.3f SCAN %.3f EXT %.3f REFILL %.3f SLOW %.3f SLOWEXT %.3f, lucky STEP %.3f; lanes per trip: STEP %.1f SEARCH %.2f SCAN %.2f EXT %.1f REFILL %.2f SLOW %.3f\n",
					m, trips, (double)pt[0] / trips, (double)pt[1] / trips, (double)pt[2] / trips, (double)pt[3] / trips, (double)pt[4] / trips,
					(double)pt[5] / trips, (double)pt[6] / trips, (double)pt[7] / trips, (double)pt[8] / trips, (double)pt[9] / trips,
					(double)pt[10] / trips, (double)pt[11] / trips, (double)pt[12] / trips, (double)pt[13] / trips);
		}
		for (int m = 0; m < 2; m++) {
			const unsigned long long *e = h + 16 + 4 * 8192 + 34 + m * 6;
			fprintf(stderr, "[lean timing] mode %d EXT lane-trips: first of a lucky check %llu, first of a candidate %llu, later %llu; of the first ones: match < 32 bases %llu, < 48 bases %llu, on to SEARCH/SLOW %llu\n",
					m, e[0], e[1], e[2], e[3], e[4], e[5]);
		}
		if (const char *wf = getenv("PHY_LEAN_WAVES_OUT")) {
			if (FILE *f = fopen(wf, "w")) { // the last call's speculative wavefronts: start, end (10 ns), trips, query
				for (size_t w = 0; w < 8192; w++)
					if (h[16 + 4 * w + 1]) fprintf(f, "%llu %llu %llu %llu\n", h[16 + 4 * w], h[16 + 4 * w + 1], h[16 + 4 * w + 2], h[16 + 4 * w + 3]);
				fclose(f);
			}
		}
		for (int m = 0; m < 2; m++)
			fprintf(stderr, "[lean timing] mode %d waves %llu  Mcycles: bookkeeping %.1f  phase+address %.1f  loads %.1f  digest %.1f  slow %.1f; resolves %llu, long compares %llu\n",
					m, h[m * 8 + 6], h[m * 8 + 0] / 1e6, h[m * 8 + 1] / 1e6, h[m * 8 + 2] / 1e6, h[m * 8 + 3] / 1e6, h[m * 8 + 4] / 1e6,
					h[m * 8 + 5] & 0xffffffffull, h[m * 8 + 5] >> 32);
	}
#endif
	HIPOK(c, hipGetLastError());
	c->homs_staged = false;
	c->eager_valid = false;
	c->att_homs = nullptr; // an attached buffer is only borrowed until the next phase A
	c->att_rng_on_device = false;
	c->host_stale.clear();
	if (device_filter) {
		uint32_t *hr = c->h_rng.p; // [0, 2nq) ranges, [2nq, 3nq) flags, then total and the four misc words
		HIPOK(c, hipMemcpyAsync(hr, c->b_hom_rng.p, 2 * nq * 4, hipMemcpyDeviceToHost, st));
		HIPOK(c, hipMemcpyAsync(hr + 2 * nq, c->a_flt.p + 1, nq * 4, hipMemcpyDeviceToHost, st));
		HIPOK(c, hipMemcpyAsync(hr + 3 * nq, c->a_flt.p, 4, hipMemcpyDeviceToHost, st));
		HIPOK(c, hipMemcpyAsync(hr + 3 * nq + 1, c->a_misc.p, 32, hipMemcpyDeviceToHost, st));
		HIPOK(c, hipGetLastError());
		if (defer && tail_eager) { // (tail_eager: all genomes, device filter, projection queued)
			c->pend_t0 = t0, c->pend_t1 = t1, c->pend_t2 = now_ms(), c->pend_total = (double)total, c->pend_nch = nch, c->pend_C = P.C;
			c->att_homs = c->b_homs.p;
			c->homs_staged = true;
			c->eager_valid = true;
			c->anchor_pending = true;
			return 0;
		}
		if (sync_stream(c)) return 1;
		double t2d = now_ms();
		const uint32_t *dmisc = hr + 3 * nq + 1;
		if (dmisc[3]) return c->fail("phase A scratch overflow (code %u: 1 chunk log, 2 bridge pool, 3 homology buffer)", dmisc[3]);
		size_t flagged = 0;
		for (size_t j = 0; j < nq; j++) flagged += hr[2 * nq + j] != 0;
		if (!flagged) {
			const size_t N = c->n;
			c->att_homs = c->b_homs.p;
			c->att_rng_on_device = false;
			if (c->att_begin.size() != N) {
				c->att_begin.assign(N, 0);
				c->att_count.assign(N, 0);
			}
			c->host_stale.assign(N, 0);
			for (size_t j = 0; j < nq; j++) {
				c->att_begin[q_begin + j] = hr[2 * j];
				c->att_count[q_begin + j] = hr[2 * j + 1] - hr[2 * j];
				c->host_stale[q_begin + j] = 1;
			}
			c->homs_staged = full;
			c->eager_valid = tail_eager;
			c->stats["ms:anchor_setup"] += t1 - t0;
			c->stats["ms:anchor_gpu"] += t2d - t1;
			c->stats["ms:anchor_total"] += now_ms() - t0;
			c->stats["n:anchor_calls"] += 1;
			c->stats["count:query_bases"] += (double)total;
			c->stats["count:chunks"] += nch;
			c->stats["count:filtered_homologies"] += (double)hr[3 * nq];
			c->stats["count:pool_blocks_used"] += dmisc[2];
			c->stats["count:overrun_runs"] += dmisc[5];
			c->stats["count:overrun_bytes_compared"] += dmisc[6];
			c->stats["anchor:chunk"] = P.C;
			return 0;
		}
		c->stats["count:queries_left_to_the_host"] += (double)flagged;
	}
	HIPOK(c, c->h_cnt.ensure(nq + 8));
	uint32_t *cnt = c->h_cnt.p, *misc = c->h_cnt.p + nq;
	HIPOK(c, hipMemcpyAsync(cnt, c->a_out_cnt.p, nq * 4, hipMemcpyDeviceToHost, st));
	HIPOK(c, hipMemcpyAsync(misc, c->a_misc.p, 32, hipMemcpyDeviceToHost, st));
	if (sync_stream(c)) return 1;
	double t2 = now_ms();
	if (misc[3]) return c->fail("phase A scratch overflow (code %u: 1 chunk log, 2 bridge pool, 3 homology buffer)", misc[3]);

	std::vector<uint64_t> cbase(nq + 1);
	uint64_t ctot = 0;
	for (size_t j = 0; j < nq; j++) {
		cbase[j] = ctot;
		ctot += cnt[j];
	}
	cbase[nq] = ctot;
	HIPOK(c, c->h_raw.ensure(ctot + 1));
	// (the workers read and, for lists that arrive out of query order, reorder c->h_raw in place)
	if (ctot) {
		HIPOK(c, c->a_raw_compact.ensure(ctot));
		HIPOK(c, hipMemcpyAsync(c->a_cmp_base.p, cbase.data(), (nq + 1) * 8, hipMemcpyHostToDevice, st));
		{
			KernelSpan s(c, "anchor_compact");
			hipLaunchKernelGGL(compact_raw_kernel, dim3((uint32_t)nq), dim3(256), 0, st, c->a_raw.p, c->a_out_base.p,
							   c->a_out_cnt.p, c->a_cmp_base.p, c->a_raw_compact.p);
		}
		HIPOK(c, hipMemcpyAsync(c->h_raw.p, c->a_raw_compact.p, ctot * sizeof(RawHom), hipMemcpyDeviceToHost, st));
		if (sync_stream(c)) return 1;
	}
	double t3 = now_ms();
	// reverseEh + std::sort + filter_overlaps_max on the host cores (process.cxx:438-443)
	uint64_t border = c->L;
	std::atomic<uint32_t> tie_lists{0};
	// When this call makes every genome's list, phase B's device copy of them is staged
	// here as well: a worker writes its list in the 16-byte device form into pinned
	// memory (slot cbase[j] + j: the raw count bounds the filtered one, and the self
	// query keeps one entry of zero raw ones) and counts its group down; the calling
	// thread sends every finished group off — records and ranges go up through the
	// copy stream, and the projection of those genomes (phase B's first kernel, for
	// the whole reference = part 0 of 1) starts behind them — so upload and projection
	// run while the other lists are still being sorted.
	const bool stage = q_begin == 0 && q_end == c->n && nq > 0;
	c->homs_staged = false;
	c->att_homs = nullptr; // an attached buffer is only borrowed until the next phase A
	c->att_rng_on_device = false;
	c->host_stale.clear();
	// genomes per group: a whole number of projection tiles — three, or an eighth of all of them
	// (measured on C3 and C4: every group pays ~30 us of hand-over between the copy engine and
	// the compute queue, one big group overlaps nothing)
	const size_t tsz = project_genomes_per_tile();
	const size_t gsz = tsz * std::max<size_t>(3, ((nq + tsz - 1) / tsz) / 8);
	const size_t ngroups = stage ? (nq + gsz - 1) / gsz : 0;
	std::vector<std::atomic<uint32_t>> group_left(ngroups);
	std::mutex group_m; // the calling thread sleeps until a group is complete (it used to spin: under the boxes' CPU-time
	std::condition_variable group_cv; // quota a spinning thread takes time from the workers it is waiting for)
	const bool eager = stage && c->backend == 0;
	c->eager_valid = false;
	Pileup EP;
	std::atomic<int> stage_err{0};
	uint32_t *rng = nullptr;
	DevHom *dh = nullptr;
	if (stage) {
		HIPOK(c, c->h_devhom.ensure(ctot + nq + 1));
		HIPOK(c, c->h_rng.ensure(2 * nq));
		HIPOK(c, c->b_homs.ensure(ctot + nq + 1));
		HIPOK(c, c->b_hom_rng.ensure(2 * nq));
		rng = c->h_rng.p;
		dh = c->h_devhom.p;
		for (size_t g = 0; g < ngroups; g++) group_left[g] = (uint32_t)(std::min(nq, (g + 1) * gsz) - g * gsz);
		while (c->copy_events.size() < ngroups) {
			hipEvent_t e;
			HIPOK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
			c->copy_events.push_back(e);
		}
		if (eager) {
			if (make_pileup(c, 0, 1, &EP)) return 1;
			HIPOK(c, c->b_flag.ensure(4));
			HIPOK(c, c->b_first.ensure(project_index_entries(EP) + 1));
			HIPOK(c, hipMemsetAsync(c->b_flag.p, 0, 16, st));
			c->eager_five = c->pileup_five && c->opt_pairs_kernel != 0; // (the matrix-core path lists the '!' instead: compare_pileup)
			HIPOK(c, c->b_bang.ensure(2 * (size_t)c->bang_cap + 2));
		}
	}
	auto stage_list = [&](size_t j, const std::vector<phylo_homology> &list) {
		const size_t o = cbase[j] + j;
		for (size_t t = 0; t < list.size(); t++)
			dh[o + t] = DevHom{(uint32_t)list[t].index_reference_projected, (uint32_t)list[t].index_query,
							   (uint32_t)list[t].length, (uint32_t)list[t].direction};
		rng[2 * j] = (uint32_t)o;
		rng[2 * j + 1] = (uint32_t)(o + list.size());
		if (group_left[j / gsz].fetch_sub(1, std::memory_order_acq_rel) == 1) {
			std::lock_guard<std::mutex> lk(group_m);
			group_cv.notify_all();
		}
	};
	double t_send_done = 0;
	auto send_groups = [&]() {
		for (size_t g = 0; g < ngroups; g++) {
			if (group_left[g].load(std::memory_order_acquire) != 0) {
				std::unique_lock<std::mutex> lk(group_m);
				group_cv.wait(lk, [&] { return group_left[g].load(std::memory_order_acquire) == 0; });
			}
			const size_t j0 = g * gsz, j1 = std::min(nq, (g + 1) * gsz);
			const size_t o0 = cbase[j0] + j0, o1 = cbase[j1] + j1;
			// the upload goes through the copy stream (DMA engine), so group g+1 travels while
			// group g is being projected; an event orders the group's kernels after its upload
			if (hipMemcpyAsync(c->b_homs.p + o0, dh + o0, (o1 - o0) * sizeof(DevHom), hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
				hipMemcpyAsync(c->b_hom_rng.p + 2 * j0, rng + 2 * j0, (j1 - j0) * 8, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
				hipEventRecord(c->copy_events[g], c->copy_stream) != hipSuccess ||
				hipStreamWaitEvent(st, c->copy_events[g], 0) != hipSuccess) {
				stage_err = 1;
			} else if (eager) {
				launch_tile_index(EP, query_src(c), c->b_homs.p, c->b_hom_rng.p, c->b_first.p, (uint32_t)j0, (uint32_t)j1, st);
				KernelSpan s(c, c->eager_five ? "pileup_project5" : "pileup_project");
				launch_project(EP, c->eager_five, query_src(c), c->b_homs.p, c->b_hom_rng.p, c->b_first.p,
							   c->b_flag.p, (uint32_t)(j0 / tsz), g + 1 == ngroups ? EP.Npad / (uint32_t)tsz : (uint32_t)(j1 / tsz), st, c->b_bang.p, c->bang_cap);
			}
		}
		t_send_done = now_ms();
	};
	std::unique_ptr<KernelSpan> stage_span; // GPU-side time from here until the last group's projection is done
	if (stage) stage_span.reset(new KernelSpan(c, "stage_all"));
	workers(c).run(nq, [&](size_t j) {
		std::vector<phylo_homology> &dst = c->homs[q_begin + j];
		if (q_begin + j == c->ref_idx) {
			// anchor_homologies(ref, threshold, subject): homology(0, 0, L), pushed iff
			// last_length / 2 >= threshold (process.cxx:285-292)
			dst.clear();
			if (border / 2 >= c->threshold) dst.push_back(project_homology(RawHom{0, 0, (uint32_t)border}, border));
			if (stage) stage_list(j, dst);
			return;
		}
		static thread_local SortFilterScratch scratch;
		static thread_local std::vector<uint32_t> kept;
		RawHom *r = c->h_raw.p + cbase[j];
		const size_t m = cnt[j];
		// several fold blocks per query hand a list over in the order they got their slots: back to query order,
		// which is what the reference's std::sort is given (process.cxx:438; query positions ascend along a chain)
		if (!std::is_sorted(r, r + m, [](const RawHom &a, const RawHom &b) { return a.iq < b.iq; }))
			std::sort(r, r + m, [](const RawHom &a, const RawHom &b) { return a.iq < b.iq; });
		auto get = [&](size_t i, uint64_t *start, uint64_t *len) {
			*len = r[i].len;
			*start = r[i].iref >= border ? 2 * border + 1 - r[i].len - r[i].iref : r[i].iref;
		};
		if (sort_filter_order(m, get, scratch, kept)) {
			dst.resize(kept.size());
			for (size_t t = 0; t < kept.size(); t++) dst[t] = project_homology(r[kept[t]], border);
		} else { // equal starts: only std::sort on the structs reproduces the reference's order
			std::vector<phylo_homology> hv(m);
			for (size_t t = 0; t < m; t++) hv[t] = project_homology(r[t], border);
			sort_and_filter(hv);
			dst = std::move(hv);
			tie_lists.fetch_add(1, std::memory_order_relaxed);
		}
		if (stage) stage_list(j, dst);
	}, stage ? std::function<void()>(send_groups) : std::function<void()>());
	stage_span.reset();
	if (stage) {
		if (stage_err) return c->fail("staging the homology lists on the device failed");
		HIPOK(c, hipGetLastError());
		c->stats["ms:stage_send_done"] += t_send_done - t3;
		c->homs_staged = true;
		c->eager_valid = eager;
	}
	double t4 = now_ms();
	c->stats["ms:anchor_setup"] += t1 - t0;
	c->stats["ms:anchor_gpu"] += t2 - t1;
	c->stats["ms:anchor_copyback"] += t3 - t2;
	c->stats["ms:host_sort_filter"] += t4 - t3;
	c->stats["ms:anchor_total"] += t4 - t0;
	c->stats["n:anchor_calls"] += 1;
	c->stats["count:query_bases"] += (double)total;
	c->stats["count:chunks"] += nch;
	c->stats["count:raw_homologies"] += (double)ctot;
	c->stats["count:lists_with_equal_starts"] += (double)tie_lists.load();
	c->stats["count:pool_blocks_used"] += misc[2];
	c->stats["count:overrun_runs"] += misc[5];
	c->stats["count:overrun_bytes_compared"] += misc[6];
	(void)pool_blocks;
	c->stats["anchor:chunk"] = P.C;
	return 0;
}

static int unpack_lists(phylo_ctx *c, size_t q_begin, size_t q_end, const uint64_t *counts,
						const phylo_packed_homology *buf);

// Host lists of genomes [g0, g1) that only exist as attached device records: fetch them.
static int fetch_att_ranges(phylo_ctx *c);
static int ensure_host_lists(phylo_ctx *c, size_t g0, size_t g1)
{
	if (c->host_stale.empty()) return 0;
	if (fetch_att_ranges(c)) return 1;
	for (size_t g = g0; g < g1; g++) {
		if (!c->host_stale[g]) continue;
		const size_t m = c->att_count[g];
		std::vector<phylo_packed_homology> tmp(m);
		if (m) {
			HIPOK(c, hipSetDevice(c->device));
			HIPOK(c, hipMemcpy(tmp.data(), c->att_homs + c->att_begin[g], m * sizeof(DevHom), hipMemcpyDeviceToHost));
		}
		uint64_t one = m;
		c->host_stale[g] = 0;
		if (unpack_lists(c, g, g + 1, &one, tmp.data())) return 1;
	}
	return 0;
}

int phylo_get_homologies(phylo_ctx *c, size_t j, const phylo_homology **h, size_t *n)
{
	if (!c || !h || !n) return 1;
	if (j >= c->n) return c->fail("genome index out of range");
	if (ensure_host_lists(c, j, j + 1)) return 1;
	*h = c->homs[j].data();
	*n = c->homs[j].size();
	return 0;
}

int phylo_set_homologies(phylo_ctx *c, size_t j, const phylo_homology *h, size_t n)
{
	if (!c) return 1;
	if (j >= c->n) return c->fail("genome index out of range");
	if (n && !h) return c->fail("null homology list");
	c->homs[j].assign(h, h + n);
	c->homs_staged = false;
	if (!c->host_stale.empty()) {
		if (ensure_host_lists(c, 0, j) || ensure_host_lists(c, j + 1, c->n)) return 1;
		c->host_stale.clear();
	}
	return 0;
}

int phylo_export_homologies(phylo_ctx *c, size_t q_begin, size_t q_end, uint64_t *counts, phylo_homology *buf,
							size_t cap, size_t *total)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !counts || !total) return c->fail("phylo_export_homologies: bad arguments");
	if (ensure_host_lists(c, q_begin, q_end)) return 1;
	size_t tot = 0;
	for (size_t j = q_begin; j < q_end; j++) {
		counts[j - q_begin] = c->homs[j].size();
		tot += c->homs[j].size();
	}
	*total = tot;
	if (buf && cap >= tot) {
		size_t o = 0;
		for (size_t j = q_begin; j < q_end; j++) {
			std::copy(c->homs[j].begin(), c->homs[j].end(), buf + o);
			o += c->homs[j].size();
		}
	}
	return 0;
}

int phylo_import_homologies(phylo_ctx *c, size_t q_begin, size_t q_end, const uint64_t *counts,
							const phylo_homology *buf)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !counts) return c->fail("phylo_import_homologies: bad arguments");
	if (!c->host_stale.empty()) {
		if (ensure_host_lists(c, 0, q_begin) || ensure_host_lists(c, q_end, c->n)) return 1;
		c->host_stale.clear();
	}
	c->homs_staged = false;
	size_t o = 0;
	for (size_t j = q_begin; j < q_end; j++) {
		if (counts[j - q_begin] && !buf) return c->fail("phylo_import_homologies: null buffer");
		c->homs[j].assign(buf + o, buf + o + counts[j - q_begin]);
		o += counts[j - q_begin];
	}
	return 0;
}

int phylo_export_packed(phylo_ctx *c, size_t q_begin, size_t q_end, uint64_t *counts, phylo_packed_homology *buf,
						size_t cap, size_t *total)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !counts || !total) return c->fail("phylo_export_packed: bad arguments");
	if (ensure_host_lists(c, q_begin, q_end)) return 1;
	size_t tot = 0;
	for (size_t j = q_begin; j < q_end; j++) {
		counts[j - q_begin] = c->homs[j].size();
		tot += c->homs[j].size();
	}
	*total = tot;
	if (buf && cap >= tot) {
		size_t o = 0;
		for (size_t j = q_begin; j < q_end; j++)
			for (const phylo_homology &h : c->homs[j])
				buf[o++] = phylo_packed_homology{(uint32_t)h.index_reference_projected, (uint32_t)h.index_query,
												 (uint32_t)h.length, (uint32_t)h.direction};
	}
	return 0;
}

// packed records → host lists of genomes [q_begin, q_end)
static int unpack_lists(phylo_ctx *c, size_t q_begin, size_t q_end, const uint64_t *counts,
						const phylo_packed_homology *buf)
{
	const uint64_t L = c->L;
	size_t o = 0;
	for (size_t j = q_begin; j < q_end; j++) {
		size_t m = counts[j - q_begin];
		if (m && !buf) return c->fail("phylo_import_packed: null buffer");
		c->homs[j].resize(m);
		for (size_t t = 0; t < m; t++, o++) {
			phylo_homology h;
			h.index_reference_projected = buf[o].start;
			h.index_query = buf[o].index_query;
			h.length = buf[o].length;
			h.direction = (int32_t)buf[o].direction;
			h._pad = 0;
			// inverse of homology::reverseEh (src/process.h:72-80)
			h.index_reference = h.direction ? 2 * L + 1 - h.length - h.index_reference_projected : h.index_reference_projected;
			c->homs[j][t] = h;
		}
	}
	return 0;
}

int phylo_import_packed(phylo_ctx *c, size_t q_begin, size_t q_end, const uint64_t *counts,
						const phylo_packed_homology *buf)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !counts) return c->fail("phylo_import_packed: bad arguments");
	if (!c->have_ref) return c->fail("phylo_import_packed: no reference set");
	if (!c->host_stale.empty()) {
		for (size_t g = q_begin; g < q_end; g++) c->host_stale[g] = 0; // replaced below
		if (ensure_host_lists(c, 0, q_begin) || ensure_host_lists(c, q_end, c->n)) return 1;
		c->host_stale.clear();
	}
	c->homs_staged = false;
	return unpack_lists(c, q_begin, q_end, counts, buf);
}

static_assert(sizeof(DevHom) == sizeof(phylo_packed_homology), "the wire record is the device record");

// lists that are device-resident already → back to back in query order: one block per genome
__global__ void gather_lists_kernel(const DevHom *__restrict__ src, const uint64_t *__restrict__ desc, DevHom *__restrict__ dst)
{
	const uint64_t begin = desc[3 * blockIdx.x], count = desc[3 * blockIdx.x + 1], to = desc[3 * blockIdx.x + 2];
	for (uint64_t t = threadIdx.x; t < count; t += blockDim.x) dst[to + t] = src[begin + t];
}

int phylo_export_packed_device(phylo_ctx *c, size_t q_begin, size_t q_end, void *dev_dst, size_t cap, uint64_t *counts,
							   size_t *total)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !counts || !total) return c->fail("phylo_export_packed_device: bad arguments");
	HIPOK(c, hipSetDevice(c->device));
	// phase A may have left these lists on the device only (device sort + filter): copy from there
	bool on_device = !c->host_stale.empty() && c->att_homs && q_end > q_begin;
	for (size_t j = q_begin; j < q_end && on_device; j++) on_device = c->host_stale[j] != 0;
	if (on_device && fetch_att_ranges(c)) return 1;
	if (on_device) {
		const size_t m = q_end - q_begin;
		size_t tot = 0;
		for (size_t j = q_begin; j < q_end; j++) {
			counts[j - q_begin] = c->att_count[j];
			tot += c->att_count[j];
		}
		*total = tot;
		if (!dev_dst || cap < tot || !tot) return 0;
		HIPOK(c, c->h_mat.ensure(3 * m + 8));
		HIPOK(c, c->b_subst.ensure(3 * m + 8)); // scratch for the descriptors (the tallies are not live between phases)
		uint64_t *desc = c->h_mat.p, to = 0;
		for (size_t j = q_begin; j < q_end; j++) {
			desc[3 * (j - q_begin)] = c->att_begin[j];
			desc[3 * (j - q_begin) + 1] = c->att_count[j];
			desc[3 * (j - q_begin) + 2] = to;
			to += c->att_count[j];
		}
		HIPOK(c, hipMemcpyAsync(c->b_subst.p, desc, 3 * m * 8, hipMemcpyHostToDevice, c->stream));
		hipLaunchKernelGGL(gather_lists_kernel, dim3((uint32_t)m), dim3(256), 0, c->stream, c->att_homs,
						   (const uint64_t *)c->b_subst.p, (DevHom *)dev_dst);
		HIPOK(c, hipGetLastError());
		return sync_stream(c);
	}
	if (ensure_host_lists(c, q_begin, q_end)) return 1;
	size_t tot = 0;
	for (size_t j = q_begin; j < q_end; j++) {
		counts[j - q_begin] = c->homs[j].size();
		tot += c->homs[j].size();
	}
	*total = tot;
	if (!dev_dst || cap < tot) return 0; // sizing call
	if (!tot) return 0;
	HIPOK(c, c->h_devhom.ensure(tot + 1));
	DevHom *dh = c->h_devhom.p;
	std::vector<size_t> off(q_end - q_begin + 1, 0);
	for (size_t j = q_begin; j < q_end; j++) off[j - q_begin + 1] = off[j - q_begin] + c->homs[j].size();
	workers(c).run(q_end - q_begin, [&](size_t t) {
		size_t o = off[t];
		for (const phylo_homology &h : c->homs[q_begin + t])
			dh[o++] = DevHom{(uint32_t)h.index_reference_projected, (uint32_t)h.index_query, (uint32_t)h.length,
							 (uint32_t)h.direction};
	});
	HIPOK(c, hipMemcpyAsync(dev_dst, dh, tot * sizeof(DevHom), hipMemcpyHostToDevice, c->stream));
	return sync_stream(c);
}

int phylo_attach_packed_device(phylo_ctx *c, const void *dev_records, const uint64_t *begin, const uint64_t *count,
							   size_t keep_begin, size_t keep_end)
{
	if (!c) return 1;
	if (!begin || !count || keep_begin > keep_end || keep_end > c->n) return c->fail("phylo_attach_packed_device: bad arguments");
	if (!c->have_ref) return c->fail("phylo_attach_packed_device: no reference set");
	HIPOK(c, hipSetDevice(c->device));
	const size_t N = c->n;
	HIPOK(c, c->h_rng.ensure(2 * N));
	HIPOK(c, c->b_hom_rng.ensure(2 * N));
	uint64_t total = 0;
	for (size_t g = 0; g < N; g++) {
		if (begin[g] + count[g] > 0xffffffffull) return c->fail("phylo_attach_packed_device: more than 2^32 records");
		c->h_rng.p[2 * g] = (uint32_t)begin[g];
		c->h_rng.p[2 * g + 1] = (uint32_t)(begin[g] + count[g]);
		total += count[g];
	}
	if (total && !dev_records) return c->fail("phylo_attach_packed_device: null records");
	HIPOK(c, hipMemcpyAsync(c->b_hom_rng.p, c->h_rng.p, 2 * N * 4, hipMemcpyHostToDevice, c->stream));
	// the pileup needs every list sorted by projected start, disjoint and inside the reference (what phase A's
	// filter leaves): a buffer that is anything else is refused here rather than tallied wrongly later
	HIPOK(c, c->b_flag.ensure(4));
	HIPOK(c, hipMemsetAsync(c->b_flag.p + 1, 0, 4, c->stream));
	launch_check_lists((const DevHom *)dev_records, c->b_hom_rng.p, (uint32_t)N, c->L, c->b_flag.p + 1, c->stream);
	uint32_t bad_lists = 0;
	HIPOK(c, hipMemcpyAsync(&bad_lists, c->b_flag.p + 1, 4, hipMemcpyDeviceToHost, c->stream));
	if (sync_stream(c)) return 1;
	if (bad_lists) return c->fail("phylo_attach_packed_device: a genome's list is not sorted by projected start, disjoint and inside the reference");
	c->att_homs = (const DevHom *)dev_records;
	c->att_rng_on_device = false;
	c->att_begin.assign(begin, begin + N);
	c->att_count.assign(count, count + N);
	// lists of the kept range that this context has only on the device so far (device sort +
	// filter) stay to be fetched — from the new buffer, which holds them too
	std::vector<uint8_t> was = c->host_stale;
	c->host_stale.assign(N, 1);
	for (size_t g = keep_begin; g < keep_end; g++) c->host_stale[g] = was.size() == N ? was[g] : 0;
	c->homs_staged = true;
	c->eager_valid = false;
	return 0;
}

// ── the exchange between ranks without the host in it ──
// A rank's block: 4 header words {records in the block, overflow, queries, 0}, max_queries list lengths, then cap
// records of 16 bytes.  Every rank's block has the same size, so one all-gather assembles all of them; the receiving
// side works the per-genome ranges out on the device.
static const uint32_t XB_HDR = 4;
__global__ __launch_bounds__(256) void block_export_kernel(const DevHom *__restrict__ src, const uint32_t *__restrict__ rng, uint32_t nq,
															 uint32_t maxq, uint32_t cap, uint32_t *__restrict__ block)
{
	const uint32_t j = blockIdx.x; // a slot of the header: the block's query j, or padding
	__shared__ uint32_t part[256];
	uint32_t acc = 0;
	for (uint32_t t = threadIdx.x; t < j && t < nq; t += blockDim.x) acc += rng[2 * t + 1] - rng[2 * t];
	part[threadIdx.x] = acc;
	__syncthreads();
	for (uint32_t s2 = 128; s2 > 0; s2 >>= 1) {
		if (threadIdx.x < s2) part[threadIdx.x] += part[threadIdx.x + s2];
		__syncthreads();
	}
	const uint32_t off = part[0];
	const uint32_t cnt = j < nq ? rng[2 * j + 1] - rng[2 * j] : 0u;
	if (threadIdx.x == 0) {
		block[XB_HDR + j] = cnt;
		if (j + 1 == maxq) {
			block[0] = off + cnt;
			block[2] = nq;
			block[3] = 0;
		}
		if (off + cnt > cap) block[1] = 1; // (cleared by the caller's memset of the header)
	}
	if (off + cnt > cap) return;
	DevHom *dst = (DevHom *)(block + XB_HDR + maxq) + off;
	const DevHom *from = src + (j < nq ? rng[2 * j] : 0u);
	for (uint32_t t = threadIdx.x; t < cnt; t += blockDim.x) dst[t] = from[t];
}
// one thread block per rank: the ranges of its genomes in the gathered buffer (in records from the buffer's start)
__global__ __launch_bounds__(256) void block_attach_kernel(const uint32_t *__restrict__ all, const uint32_t *__restrict__ bounds,
															 uint32_t maxq, uint32_t cap, uint32_t *__restrict__ rng, uint32_t *__restrict__ flags)
{
	const uint32_t r = blockIdx.x;
	const uint32_t words = XB_HDR + maxq + 4u * cap;
	const uint32_t *blk = all + (size_t)r * words;
	const uint32_t g0 = bounds[r], nq = bounds[r + 1] - g0;
	// an overflowed or mismatched block: reported (flags[2]), and its genomes' lists are left empty — the lengths in
	// its header reach beyond the records it holds, and the kernels that follow must not read there
	const bool usable = !(blk[1] || blk[2] != nq || blk[0] > cap);
	if (threadIdx.x == 0 && !usable) flags[2] = 1;
	__shared__ uint32_t carry;
	__shared__ uint32_t scan[256];
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	const uint32_t base = (uint32_t)(((size_t)r * words + XB_HDR + maxq) / 4u);
	for (uint32_t t0 = 0; t0 < nq; t0 += 256) {
		const uint32_t t = t0 + threadIdx.x;
		const uint32_t cnt = (usable && t < nq) ? blk[XB_HDR + t] : 0u;
		scan[threadIdx.x] = cnt;
		__syncthreads();
		for (uint32_t d = 1; d < 256; d <<= 1) {
			const uint32_t v = threadIdx.x >= d ? scan[threadIdx.x - d] : 0u;
			__syncthreads();
			scan[threadIdx.x] += v;
			__syncthreads();
		}
		const uint32_t begin = carry + scan[threadIdx.x] - cnt;
		if (t < nq) {
			const bool inside = begin + cnt <= cap; // (lengths that do not add up to the header's total)
			if (!inside) flags[2] = 1;
			rng[2 * (g0 + t)] = base + (inside ? begin : 0u);
			rng[2 * (g0 + t) + 1] = base + (inside ? begin + cnt : 0u);
		}
		__syncthreads();
		if (threadIdx.x == 255) carry += scan[255];
		__syncthreads();
	}
}

size_t phylo_exchange_block_bytes(size_t max_queries, size_t cap_records) { return (XB_HDR + max_queries + 4 * cap_records) * 4; }

int phylo_export_block_device(phylo_ctx *c, size_t q_begin, size_t q_end, void *dev_block, size_t max_queries, size_t cap_records)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !dev_block) return c->fail("phylo_export_block_device: bad arguments");
	const size_t nq = q_end - q_begin;
	if (max_queries < nq || max_queries % 4 || max_queries == 0) return c->fail("phylo_export_block_device: max_queries must be a multiple of 4 and hold the block's queries");
	if (XB_HDR + max_queries + 4 * cap_records >= 0xffffffffull) return c->fail("phylo_export_block_device: block too large");
	// the lists must be where phase A's device filter left them: this context's buffer, ranges by local query index
	bool on_device = !c->host_stale.empty() && c->att_homs == c->b_homs.p && c->plan_valid && c->plan_qb == q_begin && c->plan_qe == q_end;
	for (size_t j = q_begin; j < q_end && on_device; j++) on_device = c->host_stale[j] != 0;
	HIPOK(c, hipSetDevice(c->device));
	if (!on_device) {
		// the lists live on the host (a query with tied starts went through std::sort, the host filter was asked for,
		// lists were installed by the caller): the block is put together here and uploaded — the other ranks' blocks
		// do not care how this one was made
		if (ensure_host_lists(c, q_begin, q_end)) return 1;
		size_t tot = 0;
		for (size_t j = q_begin; j < q_end; j++) tot += c->homs[j].size();
		const bool over = tot > cap_records;
		const size_t words = XB_HDR + max_queries + (over ? 0 : 4 * tot);
		HIPOK(c, c->h_devhom.ensure(words / 4 + 2));
		uint32_t *blk = (uint32_t *)c->h_devhom.p;
		blk[0] = (uint32_t)tot;
		blk[1] = over ? 1u : 0u;
		blk[2] = (uint32_t)nq;
		blk[3] = 0;
		for (size_t t = 0; t < max_queries; t++) blk[XB_HDR + t] = t < nq ? (uint32_t)c->homs[q_begin + t].size() : 0u;
		if (!over) {
			DevHom *rec = (DevHom *)(blk + XB_HDR + max_queries);
			for (size_t j = q_begin; j < q_end; j++)
				for (const phylo_homology &h : c->homs[j])
					*rec++ = DevHom{(uint32_t)h.index_reference_projected, (uint32_t)h.index_query, (uint32_t)h.length, (uint32_t)h.direction};
		}
		HIPOK(c, hipMemcpyAsync(dev_block, blk, words * 4, hipMemcpyHostToDevice, c->stream));
		return sync_stream(c); // (the staging buffer is reused by other calls)
	}
	HIPOK(c, hipMemsetAsync(dev_block, 0, XB_HDR * 4, c->stream));
	hipLaunchKernelGGL(block_export_kernel, dim3((uint32_t)max_queries), dim3(256), 0, c->stream, (const DevHom *)c->b_homs.p,
					   (const uint32_t *)c->b_hom_rng.p, (uint32_t)nq, (uint32_t)max_queries, (uint32_t)cap_records, (uint32_t *)dev_block);
	HIPOK(c, hipGetLastError());
	return 0;
}

static int fetch_att_ranges(phylo_ctx *c)
{
	if (!c->att_rng_on_device) return 0;
	const size_t N = c->n;
	std::vector<uint32_t> r(2 * N);
	uint32_t fl[2] = {0, 0};
	HIPOK(c, hipSetDevice(c->device));
	HIPOK(c, hipMemcpyAsync(r.data(), c->b_hom_rng.p, 2 * N * 4, hipMemcpyDeviceToHost, c->stream));
	if (c->att_unchecked) HIPOK(c, hipMemcpyAsync(fl, c->b_flag.p + 1, 8, hipMemcpyDeviceToHost, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	if (c->att_unchecked && (fl[0] || fl[1])) return c->fail("the lists gathered from the ranks are not usable (overflowed block or unsorted list)");
	c->att_begin.assign(N, 0);
	c->att_count.assign(N, 0);
	for (size_t g = 0; g < N; g++) {
		c->att_begin[g] = r[2 * g];
		c->att_count[g] = r[2 * g + 1] - r[2 * g];
	}
	c->att_rng_on_device = false;
	return 0;
}

int phylo_attach_blocks_device(phylo_ctx *c, const void *dev_all, size_t world, const size_t *bounds, size_t max_queries,
							   size_t cap_records, size_t keep_begin, size_t keep_end)
{
	if (!c) return 1;
	if (!dev_all || !world || !bounds || keep_begin > keep_end || keep_end > c->n) return c->fail("phylo_attach_blocks_device: bad arguments");
	if (!c->have_ref) return c->fail("phylo_attach_blocks_device: no reference set");
	if (bounds[0] != 0 || bounds[world] != c->n) return c->fail("phylo_attach_blocks_device: the blocks must cover all genomes");
	const size_t words = XB_HDR + max_queries + 4 * cap_records;
	if (max_queries % 4 || world * words >= 0xffffffffull) return c->fail("phylo_attach_blocks_device: bad block shape");
	for (size_t r = 0; r < world; r++)
		if (bounds[r + 1] < bounds[r] || bounds[r + 1] - bounds[r] > max_queries) return c->fail("phylo_attach_blocks_device: bad bounds");
	HIPOK(c, hipSetDevice(c->device));
	const size_t N = c->n;
	HIPOK(c, c->b_hom_rng.ensure(2 * N + world + 2));
	HIPOK(c, c->b_flag.ensure(4));
	HIPOK(c, c->h_rng.ensure(3 * N + world + 16));
	uint32_t *hb = c->h_rng.p; // pinned: the copy below must not wait for pageable staging
	for (size_t r = 0; r <= world; r++) hb[r] = (uint32_t)bounds[r];
	uint32_t *d_bounds = c->b_hom_rng.p + 2 * N;
	HIPOK(c, hipMemcpyAsync(d_bounds, hb, (world + 1) * 4, hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipMemsetAsync(c->b_flag.p + 1, 0, 8, c->stream));
	hipLaunchKernelGGL(block_attach_kernel, dim3((uint32_t)world), dim3(256), 0, c->stream, (const uint32_t *)dev_all, d_bounds,
					   (uint32_t)max_queries, (uint32_t)cap_records, c->b_hom_rng.p, c->b_flag.p);
	launch_check_lists((const DevHom *)dev_all, c->b_hom_rng.p, (uint32_t)N, c->L, c->b_flag.p + 1, c->stream);
	HIPOK(c, hipGetLastError());
	// nothing is waited for: b_flag[1] (a list that is not sorted and disjoint) and b_flag[2] (a block that overflowed
	// its capacity) are read with the result of the comparison that follows
	c->att_homs = (const DevHom *)dev_all;
	c->att_rng_on_device = true;
	c->att_unchecked = true;
	std::vector<uint8_t> was = c->host_stale;
	c->host_stale.assign(N, 1);
	for (size_t g = keep_begin; g < keep_end; g++) c->host_stale[g] = was.size() == N ? was[g] : 0;
	c->homs_staged = true;
	c->eager_valid = false;
	return 0;
}

int phylo_complete_delete(phylo_ctx *c)
{
	if (!c) return 1;
	if (ensure_host_lists(c, 0, c->n)) return 1;
	c->host_stale.clear();
	c->homs = complete_delete(c->homs);
	c->homs_staged = false;
	return 0;
}

// ───────────────────────── phase B ─────────────────────────

// compare(list, list) of process.cxx:566-611 as a segment generator (host):
// every overlapping (ha, hb) pair becomes one seqcmp / revseqcmp segment
// (process.cxx:620-658).  Used by the segment-list backend.
static void pair_segments(const phylo_ctx *c, size_t i, size_t j, std::vector<Segment> &out)
{
	const auto &ha = c->homs[i], &hb = c->homs[j];
	size_t right = 0;
	for (const phylo_homology &h : ha) {
		uint64_t hs = h.index_reference_projected, he = hs + h.length;
		while (right < hb.size() && hb[right].index_reference_projected + hb[right].length <= hs) right++;
		for (size_t r = right; r < hb.size(); r++) {
			const phylo_homology &o = hb[r];
			uint64_t os = o.index_reference_projected, oe = os + o.length;
			if (os >= he) break;
			uint64_t cs = std::max(hs, os), ce = std::min(he, oe);
			if (cs >= ce) continue;
			phylo_homology hat = trim_homology(h, cs, ce), hbt = trim_homology(o, cs, ce);
			Segment sg;
			sg.len = (uint32_t)(ce - cs);
			if (h.direction == o.direction) {
				sg.a = c->goff[i] + hat.index_query;
				sg.b = c->goff[j] + hbt.index_query;
				sg.rev = 0;
			} else if (o.direction == 1) { // account_rev(sa + hat.start_query(), sb, hbt.end_query(), n)
				sg.a = c->goff[i] + hat.index_query;
				sg.b = c->goff[j] + hbt.index_query + hbt.length - sg.len;
				sg.rev = 1;
			} else {
				sg.a = c->goff[j] + hbt.index_query;
				sg.b = c->goff[i] + hat.index_query + hat.length - sg.len;
				sg.rev = 1;
			}
			out.push_back(sg);
		}
	}
}

static int compare_segments(phylo_ctx *c, size_t part, size_t nparts, uint64_t *subst, uint64_t *homologs)
{
	if (ensure_host_lists(c, 0, c->n)) return 1;
	size_t N = c->n;
	std::vector<Segment> segs;
	std::vector<uint32_t> seg_pair; // pair index of each segment
	std::vector<std::pair<uint32_t, uint32_t>> pairs;
	size_t pid = 0;
	for (size_t i = 0; i < N; i++)
		for (size_t j = i + 1; j < N; j++, pid++) {
			if (pid % nparts != part) continue;
			size_t before = segs.size();
			pair_segments(c, i, j, segs);
			pairs.emplace_back((uint32_t)i, (uint32_t)j);
			seg_pair.resize(segs.size(), (uint32_t)(pairs.size() - 1));
			(void)before;
		}
	std::vector<uint64_t> out(segs.size());
	if (!segs.empty()) {
		HIPOK(c, c->s_segs.ensure(segs.size()));
		HIPOK(c, c->s_out.ensure(segs.size()));
		HIPOK(c, hipMemcpyAsync(c->s_segs.p, segs.data(), segs.size() * sizeof(Segment), hipMemcpyHostToDevice, c->stream));
		int blocks = std::min<int>(c->n_cu * 8, (int)((segs.size() + 3) / 4));
		{
			KernelSpan s(c, "seqcmp_batch");
			launch_seqcmp_batch(c->d_genomes, c->s_segs.p, (uint32_t)segs.size(), c->s_out.p, blocks, c->stream);
		}
		HIPOK(c, hipGetLastError());
		HIPOK(c, hipMemcpyAsync(out.data(), c->s_out.p, segs.size() * 8, hipMemcpyDeviceToHost, c->stream));
		if (sync_stream(c)) return 1;
	}
	double sites = 0;
	for (size_t s = 0; s < segs.size(); s++) {
		auto pr = pairs[seg_pair[s]];
		size_t a = (size_t)pr.first * N + pr.second, b = (size_t)pr.second * N + pr.first;
		subst[a] += out[s];
		homologs[a] += segs[s].len;
		subst[b] = subst[a];
		homologs[b] = homologs[a];
		sites += segs[s].len;
	}
	c->stats["count:compare_sites"] += sites;
	c->stats["count:segments"] += (double)segs.size();
	return 0;
}

// dev_out: leave the tallies in the caller's device buffers (subst / homologs are device pointers)
// The result on its way to the host: both matrices as symmetric u32 (a tally is at most the reference's length,
// below 2^31) — half the bytes of the u64 matrices across PCIe; the host widens them row by row, which is a streaming
// pass.  (Mirroring on the host instead is a strided walk over 16 MB: measured 0.5-2 ms at N = 1024)
__global__ __launch_bounds__(256) void sym32_from_matrices_kernel(uint32_t N, const unsigned long long *__restrict__ s,
																   const unsigned long long *__restrict__ h, uint32_t *__restrict__ out)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, NN = (uint64_t)N * N;
	if (t >= NN) return;
	const uint32_t i = (uint32_t)(t / N), j = (uint32_t)(t % N);
	const uint64_t src = i < j ? t : (uint64_t)j * N + i;
	out[t] = i == j ? 0u : (uint32_t)s[src];
	out[NN + t] = i == j ? 0u : (uint32_t)h[src];
}
__global__ __launch_bounds__(256) void sym32_from_triangle_kernel(uint32_t N, const uint32_t *__restrict__ tri, uint32_t *__restrict__ out)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, NN = (uint64_t)N * N;
	if (t >= NN) return;
	const uint32_t i = (uint32_t)(t / N), j = (uint32_t)(t % N);
	const uint32_t a = i < j ? i : j, b = i < j ? j : i;
	const uint64_t P = (uint64_t)N * (N - 1) / 2, k = (uint64_t)a * (2ull * N - a - 1) / 2 + (b - a - 1);
	out[t] = i == j ? 0u : tri[k];
	out[NN + t] = i == j ? 0u : tri[P + k];
}
// sym (2 N^2 u32 in pinned memory) -> the caller's two N x N u64 matrices; returns the sum of the homologs matrix
static double widen_result(phylo_ctx *c, const uint32_t *sym, uint64_t *subst, uint64_t *homologs)
{
	const size_t N = c->n, NN = N * N, parts = NN >= ((size_t)1 << 18) ? 32 : 1;
	std::vector<double> part_sites(parts, 0.0);
	auto widen = [&](size_t t) {
		const size_t a = NN * t / parts, b = NN * (t + 1) / parts;
		for (size_t k = a; k < b; k++) subst[k] = sym[k];
		uint64_t acc = 0;
		for (size_t k = a; k < b; k++) {
			const uint32_t v = sym[NN + k];
			homologs[k] = v;
			acc += v;
		}
		part_sites[t] = (double)acc;
	};
	if (parts > 1) workers(c).run(parts, widen);
	else widen(0);
	double sites = 0;
	for (double v : part_sites) sites += v;
	return sites;
}

// u32 upper triangle: tri[k] = substitutions, tri[P + k] = homologs of pair (i < j), k = i (2N - i - 1) / 2 + (j - i - 1):
// what crosses the wire between ranks (a tally is at most the reference's length, which is below 2^31)
__global__ __launch_bounds__(256) void pack_triangle_kernel(uint32_t N, const unsigned long long *__restrict__ s,
															 const unsigned long long *__restrict__ h, uint32_t *__restrict__ tri)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= (uint64_t)N * N) return;
	const uint32_t i = (uint32_t)(t / N), j = (uint32_t)(t % N);
	if (i >= j) return;
	const uint64_t P = (uint64_t)N * (N - 1) / 2, k = (uint64_t)i * (2ull * N - i - 1) / 2 + (j - i - 1);
	tri[k] = (uint32_t)s[t];
	tri[P + k] = (uint32_t)h[t];
}

// out_mode 0: the caller's host matrices; 1: the caller's device matrices (subst / homologs are device pointers);
// 2: the caller's device u32 triangle (subst is the device pointer, homologs unused)
static int compare_pileup(phylo_ctx *c, size_t part, size_t nparts, uint64_t *subst, uint64_t *homologs, int out_mode = 0)
{
	const bool dev_out = out_mode != 0;
	size_t N = c->n;
	hipStream_t st = c->stream;
	Pileup P;
	if (make_pileup(c, part, nparts, &P)) return 1;

	// filtered homologies → device, unless phase A staged them there already
	double t0 = now_ms();
	if (!c->homs_staged) {
		if (ensure_host_lists(c, 0, N)) return 1;
		c->host_stale.clear();
		c->att_homs = nullptr;
		c->att_rng_on_device = false;
		std::vector<uint32_t> hom_rng(2 * N);
		size_t tot = 0;
		for (size_t g = 0; g < N; g++) {
			hom_rng[2 * g] = (uint32_t)tot;
			tot += c->homs[g].size();
			hom_rng[2 * g + 1] = (uint32_t)tot;
		}
		HIPOK(c, c->h_devhom.ensure(tot + 1));
		DevHom *dh = c->h_devhom.p;
		std::atomic<size_t> bad{(size_t)-1};
		std::atomic<bool> entangled{false};
		workers(c).run(N, [&](size_t g) {
			size_t o = hom_rng[2 * g];
			uint64_t prev_end = 0;
			for (const phylo_homology &h : c->homs[g]) {
				if (h.index_reference_projected + h.length > c->L) bad = g;
				if (h.index_reference_projected < prev_end) entangled = true;
				prev_end = h.index_reference_projected + h.length;
				dh[o++] = DevHom{(uint32_t)h.index_reference_projected, (uint32_t)h.index_query, (uint32_t)h.length,
								 (uint32_t)h.direction};
			}
		});
		if (bad != (size_t)-1) return c->fail("genome %zu: homology reaches beyond the reference", bad.load());
		if (entangled) {
			// Lists installed by the caller (phylo_set_homologies, phylo_import_*) that are not sorted and disjoint on
			// the reference: the pileup would not be compare(list, list) of process.cxx:566-611 for them — the segment
			// backend, which restates that merge-join literally, takes the call.
			c->stats["count:compare_calls_rerouted_to_segments"] += 1;
			if (dev_out) {
				std::vector<uint64_t> hs(N * N, 0), hh(N * N, 0);
				if (compare_segments(c, part, nparts, hs.data(), hh.data())) return 1;
				HIPOK(c, hipMemcpy(subst, hs.data(), N * N * 8, hipMemcpyHostToDevice));
				HIPOK(c, hipMemcpy(homologs, hh.data(), N * N * 8, hipMemcpyHostToDevice));
				return 0;
			}
			std::fill(subst, subst + N * N, 0);
			std::fill(homologs, homologs + N * N, 0);
			return compare_segments(c, part, nparts, subst, homologs);
		}
		HIPOK(c, c->b_hom_rng.ensure(2 * N));
		HIPOK(c, c->b_homs.ensure(tot + 1));
		HIPOK(c, hipMemcpyAsync(c->b_hom_rng.p, hom_rng.data(), 2 * N * 4, hipMemcpyHostToDevice, st));
		if (tot) HIPOK(c, hipMemcpyAsync(c->b_homs.p, dh, tot * sizeof(DevHom), hipMemcpyHostToDevice, st));
		if (sync_stream(c)) return 1; // hom_rng goes out of scope
		c->stats["ms:compare_hom_upload"] += now_ms() - t0;
	}
	HIPOK(c, c->b_flag.ensure(4));
	HIPOK(c, c->b_first.ensure(project_index_entries(P) + 1));
	unsigned long long *acc_s, *acc_h; // where the pair kernel accumulates
	if (out_mode == 1) {
		acc_s = (unsigned long long *)subst;
		acc_h = (unsigned long long *)homologs;
	} else {
		HIPOK(c, c->b_subst.ensure(N * N));
		HIPOK(c, c->b_homologs.ensure(N * N));
		acc_s = c->b_subst.p;
		acc_h = c->b_homologs.p;
	}
	HIPOK(c, c->h_mat.ensure(2 * N * N + 8));
	if (out_mode == 0) HIPOK(c, c->b_sym32.ensure(2 * N * N + 4));
	HIPOK(c, hipMemsetAsync(acc_s, 0, N * N * 8, st));
	HIPOK(c, hipMemsetAsync(acc_h, 0, N * N * 8, st));
	const DevHom *dev_homs = c->att_homs ? c->att_homs : c->b_homs.p;
	// phase A may have projected the lists already (whole reference, i.e. part 0 of 1)
	const bool projected = c->homs_staged && c->eager_valid && part == 0 && nparts == 1;
	HIPOK(c, c->b_bang.ensure(2 * (size_t)c->bang_cap + 2));
	if (!projected) {
		HIPOK(c, hipMemsetAsync(c->b_flag.p, 0, 4, st)); // (words 1 and 2 belong to phylo_attach_blocks_device)
		HIPOK(c, hipMemsetAsync(c->b_flag.p + 3, 0, 4, st)); // the count of listed '!'
		launch_tile_index(P, query_src(c), dev_homs, c->b_hom_rng.p, c->b_first.p, 0, (uint32_t)N, st);
	}
	// Three planes and the plain pair kernel unless '!' turns up among the projected positions (the projection raises
	// a flag); then all five planes and the kernel that reads D and B.  The context remembers the outcome of its last
	// call and goes ahead on that assumption — projection, pairs, mirror image and the copy back are queued without
	// waiting for the flag, which is read with the result; only when it says '!' and the plain kernel ran is the
	// work repeated with five planes (once per set of genomes: the next call expects it).  The other way round —
	// five planes and no '!' after all — the result is right as it stands, since B is empty.
	uint32_t *flagp = (uint32_t *)(c->h_mat.p + 2 * N * N);
	// pair tiles (ig, jt) holding at least one pair i<j
	std::vector<uint32_t> tiles;
	uint32_t nig = (uint32_t)((N + PAIR_IG - 1) / PAIR_IG), njt = (uint32_t)((N + PAIR_JT - 1) / PAIR_JT);
	for (uint32_t ig = 0; ig < nig; ig++)
		for (uint32_t jt = 0; jt < njt; jt++) {
			if ((uint64_t)ig * PAIR_IG >= (uint64_t)jt * PAIR_JT + PAIR_JT - 1) continue; // no i<j inside
			tiles.push_back((ig << 16) | jt);
		}
	if (!tiles.empty()) {
		HIPOK(c, c->b_tiles.ensure(tiles.size() + (N / 64 + 2) * (N / 64 + 2))); // room for the matrix-core kernel's tiles behind them
		HIPOK(c, hipMemcpyAsync(c->b_tiles.p, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice, st));
	}
	// tiles of the matrix-core kernel: 64 x 64 genomes, ti <= tj
	std::vector<uint32_t> mtiles;
	if (c->opt_pairs_kernel == 0) {
		const uint32_t T = pairs_mfma_tile(), nt = (uint32_t)((N + T - 1) / T);
		for (uint32_t a = 0; a < nt; a++)
			for (uint32_t b = a; b < nt; b++) mtiles.push_back((a << 16) | b);
		HIPOK(c, c->b_tiles.ensure(tiles.size() + mtiles.size()));
		if (!mtiles.empty()) HIPOK(c, hipMemcpyAsync(c->b_tiles.p + tiles.size(), mtiles.data(), mtiles.size() * 4, hipMemcpyHostToDevice, st));
	}
	bool do_correct = false; // three planes under the matrix-core kernel: the listed '!' are settled before the tallies leave
	auto finish_tallies = [&]() { // the packed triangle for the wire; mirror images for the matrices (u32 on the way to the host)
		if (do_correct && c->bang_cap) {
			KernelSpan s(c, "pileup_bang_correct");
			launch_bang_correct(P, query_src(c), dev_homs, c->b_hom_rng.p, c->b_bang.p, c->b_flag.p + 3, c->bang_cap, acc_s, st);
		}
		if (out_mode == 2)
			hipLaunchKernelGGL(pack_triangle_kernel, dim3((uint32_t)((N * N + 255) / 256)), dim3(256), 0, st, (uint32_t)N, acc_s, acc_h, (uint32_t *)subst);
		else if (out_mode == 1)
			launch_symmetrise((uint32_t)N, acc_s, acc_h, st);
		else
			hipLaunchKernelGGL(sym32_from_matrices_kernel, dim3((uint32_t)((N * N + 255) / 256)), dim3(256), 0, st, (uint32_t)N, acc_s, acc_h, c->b_sym32.p);
	};
	auto pairs = [&](bool bang) -> int {
		if (tiles.empty() || !P.W) {
			finish_tallies();
			return 0;
		}
		if (!bang && !mtiles.empty()) {
			// Without '!' the tallies are a contraction over {-1, 0, 1} channels: the matrix cores take it
			// (pileup_kernels.hip: pairs_mfma_kernel).  Window chunks: the chunk's rows of three planes in an XCD's L2,
			// ~16 wavefronts per CU in all (measured best at N = 256; indifferent at N = 1024), a multiple of 8 chunks
			// (dealt round-robin over the XCDs) of a whole number of steps (2 windows) each.
			const uint32_t row_bytes = 3u * P.Npad * 4u;
			const uint32_t l2_fit = std::max<uint32_t>(64, (3u << 20) / row_bytes);
			const uint32_t want_chunks = std::max<uint32_t>(1, ((uint32_t)c->n_cu * 16u) / (uint32_t)mtiles.size());
			uint32_t wchunk = std::max<uint32_t>(48, (P.W + want_chunks - 1) / want_chunks);
			wchunk = std::min(wchunk, l2_fit);
			const uint32_t groups = (P.W + 8u * wchunk - 1) / (8u * wchunk);
			wchunk = std::max<uint32_t>(2, (P.W + 8u * groups - 1) / (8u * groups));
			wchunk = (wchunk + 1u) & ~1u;
			if (c->opt_pairs_wchunk) wchunk = (c->opt_pairs_wchunk + 1u) & ~1u;
			wchunk = std::min(wchunk, pairs_mfma_max_wchunk() & ~1u);
			{
				KernelSpan s(c, "pileup_pairs_mfma");
				launch_pairs_mfma(P, c->b_tiles.p + tiles.size(), (uint32_t)mtiles.size(), wchunk, acc_s, acc_h, st);
			}
			HIPOK(c, hipGetLastError());
			finish_tallies();
			return 0;
		}
		// window chunks: small enough that a chunk's plane rows (3 or 5 planes x Npad x 4 B
		// per window) fit an XCD's 4 MiB L2 with room to spare, and small enough that
		// tiles x chunks fills the chip several times over; at least 64 windows
		uint32_t row_bytes = (bang ? 5u : 3u) * P.Npad * 4u;
		uint32_t l2_fit = std::max<uint32_t>(64, (3u << 20) / row_bytes);
		// (four rounds of the chip's n_cu x 32 wavefront slots: with one round — what n_cu x 32 gave at N = 256 — the
		// wavefronts all end together and the tail is a whole wavefront long; measured 1.30 -> 1.13 ms on C3)
		uint32_t want_chunks = std::max<uint32_t>(1, ((uint32_t)c->n_cu * 128u) / (uint32_t)tiles.size());
		uint32_t wchunk = std::max<uint32_t>(64, (P.W + want_chunks - 1) / want_chunks);
		wchunk = std::min(wchunk, l2_fit);
		{ // chunks are dealt round-robin over the 8 XCDs: a multiple of 8 of them keeps the XCDs level
			const uint32_t groups = (P.W + 8u * wchunk - 1) / (8u * wchunk);
			wchunk = std::max<uint32_t>(1, (P.W + 8u * groups - 1) / (8u * groups));
		}
		if (c->opt_pairs_wchunk) wchunk = c->opt_pairs_wchunk;
		{
			KernelSpan s(c, bang ? "pileup_pairs_bang" : "pileup_pairs");
			launch_pairs(P, bang, c->b_tiles.p, (uint32_t)tiles.size(), wchunk, acc_s, acc_h, st);
		}
		HIPOK(c, hipGetLastError());
		finish_tallies();
		return 0;
	};
	auto project = [&](bool five) -> int {
		KernelSpan s(c, five ? "pileup_project5" : "pileup_project");
		launch_project(P, five, query_src(c), dev_homs, c->b_hom_rng.p, c->b_first.p, c->b_flag.p, 0, P.Npad / project_genomes_per_tile(), st,
					   c->b_bang.p, c->bang_cap);
		return 0;
	};
	uint64_t *hs = c->h_mat.p;
	auto fetch = [&]() -> int { // the flag, and the result unless it stays on the device
		HIPOK(c, hipGetLastError());
		HIPOK(c, hipMemcpyAsync(flagp, c->b_flag.p, 12, hipMemcpyDeviceToHost, st));
		if (!dev_out) HIPOK(c, hipMemcpyAsync(hs, c->b_sym32.p, 2 * N * N * 4, hipMemcpyDeviceToHost, st));
		return sync_stream(c);
	};
	// The matrix-core path (option "pairs_kernel" = 0) works on three planes whatever the genomes hold: the projection
	// lists the '!' it meets — a handful: contig joins inside homologies — and launch_bang_correct settles them after the
	// pair kernel.  (Only a list beyond its capacity — lists installed by a caller that overlap on the query — falls
	// back to the five planes below.)
	const bool sparse = !mtiles.empty() && !(projected && c->eager_five);
	const bool have_five = sparse ? false : (projected ? c->eager_five : c->pileup_five); // the planes this attempt works on
	bool bang = !sparse && c->pileup_five && have_five;
	if (!projected && project(have_five)) return 1;
	double t1 = now_ms();
	do_correct = sparse;
	if (pairs(bang) || fetch()) return 1;
	const bool att_bad = c->att_unchecked && (flagp[1] || flagp[2]);
	c->att_unchecked = false;
	if (att_bad)
		return c->fail(flagp[2] ? "the lists gathered from the ranks overflowed their blocks' capacity (phylo_attach_blocks_device)"
								: "a gathered list is not sorted by projected start, disjoint and inside the reference");
	uint32_t flag = *flagp;
	if (sparse) flag = (flag & 2u) ? 1u : 0u; // only a '!' list beyond its capacity sends this path to the five planes
	do_correct = false;
	if (flag && !bang) { // '!' among the projected positions, and the plain kernel ran: once more with all five planes
		c->stats["count:compare_repeated_with_five_planes"] += 1;
		HIPOK(c, hipMemsetAsync(acc_s, 0, N * N * 8, st));
		HIPOK(c, hipMemsetAsync(acc_h, 0, N * N * 8, st));
		HIPOK(c, hipMemsetAsync(c->b_flag.p, 0, 4, st));
		bang = true;
		if (project(true) || pairs(true) || fetch()) return 1;
		flag = *flagp;
	}
	if (!sparse) c->pileup_five = flag != 0;
	if (dev_out) {
		c->stats["ms:compare_project_phase"] += t1 - t0;
		c->stats["ms:compare_pairs_phase"] += now_ms() - t1;
		c->stats["pileup:bang"] = flag;
		return 0;
	}
	double t2 = now_ms();
	// out of the pinned buffer into the caller's matrices, widened (16 MB at N = 1024: worth several threads)
	double sites = widen_result(c, (const uint32_t *)hs, subst, homologs);
	sites *= 0.5;
	c->stats["ms:compare_project_phase"] += t1 - t0;
	c->stats["ms:compare_pairs_phase"] += t2 - t1;
	c->stats["ms:compare_symmetrise"] += now_ms() - t2;
	c->stats["count:compare_sites"] += sites;
	c->stats["pileup:bang"] = flag;
	return 0;
}

int phylo_compare(phylo_ctx *c, size_t part, size_t nparts, uint64_t *subst, uint64_t *homologs)
{
	if (!c) return 1;
	if (!subst || !homologs) return c->fail("null output matrix");
	if (nparts == 0 || part >= nparts) return c->fail("bad part %zu of %zu", part, nparts);
	if (!c->have_ref) return c->fail("phylo_compare: no reference set");
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	size_t N = c->n;
	int rc;
	if (c->backend == 1) { // the segment backend adds into the matrices; the pileup one writes every cell
		std::fill(subst, subst + N * N, 0);
		std::fill(homologs, homologs + N * N, 0);
		rc = compare_segments(c, part, nparts, subst, homologs);
	} else {
		rc = compare_pileup(c, part, nparts, subst, homologs);
	}
	c->stats["ms:compare_total"] += now_ms() - t0;
	c->stats["n:compare_calls"] += 1;
	return rc;
}

int phylo_compare_device(phylo_ctx *c, size_t part, size_t nparts, uint64_t *dev_subst, uint64_t *dev_homologs)
{
	if (!c) return 1;
	if (!dev_subst || !dev_homologs) return c->fail("null output matrix");
	if (nparts == 0 || part >= nparts) return c->fail("bad part %zu of %zu", part, nparts);
	if (!c->have_ref) return c->fail("phylo_compare_device: no reference set");
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	int rc;
	if (c->backend == 1) { // the segment backend tallies on the host: copy its result over
		size_t N = c->n;
		std::vector<uint64_t> s(N * N, 0), h(N * N, 0);
		rc = compare_segments(c, part, nparts, s.data(), h.data());
		if (!rc) {
			HIPOK(c, hipMemcpy(dev_subst, s.data(), N * N * 8, hipMemcpyHostToDevice));
			HIPOK(c, hipMemcpy(dev_homologs, h.data(), N * N * 8, hipMemcpyHostToDevice));
		}
	} else {
		rc = compare_pileup(c, part, nparts, dev_subst, dev_homologs, 1);
	}
	c->stats["ms:compare_total"] += now_ms() - t0;
	c->stats["n:compare_calls"] += 1;
	return rc;
}

int phylo_compare_triangle_device(phylo_ctx *c, size_t part, size_t nparts, uint32_t *dev_tri)
{
	if (!c) return 1;
	if (!dev_tri) return c->fail("null output triangle");
	if (nparts == 0 || part >= nparts) return c->fail("bad part %zu of %zu", part, nparts);
	if (!c->have_ref) return c->fail("phylo_compare_triangle_device: no reference set");
	if (c->backend == 1) return c->fail("phylo_compare_triangle_device: the segment backend tallies on the host (use phylo_compare)");
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	const int rc = compare_pileup(c, part, nparts, (uint64_t *)dev_tri, nullptr, 2);
	c->stats["ms:compare_total"] += now_ms() - t0;
	c->stats["n:compare_calls"] += 1;
	return rc;
}

int phylo_triangle_to_matrices(phylo_ctx *c, const uint32_t *dev_tri, uint64_t *subst, uint64_t *homologs)
{
	if (!c) return 1;
	if (!dev_tri || !subst || !homologs) return c->fail("phylo_triangle_to_matrices: null argument");
	HIPOK(c, hipSetDevice(c->device));
	const size_t N = c->n;
	if (!N) return 0;
	HIPOK(c, c->h_mat.ensure(2 * N * N + 8));
	HIPOK(c, c->b_sym32.ensure(2 * N * N + 4));
	const double t0 = now_ms();
	hipLaunchKernelGGL(sym32_from_triangle_kernel, dim3((uint32_t)((N * N + 255) / 256)), dim3(256), 0, c->stream, (uint32_t)N, dev_tri, c->b_sym32.p);
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipMemcpyAsync(c->h_mat.p, c->b_sym32.p, 2 * N * N * 4, hipMemcpyDeviceToHost, c->stream));
	if (sync_stream(c)) return 1;
	const double t1 = now_ms();
	const double sites = widen_result(c, (const uint32_t *)c->h_mat.p, subst, homologs);
	c->stats["ms:triangle_copy"] += t1 - t0;
	c->stats["ms:triangle_widen"] += now_ms() - t1;
	c->stats["count:compare_sites"] += 0.5 * sites;
	return 0;
}

int phylo_anchor(phylo_ctx *c, size_t q_begin, size_t q_end) { return anchor_impl(c, q_begin, q_end, false); }

// what phylo_anchor does with the flags it waits for, after a deferred call's stream has been synchronised by somebody
// else: 0 lists and ranges are in place, 1 error, 2 a list needs the host (the optimistic state is withdrawn)
static int anchor_finish(phylo_ctx *c)
{
	const size_t N = c->n;
	const uint32_t *hr = c->h_rng.p, *dmisc = hr + 3 * N + 1;
	c->anchor_pending = false;
	size_t flagged = 0;
	for (size_t j = 0; j < N; j++) flagged += hr[2 * N + j] != 0;
	if (dmisc[3] || flagged) {
		c->homs_staged = false;
		c->eager_valid = false;
		c->att_homs = nullptr;
		c->host_stale.clear();
		if (dmisc[3]) return c->fail("phase A scratch overflow (code %u: 1 chunk log, 2 bridge pool, 3 homology buffer)", dmisc[3]);
		return 2;
	}
	c->att_rng_on_device = false;
	if (c->att_begin.size() != N) {
		c->att_begin.assign(N, 0);
		c->att_count.assign(N, 0);
	}
	c->host_stale.assign(N, 0);
	for (size_t j = 0; j < N; j++) {
		c->att_begin[j] = hr[2 * j];
		c->att_count[j] = hr[2 * j + 1] - hr[2 * j];
		c->host_stale[j] = 1;
	}
	c->stats["ms:anchor_setup"] += c->pend_t1 - c->pend_t0;
	c->stats["ms:anchor_total"] += c->pend_t2 - c->pend_t0; // (the host's part: the device's time is in phase B's wait)
	c->stats["n:anchor_calls"] += 1;
	c->stats["n:anchor_calls_without_a_wait"] += 1;
	c->stats["count:query_bases"] += c->pend_total;
	c->stats["count:chunks"] += c->pend_nch;
	c->stats["count:filtered_homologies"] += (double)hr[3 * N];
	c->stats["count:pool_blocks_used"] += dmisc[2];
	c->stats["count:overrun_runs"] += dmisc[5];
	c->stats["count:overrun_bytes_compared"] += dmisc[6];
	c->stats["anchor:chunk"] = c->pend_C;
	return 0;
}

// phylo_anchor(all genomes) + phylo_compare_all as the one call they are in the reference (process(), process.cxx:408-556):
// phase B is queued behind phase A without the host reading phase A's flags in between — one wait instead of two — and
// the flags are read with the result.  A list that needs the host after all (two homologies with the same projected
// start: the reference's order of such ties is libstdc++'s) sends the call the long way round.
int phylo_anchor_compare(phylo_ctx *c, uint64_t *subst, uint64_t *homologs)
{
	if (!c) return 1;
	int rc = anchor_impl(c, 0, c->n, true);
	if (rc) return rc;
	if (!c->anchor_pending) return phylo_compare(c, 0, 1, subst, homologs);
	rc = phylo_compare(c, 0, 1, subst, homologs); // (synchronises the stream whichever way it ends)
	if (c->anchor_pending && hipStreamSynchronize(c->stream) != hipSuccess) return c->fail("phylo_anchor_compare: the device failed");
	const int f = anchor_finish(c);
	if (f == 1) return 1;
	if (f == 0) return rc;
	c->stats["count:anchor_compare_calls_repeated"] += 1;
	rc = anchor_impl(c, 0, c->n, false);
	if (rc) return rc;
	return phylo_compare(c, 0, 1, subst, homologs);
}

int phylo_compare_all(phylo_ctx *c, uint64_t *subst, uint64_t *homologs)
{
	return phylo_compare(c, 0, 1, subst, homologs);
}

int phylo_process(phylo_ctx *c, size_t ref_idx, int flags, uint64_t *subst, uint64_t *homologs)
{
	if (!c) return 1;
	int rc = phylo_set_reference(c, ref_idx, nullptr, 0);
	if (rc) return rc;
	if (!(flags & PHYLO_COMPLETE_DELETION)) return phylo_anchor_compare(c, subst, homologs);
	rc = phylo_anchor(c, 0, c->n);
	if (rc) return rc;
	rc = phylo_complete_delete(c);
	if (rc) return rc;
	return phylo_compare_all(c, subst, homologs);
}

// ───────────────────────── B0 ─────────────────────────

int phylo_seqcmp_batch(phylo_ctx *c, size_t n, const uint32_t *ga, const uint64_t *offa, const uint32_t *gb,
					   const uint64_t *offb, const uint64_t *len, const uint8_t *rev, uint64_t *out)
{
	if (!c) return 1;
	if (n == 0) return 0;
	if (!ga || !offa || !gb || !offb || !len || !out) return c->fail("null segment arrays");
	HIPOK(c, hipSetDevice(c->device));
	std::vector<Segment> segs(n);
	for (size_t s = 0; s < n; s++) {
		if (ga[s] >= c->n || gb[s] >= c->n) return c->fail("segment %zu: genome index out of range", s);
		if (offa[s] + len[s] > c->glen[ga[s]] || offb[s] + len[s] > c->glen[gb[s]])
			return c->fail("segment %zu reaches beyond its genome", s);
		if (len[s] > 0xffffffffull) return c->fail("segment %zu too long", s);
		segs[s] = Segment{c->goff[ga[s]] + offa[s], c->goff[gb[s]] + offb[s], (uint32_t)len[s], rev && rev[s] ? 1u : 0u};
	}
	HIPOK(c, c->s_segs.ensure(n));
	HIPOK(c, c->s_out.ensure(n));
	HIPOK(c, hipMemcpyAsync(c->s_segs.p, segs.data(), n * sizeof(Segment), hipMemcpyHostToDevice, c->stream));
	int blocks = std::min<int>(c->n_cu * 8, (int)((n + 3) / 4));
	{
		KernelSpan s(c, "seqcmp_batch");
		launch_seqcmp_batch(c->d_genomes, c->s_segs.p, (uint32_t)n, c->s_out.p, blocks, c->stream);
	}
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipMemcpyAsync(out, c->s_out.p, n * 8, hipMemcpyDeviceToHost, c->stream));
	return sync_stream(c);
}

// seqcmp / revseqcmp with the reference's signature and calling convention (libs/seqcmp.h:14-25,
// libs/revseqcmp.h:25-33): pure, borrowing both host buffers, callable from many threads at once
// (evo_model::account* runs inside an OpenMP team, src/evo_model.cxx:53-75).  Every calling thread
// gets its own context — stream, scratch, nothing shared — created on its first call and released
// when the thread ends.  The signature has no error channel and the library never aborts or falls
// back to the CPU: a failed call returns SIZE_MAX, the message is in phylo_last_error(NULL) of that
// thread and is printed to stderr once per thread.
namespace {
struct B0Thread {
	phylo_ctx *ctx = nullptr;
	DevBuf<uint8_t> buf;
	bool complained = false;
	~B0Thread()
	{
		if (ctx) {
			(void)hipSetDevice(ctx->device);
			buf.release();
			phylo_ctx_destroy(ctx);
		}
	}
};
thread_local B0Thread g_b0;

size_t b0_fail(const char *what)
{
	if (what) g_last_error = what;
	if (!g_b0.complained) {
		fprintf(stderr, "phylonium_amd: seqcmp/revseqcmp: %s\n", g_last_error.c_str());
		g_b0.complained = true;
	}
	return (size_t)-1;
}

size_t b0_call(const char *a, const char *b, size_t length, int rev)
{
	if (length == 0) return 0;
	if (!a || !b) return b0_fail("null buffer");
	B0Thread &t = g_b0;
	if (!t.ctx && phylo_ctx_create(&t.ctx, 0)) return b0_fail(nullptr);
	phylo_ctx *c = t.ctx;
	if (hipSetDevice(c->device) != hipSuccess) return b0_fail("hipSetDevice failed");
	// both strings into this thread's scratch, 64-byte aligned starts; pieces of < 2^32 bytes
	const uint64_t stride = (length + 63) / 64 * 64 + 64;
	if (t.buf.ensure(2 * stride + 64) != hipSuccess) return b0_fail("out of device memory");
	const uint64_t piece = 1ull << 30;
	std::vector<Segment> segs;
	for (uint64_t o = 0; o < length; o += piece) {
		const uint64_t m = std::min<uint64_t>(piece, length - o);
		segs.push_back(Segment{o, stride + (rev ? length - o - m : o), (uint32_t)m, rev ? 1u : 0u});
	}
	std::vector<uint64_t> out(segs.size());
	if (c->s_segs.ensure(segs.size()) != hipSuccess || c->s_out.ensure(segs.size()) != hipSuccess) return b0_fail("out of device memory");
	hipStream_t st = c->stream;
	if (hipMemcpyAsync(t.buf.p, a, length, hipMemcpyHostToDevice, st) != hipSuccess ||
		hipMemcpyAsync(t.buf.p + stride, b, length, hipMemcpyHostToDevice, st) != hipSuccess ||
		hipMemcpyAsync(c->s_segs.p, segs.data(), segs.size() * sizeof(Segment), hipMemcpyHostToDevice, st) != hipSuccess)
		return b0_fail("upload failed");
	const int blocks = std::max<int>(1, std::min<int>(c->n_cu * 8, (int)((length / 4096) + 1)));
	launch_seqcmp_batch(t.buf.p, c->s_segs.p, (uint32_t)segs.size(), c->s_out.p, blocks, st);
	if (hipGetLastError() != hipSuccess ||
		hipMemcpyAsync(out.data(), c->s_out.p, segs.size() * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
		hipStreamSynchronize(st) != hipSuccess)
		return b0_fail("kernel launch or read-back failed");
	uint64_t total = 0;
	for (uint64_t v : out) total += v;
	return (size_t)total;
}
} // namespace

size_t phylo_seqcmp(const char *begin, const char *other, size_t length) { return b0_call(begin, other, length, 0); }
size_t phylo_revseqcmp(const char *begin, const char *other, size_t length) { return b0_call(begin, other, length, 1); }

// ───────────────────────── host-side helpers ─────────────────────────

// `s` is followed by 16 zero bytes
static int host_suffix_array_padded(const uint8_t *s, size_t n, int64_t *sa)
{
	std::vector<uint32_t> tmp(n);
	unsigned h = std::thread::hardware_concurrency();
	ThreadFan fan{std::min<size_t>(h ? h : 1, 16)};
	suffix_array_u32_par(s, (uint32_t)n, tmp.data(), fan, fan.nthreads);
	fan(64, [&](size_t t) {
		for (size_t i = n * t / 64, e = n * (t + 1) / 64; i < e; i++) sa[i] = tmp[i];
	});
	return 0;
}

int phylo_host_suffix_array(const char *s, size_t n, int64_t *sa)
{
	if (!s || !sa || n >= 0x7fffffffull) return 1;
	std::vector<uint8_t> padded(n + 16, 0);
	memcpy(padded.data(), s, n);
	return host_suffix_array_padded(padded.data(), n, sa);
}

int phylo_host_reference_suffix_array(const char *ref, size_t len, int64_t *sa)
{
	if (!ref || !sa || 2 * len + 1 >= 0x7fffffffull) return 1;
	std::vector<uint8_t> S(2 * len + 1 + 64, 0);
	memcpy(S.data(), ref, len);
	S[len] = '#';
	revcomp((const uint8_t *)ref, len, S.data() + len + 1);
	return host_suffix_array_padded(S.data(), 2 * len + 1, sa);
}

size_t phylo_host_min_anchor_length(double p, double gc, size_t l) { return min_anchor_length(p, gc, l); }

int phylo_host_device_count(int *count)
{
	if (!count) return 1;
	*count = 0;
	return hipGetDeviceCount(count) == hipSuccess ? 0 : 1;
}

int phylo_host_read_fasta(size_t n, const char *const *paths, size_t threads, char **out, size_t *len)
{
	if (!paths || !out || !len) return 1;
	std::vector<std::string> files(paths, paths + n);
	std::vector<phyfasta::ReadResult> res(n);
	ThreadFan fan{std::max<size_t>(1, threads)};
	std::vector<char *> bufs(n, nullptr);
	std::vector<size_t> sizes(n, 0);
	std::vector<std::string> errors(n);
	fan(n, [&](size_t i) { // read, filter and hand over in the same task: the copies run on all threads
		phyfasta::ReadResult r = phyfasta::read_genome(files[i]);
		if (!r.error.empty()) {
			errors[i] = r.error;
			return;
		}
		const std::string &g = r.g.nucl;
		bufs[i] = (char *)malloc(g.size() + 1);
		if (!bufs[i]) {
			errors[i] = files[i] + ": out of memory";
			return;
		}
		memcpy(bufs[i], g.data(), g.size());
		bufs[i][g.size()] = 0;
		sizes[i] = g.size();
	});
	for (size_t i = 0; i < n; i++)
		if (!errors[i].empty()) {
			g_last_error = errors[i];
			for (size_t k = 0; k < n; k++) free(bufs[k]);
			return (int)(i + 1);
		}
	for (size_t i = 0; i < n; i++) {
		out[i] = bufs[i];
		len[i] = sizes[i];
	}
	return 0;
}

int phylo_host_read_fasta_packed(size_t n, const char *const *paths, size_t threads, uint32_t **q2, size_t *len, uint32_t **bad,
								 size_t *nbad, void **arena)
{
	if (!paths || !q2 || !len || !bad || !nbad || !arena) return 1;
	std::vector<std::string> files(paths, paths + n);
	std::string error;
	uint32_t *words = nullptr;
	size_t bad_idx = 0; // the first file, in the order given, that failed
	std::vector<phyfasta::PackedGenome> g = phyfasta::read_genomes_packed(files, std::max<size_t>(1, threads), &error, &words, &bad_idx);
	// one more allocation holds the separator lists and whatever did not fit its place in the first (a file that was
	// not a regular file): both leave with the arena
	size_t extra = 0;
	for (auto &x : g) extra += x.bad.size() + 1 + (x.own ? (size_t)((x.len + 15) / 16) + 16 : 0);
	uint32_t *second = error.empty() ? (uint32_t *)malloc((extra + 16) * sizeof(uint32_t)) : nullptr;
	if (error.empty() && (!words || !second)) error = "out of memory";
	if (!error.empty()) {
		g_last_error = error;
		for (size_t i = 0; i < n; i++)
			if (g[i].own) free(g[i].q2);
		free(words);
		free(second);
		return (int)(bad_idx + 1);
	}
	// header of the handle: the two allocations
	void **handle = (void **)malloc(2 * sizeof(void *));
	if (!handle) {
		for (auto &x : g)
			if (x.own) free(x.q2);
		free(words);
		free(second);
		g_last_error = "out of memory";
		return 1;
	}
	handle[0] = words;
	handle[1] = second;
	size_t w = 0;
	for (size_t i = 0; i < n; i++) {
		if (g[i].own) {
			const size_t nw = (size_t)((g[i].len + 15) / 16);
			w = (w + 15) / 16 * 16;
			if (nw) memcpy(second + w, g[i].q2, nw * sizeof(uint32_t));
			free(g[i].q2);
			q2[i] = second + w;
			w += nw;
		} else {
			q2[i] = g[i].q2;
		}
		len[i] = (size_t)g[i].len;
	}
	for (size_t i = 0; i < n; i++) {
		bad[i] = second + w;
		nbad[i] = g[i].bad.size();
		if (nbad[i]) memcpy(second + w, g[i].bad.data(), nbad[i] * sizeof(uint32_t));
		w += nbad[i];
	}
	*arena = handle;
	return 0;
}

void phylo_host_free_packed(void *arena)
{
	if (!arena) return;
	void **handle = (void **)arena;
	free(handle[0]);
	free(handle[1]);
	free(handle);
}

void phylo_host_free(void *p) { free(p); }

size_t phylo_host_median_length_index(size_t n, const size_t *len)
{
	if (!n || !len) return 0;
	std::vector<size_t> idx(n);
	for (size_t i = 0; i < n; i++) idx[i] = i;
	std::nth_element(idx.begin(), idx.begin() + n / 2, idx.end(), [&](size_t a, size_t b) { return len[a] < len[b]; });
	return idx[n / 2];
}

size_t phylo_host_sort_filter(phylo_homology *h, size_t n, int do_sort)
{
	std::vector<phylo_homology> v(h, h + n);
	if (do_sort) {
		// the path phase A takes: packed keys, structs only when two entries share a start
		SortFilterScratch scratch;
		std::vector<uint32_t> kept;
		auto get = [&](size_t i, uint64_t *start, uint64_t *len) {
			*start = v[i].index_reference_projected;
			*len = v[i].length;
		};
		if (sort_filter_order(n, get, scratch, kept)) {
			for (size_t t = 0; t < kept.size(); t++) h[t] = v[kept[t]];
			return kept.size();
		}
		sort_and_filter(v);
	} else {
		filter_overlaps_max(v);
	}
	std::copy(v.begin(), v.end(), h);
	return v.size();
}

double phylo_estimate(int kind, uint64_t subst, uint64_t homologs, int zero_on_error)
{
	switch (kind) {
		case 0: return estimate_jc(subst, homologs, zero_on_error != 0);
		case 1: return estimate_raw(subst, homologs, zero_on_error != 0);
		default: return estimate_ani(subst, homologs, zero_on_error != 0);
	}
}

// src/io.cxx:141-163
size_t phylo_format_phylip(size_t n, const char *const *names, const uint64_t *subst, const uint64_t *homologs,
						   int kind, char *out, size_t cap)
{
	// just_print, io.cxx:141-163: precision 4 with std::scientific ("%.4e"), or the default float format for ANI
	// (std::dec does not touch it: "%.4g").  Row blocks are formatted on the host threads and joined in order.
	const char *fmt = kind == 2 ? "  %.4g" : "  %.4e";
	unsigned hw = std::thread::hardware_concurrency();
	const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 1, 16), n / 16 + 1));
	std::vector<std::string> part(nt);
	ThreadFan fan{nt};
	fan(nt, [&](size_t t) {
		std::string &o = part[t];
		const size_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
		o.reserve((i1 - i0) * (n * 12 + 32));
		char buf[64];
		for (size_t i = i0; i < i1; i++) {
			o += names[i];
			for (size_t j = 0; j < n; j++) {
				const double d = (i == j) ? 0.0 : phylo_estimate(kind, subst[i * n + j], homologs[i * n + j], 0);
				o.append(buf, (size_t)snprintf(buf, sizeof buf, fmt, d));
			}
			o += '\n';
		}
	});
	std::string head = std::to_string(n) + "\n";
	size_t need = head.size() + 1;
	for (auto &p : part) need += p.size();
	if (out && cap >= need) {
		char *w = out;
		memcpy(w, head.data(), head.size());
		w += head.size();
		for (auto &p : part) {
			memcpy(w, p.data(), p.size());
			w += p.size();
		}
		*w = 0;
	}
	return need;
}

} // extern "C"

// abi_ctx.hpp — what the translation units behind the C ABI (include/phylonium_amd.h) share: the context
// (struct phylo_ctx: device buffers, options, the host-side lists, statistics), grow-only device and pinned
// buffers, the host worker pool, kernel timing spans, and the helpers that cross the files:
//   abi_context.hip    create / destroy / options / statistics
//   abi_genomes.hip    phylo_set_genomes*: layout, upload, 2-bit packing, separator lists
//   abi_reference.hip  phylo_set_reference: S, suffix array, LCP, SAX, the k-mer slot table
//   abi_anchor.hip     phase A: plan, chain kernels, fold, sort + filter (process.cxx:433-458)
//   abi_lists.hip      homology lists in and out, the exchange between ranks, complete deletion
//   abi_compare.hip    phase B: projection, pair tallies, results; both phases as one call (process.cxx:517-549)
//   abi_result.hip     the result's home in page-locked memory, private or shared by the ranks of a node
//   abi_host.hip       seam B0 (seqcmp / revseqcmp) and the host-side helpers (FASTA, suffix array, PHYLIP)
// Mirrors process() of /root/reference/src/process.cxx:408-556.  There is no CPU compute fallback: every entry
// point that does the path's arithmetic launches HIP kernels and fails if no device is usable.
#pragma once
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
#include "../host/fmt_e4.hpp"
#include "hostlogic.hpp"
#include "kernels.h"

using namespace phy;

extern thread_local std::string g_phylo_last_error; // abi_context.hip

namespace {


double now_ms()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

#ifdef PHY_DEV_HOOKS
// PHY_DEBUG_ALLOC=1 (development builds): every device / page-locked allocation and release on stderr, with the process id
inline void debug_alloc(const char *what, const void *p, size_t bytes)
{
	static const bool on = getenv("PHY_DEBUG_ALLOC") != nullptr;
	if (on) fprintf(stderr, "[phylonium_amd alloc %d] %s %p .. %p (%zu bytes)\n", (int)getpid(), what, p, (const char *)p + bytes, bytes);
}
// PHY_DEBUG_POISON=<byte> (development builds): every new device buffer is filled with that byte — a kernel that reads what
// nobody has written yet (fresh memory is zero in a new process, and whatever was there in a warm one) then fails every time
inline void debug_poison(void *p, size_t bytes)
{
	static const char *e = getenv("PHY_DEBUG_POISON");
	if (e && p) {
		(void)hipMemset(p, (int)strtol(e, nullptr, 0), bytes);
		(void)hipDeviceSynchronize();
	}
}
#else
inline void debug_alloc(const char *, const void *, size_t) {}
inline void debug_poison(void *, size_t) {}
#endif

template <class T> struct DevBuf {
	T *p = nullptr;
	size_t cap = 0; // elements
	hipError_t ensure(size_t n)
	{
		if (n <= cap) return hipSuccess;
		if (p) {
			debug_alloc("free  ", p, cap * sizeof(T));
			(void)hipFree(p);
		}
		p = nullptr;
		cap = 0;
		size_t want = n + n / 8 + 16;
		hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
		if (e == hipSuccess) cap = want;
		debug_alloc("malloc", p, want * sizeof(T));
		if (e == hipSuccess) debug_poison(p, want * sizeof(T));
		return e;
	}
	void release()
	{
		if (p) {
			debug_alloc("free  ", p, cap * sizeof(T));
			(void)hipFree(p);
		}
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
		debug_alloc("pinned", p, want * sizeof(T));
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
	int opt_filter_kernel = 0; // option "filter_kernel": 0 stretch-wise chain filter (then the general kernel for what it hands over), 1 general only
	int opt_sa_builder = 1; // option "sa_builder": who builds the suffix array when the caller brings none — 1 the device, 0 the host cores
	uint32_t opt_pairs_wchunk = 0; // option "pairs_wchunk": windows per chunk of the pair kernel (0: chosen from the L2 size)
	uint32_t opt_spec_blocks = 0; // option "spec_blocks": at most that many blocks of the speculative chain kernel (0: the plan's) — few blocks make
	                              // every lane take chunk after chunk from the queue
	uint32_t opt_fold_blocks = 0; // option "fold_blocks": blocks per query of the fold kernel (0: chosen from the number of queries)
	int opt_pairs_kernel = 0; // option "pairs_kernel": 0 the matrix-core kernel when no projected position holds '!' (default), 1 the vector-ALU kernel always
	std::string err;
	int n_cu = 256;
	int proj_resident[2] = {256, 256}; // blocks of the projection this device holds at once (project_resident_blocks)

	// options
	uint32_t opt_chunk = 0, opt_kmer = 0;
	int plan_spec_per_cu = 4; // blocks of the speculative chain kernel per CU the plan was made for
	int profile = 0; // option "profile": 0 no kernel timing, 1 every kernel (two HIP events each: ~4 us a kernel), 2 the chain kernel only (the bench's timed loop)
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
	U4 *slot_at = nullptr; // the slot table inside d_SLOT (development builds can place it on a 1 GiB boundary: PHY_SLOT_ALIGN_GB)
	DevBuf<uint32_t> d_SA;
	DevBuf<uint32_t> d_LCP, d_T;
	// 2-bit packed companions for the lean chain kernels (lean_core.h): genomes and S, 16 bases per
	// dword, and the sorted positions of their non-ACGT bytes
	DevBuf<uint32_t> d_Q2, d_QBAD, d_qbad_off, d_S2, d_SBAD, d_badscr;
	DevBuf<uint64_t> d_badoff;
	uint32_t nsb = 0, sb_first = 0;
	bool cache_quirk = false; // the reference's 6-mer cache over-reports matches on this subject (hostlogic.hpp: esa_cache_quirks)
	DevBuf<U4> d_quirk;       // its over-deep entries {prefix, k | depth << 8, lo, hi} for the chains' slow resolver (lean_core.h)
	uint32_t nquirk = 0;
	int opt_cache_quirk = 1; // option "cache_quirk": 1 reproduce what the reference answers on such a subject (default), 0 the true longest matches
	int lean_force_slow = 0;

	// phase A scratch
	DevBuf<uint64_t> a_qoff;
	DevBuf<uint32_t> a_qlen, a_qchunk0, a_qanc0, a_items, a_chunk_query, a_spec_cnt, a_visited, a_misc;
	DevBuf<WorkItem> a_work; // the plan's work order as items to start from, and the queries' descriptors (lean_work_kernel)
	DevBuf<QDesc> a_qdesc;
	bool work_stale = true;
	DevBuf<Anchor> a_spec_anchors;
	DevBuf<SpecExit> a_spec_exit;
	DevBuf<BridgeRec> a_bridge;
	DevBuf<uint32_t> a_bridge_start; // the bridges that need walking, packed (lean_core.h: LeanBridge::pack)
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
	// ... for a range of the queries, its exchange block written behind it (phylo_anchor_block_device): the lists are described
	// on the device only until phyabi::settle_anchor has read the flags — or the gathered blocks are attached, after which only
	// its statistics are still to be collected (pend_stats_only)
	bool pend_range = false, pend_stats_only = false;
	size_t pend_qb = 0, pend_qe = 0;
	void *xb_block = nullptr; // where anchor_impl (defer = 2) writes the block
	size_t xb_maxq = 0, xb_cap = 0;
	std::vector<uint32_t> xb_bounds;        // the ranks' bounds as phylo_attach_blocks_device uploaded them last ...
	const uint32_t *xb_bounds_at = nullptr; // ... and where (skipped while both stay the same)
	bool flags_zeroed = false;              // the block export of a queued phase A has zeroed b_flag for the attach that follows
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
	DevBuf<unsigned long long> b_subst; // both tallies, N x N each
	uint64_t tiles_key = 0;         // what b_tiles holds (compare_pileup)
	const uint32_t *tiles_at = nullptr;
	DevBuf<uint32_t> b_sym32; // both result matrices as symmetric u32, on their way to the host
	DevBuf<uint32_t> b_bang;  // the projected '!' of the three-plane projection: {genome | reverse << 31, position} each
	DevBuf<unsigned long long> b_clk; // option "profile": the matrix-core pair kernel's wavefront lifetimes {shader cycles, 100 MHz ticks}
	uint32_t bang_cap = 0;    // as many as the genomes hold separators (a separator is projected at most once)
	DevBuf<Segment> s_segs;
	DevBuf<uint64_t> s_out;
	DevBuf<uint32_t> s_piece0; // a batch's segments: their rounds' prefix sums (seqcmp_kernels.hip)
	DevBuf<uint8_t> s_rounds;  //   and the rounds' descriptors, written by the device
	PinBuf<uint32_t> h_piece0;

	// host staging and workers
	PinBuf<uint32_t> h_cnt;
	PinBuf<RawHom> h_raw;
	PinBuf<DevHom> h_devhom;
	PinBuf<uint64_t> h_mat;
	// result matrices of a caller that keeps handing the same host buffers over, registered so that the device writes them
	// itself (phylo_triangle_to_matrices): option "result_zero_copy" = 1.  Off by default: memory registered with the HIP
	// runtime must not be freed, forked over or handed to another registration while it is — a promise only the caller can
	// make (measured: a process that registered and released numpy arrays took a GPU "write access to a read-only page"
	// fault in unrelated work minutes later)
	struct HostReg {
		void *ptr;
		size_t bytes;
		void *dev;
		bool failed;
	};
	std::vector<HostReg> host_regs;
	int opt_result_zero_copy = 0;
	// the result's own home (abi_result.hip): page-locked host memory the library owns — private to this context, or a POSIX
	// shared-memory segment every rank of a node maps, so that each rank's device writes its rows of the result itself
	struct ResultHome {
		void *map = nullptr;      // header (a 64-byte slot per rank: the steps it has delivered), then the two n x n u64 matrices
		size_t bytes = 0, n = 0, ranks = 0;
		void *dev = nullptr;      // the mapping as this context's device addresses it
		bool shared = false, creator = false, linked = false;
		std::string name;
		uint64_t step = 0;        // deliveries so far (the ranks call in step)
		size_t matrix_words() const { return (n * n + 7) / 8 * 8; } // (either matrix starts on a 64-byte boundary: the device stores 16 bytes at a time)
		uint64_t *subst() const { return (uint64_t *)((char *)map + 4096); }
		uint64_t *homologs() const { return subst() + matrix_words(); }
	} res;
	std::unique_ptr<WorkerPool> pool;
	// cached phase-A plan
	bool plan_valid = false;
	size_t plan_qb = 0, plan_qe = 0;
	ChunkPlan plan;
	std::vector<uint64_t> plan_out_base;
	uint64_t plan_raw_total = 0;
	size_t plan_nq_real = 0; // the plan's queries that have chunks

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
		g_phylo_last_error = buf;
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
	bool on_ = false;
	KernelSpan(phylo_ctx *ctx, const char *nm, hipStream_t on = nullptr) : c(ctx), name(nm), st(on ? on : ctx->stream)
	{
		on_ = c->profile == 1 || (c->profile == 2 && !strcmp(nm, "anchor_spec"));
		if (on_) {
			a = get_event(c);
			b = get_event(c);
			(void)hipEventRecord(a, st);
		}
	}
	~KernelSpan()
	{
		if (on_) {
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

// a phase A queued by phylo_anchor_block_device that nobody will ask about any more (new genomes, another reference)
void drop_pending_anchor(phylo_ctx *c)
{
	if (c->anchor_pending && c->pend_range) {
		(void)hipSetDevice(c->device);
		(void)sync_stream(c);
	}
	c->anchor_pending = c->pend_range = c->pend_stats_only = false;
}

QuerySrc query_src(const phylo_ctx *c) { return QuerySrc{c->d_Q2.p, c->d_goff.p, c->d_QBAD.p, c->d_qbad_off.p}; }

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

// helpers that cross the files (defined where their section lives, inside the files' extern "C" blocks; not exported)
#define PHYABI_LOCAL __attribute__((visibility("hidden")))
extern "C" {
namespace phyabi {
// abi_genomes.hip: sorted non-ACGT positions of `lens.size()` sequences (genomes, or S as one) into `out`
PHYABI_LOCAL int bad_lists(phylo_ctx *c, const uint8_t *base, const uint64_t *d_off, const uint32_t *d_len, const std::vector<uint64_t> &lens,
			  DevBuf<uint32_t> &out, std::vector<uint32_t> &off, size_t n);
// abi_anchor.hip: phase A for queries [q_begin, q_end); defer 1: all genomes, the projection queued behind it, its flags left for
// phylo_anchor_compare to read; defer 2: a range, its exchange block (c->xb_*) written behind it, its flags left for settle_anchor
PHYABI_LOCAL int anchor_impl(phylo_ctx *c, size_t q_begin, size_t q_end, int defer);
// a phase A queued by phylo_anchor_block_device whose flags nobody has read yet: wait for it and take its lists in (repeating
// it the long way when a list needs the host); after the gathered blocks were attached: its statistics only.  0, or 1 on error
PHYABI_LOCAL int settle_anchor(phylo_ctx *c);
// abi_lists.hip: the exchange block of queries [q_begin, q_end) from where phase A's device filter left the lists (queued);
// flt_flags / misc: the filter's per-query flags and phase A's counters, whose verdict rides in the block's header (or null)
// host_out (with them): page-locked words the kernel fills as anchor_impl's copies would (ranges, total, flags, counters)
PHYABI_LOCAL int queue_block_export(phylo_ctx *c, size_t nq, void *dev_block, size_t max_queries, size_t cap_records, const uint32_t *flt_flags,
								   const uint32_t *misc, uint32_t *host_out);
// abi_lists.hip
PHYABI_LOCAL int ensure_host_lists(phylo_ctx *c, size_t g0, size_t g1);
PHYABI_LOCAL int fetch_att_ranges(phylo_ctx *c);
// abi_host.hip: out[s] = seqcmp / revseqcmp of segment s of `base` (device), for n segments given on the host; waits for the result
PHYABI_LOCAL int run_segments(phylo_ctx *c, const uint8_t *base, const phy::Segment *segs, size_t n, uint64_t *out, const char *span);
// abi_compare.hip: the pileup of part `part` of `nparts` (a range of 64-window tiles of the reference)
PHYABI_LOCAL int make_pileup(phylo_ctx *c, size_t part, size_t nparts, phy::Pileup *out);
} // namespace phyabi
} // extern "C"

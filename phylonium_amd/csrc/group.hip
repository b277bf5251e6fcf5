// group.hip — several GPUs of one node behind one host: phylo_group_* of include/phylonium_amd.h.
//
// process() of /root/reference/src/process.cxx:408-556 shards without a data-path collective inside either phase
// (SURVEY §8e): phase A is independent per query (the OpenMP loop at process.cxx:433-434), phase B per reference
// window of the pileup (the pair loop at process.cxx:524-529, re-cut by window range so that projection and pair
// kernel both shrink with the ranks).  A group is one context (phylo_ctx) and one host thread per rank; the ranks
// meet three times per pass:
//   genomes     every rank uploads its block of the packed genomes, one all-gather leaves all of them on every GPU
//   lists       after phase A: fixed-shape exchange blocks, one all-gather (phylo_export_block_device /
//               phylo_attach_blocks_device — the records never visit the host)
//   tallies     after phase B: the parts' u32 triangles, one reduce to rank 0
// over RCCL (xGMI) when every rank has a GPU of its own — the library is loaded when a group asks for it, a
// single-GPU host never pays for it — and by device-to-device copies when ranks share a GPU (a test box with one)
// or RCCL is not there.  Written over the public C ABI and the HIP runtime: it is the host a maintainer would write.
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define PHY_HAVE_RCCL_HEADER 1
#else // a build without the RCCL development package: the handful of declarations the dlopen'ed library is called through
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1, ncclUint32 = 3 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm,
						hipStream_t stream);
const char *ncclGetErrorString(ncclResult_t result);
}
#endif

#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/phylonium_amd.h"

namespace {

struct Rccl {
	void *lib = nullptr;
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclReduce) Reduce = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	bool load()
	{
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (lib) break;
		}
		if (!lib) return false;
		CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
		CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
		AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
		Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
		GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
		return CommInitAll && CommDestroy && AllGather && Reduce && GetErrorString;
	}
};

// the ranks' host threads: run(f) executes f(r) on rank r's thread for every r and returns when all are done
class RankThreads
{
	std::vector<std::thread> threads;
	std::mutex m;
	std::condition_variable cv_work, cv_done;
	std::function<void(size_t)> job;
	size_t generation = 0, running = 0;
	bool stop = false;

	void loop(size_t r)
	{
		size_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> lk(m);
				cv_work.wait(lk, [&] { return stop || generation != seen; });
				if (stop) return;
				seen = generation;
			}
			job(r);
			std::unique_lock<std::mutex> lk(m);
			if (--running == 0) cv_done.notify_all();
		}
	}

  public:
	explicit RankThreads(size_t n)
	{
		for (size_t r = 0; r < n; r++) threads.emplace_back([this, r] { loop(r); });
	}
	~RankThreads()
	{
		{
			std::unique_lock<std::mutex> lk(m);
			stop = true;
		}
		cv_work.notify_all();
		for (auto &t : threads) t.join();
	}
	void run(std::function<void(size_t)> f)
	{
		std::unique_lock<std::mutex> lk(m);
		job = std::move(f);
		running = threads.size();
		generation++;
		cv_work.notify_all();
		cv_done.wait(lk, [&] { return running == 0; });
	}
};

// Where the ranks' threads wait for each other inside a job.  Every rank says whether it has failed so far and all
// of them get the same answer — "somebody has" — so that the ranks leave a job together and never wait for a rank
// that has given up (in a barrier or, worse, inside a collective).
class Barrier
{
	std::mutex m;
	std::condition_variable cv;
	size_t n, waiting = 0, phase = 0;
	bool acc = false, result = false;

  public:
	explicit Barrier(size_t count) : n(count) {}
	bool wait(bool bad = false)
	{
		std::unique_lock<std::mutex> lk(m);
		acc = acc || bad;
		const size_t my = phase;
		if (++waiting == n) {
			result = acc;
			acc = false;
			waiting = 0;
			phase++;
			cv.notify_all();
		} else {
			cv.wait(lk, [&] { return phase != my; });
		}
		return result;
	}
};

#ifdef PHY_DEV_HOOKS
// (development builds only: `make dev`) tests of the hosts' self-check (phylonium-amd --verify-ranks):
// PHYLONIUM_AMD_TEST_CORRUPT_RANK=r makes rank r send one damaged record — its first homology's query position moved
// by one base, a list as valid as any — into the exchange
__global__ void corrupt_block_kernel(uint32_t *block, uint32_t maxq)
{
	if (threadIdx.x == 0 && block[0] > 0) block[4 + maxq + 1] += 1u; // (block: 4 header words, the lengths, the records {start, iq, len, rev})
}
#endif

__global__ __launch_bounds__(256) void add_u32_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, size_t n)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] += src[i];
}

double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

thread_local std::string g_group_error;

#ifdef PHY_DEV_HOOKS
// PHY_DEBUG_ABORT=1 (experiments): the native stack of the thread that aborts, on stderr
void abort_trace(int sig)
{
	void *frames[64];
	const int n = backtrace(frames, 64);
	const char msg[] = "[phylonium_amd] fatal signal, native stack:\n";
	(void)!write(2, msg, sizeof msg - 1);
	backtrace_symbols_fd(frames, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}
#endif

} // namespace

struct phylo_group {
	size_t world = 0;
	std::vector<int> dev;
	std::vector<phylo_ctx *> ctx;
	std::vector<hipStream_t> stream;
	bool use_rccl = false;
	Rccl rccl;
	std::vector<ncclComm_t> comm;
	std::unique_ptr<RankThreads> threads;
	std::unique_ptr<Barrier> barrier;
	std::string err;
	std::mutex err_m;

	// genomes and their split over the ranks
	size_t n = 0;
	std::vector<uint64_t> glen;
	std::vector<size_t> bounds; // rank r anchors genomes [bounds[r], bounds[r+1])

	// the exchange's plan and buffers (per rank, on the rank's device)
	bool plan_valid = false;
	size_t maxq = 0, cap = 0, block_bytes = 0;
	std::vector<void *> d_all;
	std::vector<uint32_t *> d_tri;
	std::vector<size_t> own_total;
	bool lists_everywhere = false; // the last phylo_group_anchor left every rank with all lists
	size_t replans = 0;            // passes repeated because the lists had outgrown the planned blocks
	size_t forced_cap = 0;         // option "exchange_cap": records per exchange block of the next plan (0: from the lists' lengths)

	std::vector<double> t_anchor, t_exchange, t_compare, t_reduce; // ms of the last pass, per rank

	int fail(const char *fmt, ...)
	{
		char buf[1024];
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(buf, sizeof buf, fmt, ap);
		va_end(ap);
		std::lock_guard<std::mutex> lk(err_m);
		if (err.empty()) err = buf;
		g_group_error = buf;
		return 1;
	}
	void clear_error()
	{
		std::lock_guard<std::mutex> lk(err_m);
		err.clear();
	}
	bool failed()
	{
		std::lock_guard<std::mutex> lk(err_m);
		return !err.empty();
	}
	void release_plan()
	{
		for (size_t r = 0; r < world; r++) {
			(void)hipSetDevice(dev[r]);
			if (r < d_all.size() && d_all[r]) (void)hipFree(d_all[r]);
			if (r < d_tri.size() && d_tri[r]) (void)hipFree(d_tri[r]);
		}
		d_all.assign(world, nullptr);
		d_tri.assign(world, nullptr);
		plan_valid = false;
	}
};

namespace {

// Rank r's piece (bytes at recvbuf[r] + r * bytes) to every rank's recvbuf: one RCCL all-gather, or — ranks sharing
// a device, no RCCL — every rank fetching the other ranks' pieces by device-to-device copies.  `bad`: this rank has
// failed before; every rank returns true when any rank has (then nothing was exchanged).
bool all_gather(phylo_group *g, size_t r, const std::vector<void *> &recvbuf, size_t bytes, bool bad)
{
	hipStream_t st = g->stream[r];
	if (g->use_rccl) {
		if (g->barrier->wait(bad)) return true;
		const ncclResult_t rc = g->rccl.AllGather((const char *)recvbuf[r] + r * bytes, recvbuf[r], bytes, ncclUint8, g->comm[r], st);
		if (rc != ncclSuccess) {
			g->fail("ncclAllGather: %s", g->rccl.GetErrorString(rc));
			return true;
		}
		return false;
	}
	if (!bad && hipStreamSynchronize(st) != hipSuccess) bad = g->fail("all-gather by copies: stream synchronisation failed") != 0;
	if (g->barrier->wait(bad)) return true; // every rank's own piece is complete
	for (size_t o = 0; o < g->world && !bad; o++)
		if (o != r && hipMemcpyPeerAsync((char *)recvbuf[r] + o * bytes, g->dev[r], (const char *)recvbuf[o] + o * bytes, g->dev[o], bytes, st) != hipSuccess)
			bad = g->fail("all-gather by copies: device-to-device copy failed") != 0;
	if (!bad && hipStreamSynchronize(st) != hipSuccess) bad = g->fail("all-gather by copies: stream synchronisation failed") != 0;
	return g->barrier->wait(bad); // nobody overwrites a piece another rank is still reading
}

// the sum of every rank's buf (count u32) into rank 0's; `bad` and the result as above
bool reduce_to_rank0(phylo_group *g, size_t r, const std::vector<uint32_t *> &buf, size_t count, bool bad)
{
	hipStream_t st = g->stream[r];
	if (g->use_rccl) {
		if (g->barrier->wait(bad)) return true;
		const ncclResult_t rc = g->rccl.Reduce(buf[r], buf[r], count, ncclUint32, ncclSum, 0, g->comm[r], st);
		if (rc != ncclSuccess) {
			g->fail("ncclReduce: %s", g->rccl.GetErrorString(rc));
			return true;
		}
		return false;
	}
	if (!bad && hipStreamSynchronize(st) != hipSuccess) bad = g->fail("reduce by copies: stream synchronisation failed") != 0;
	if (g->barrier->wait(bad)) return true;
	if (r == 0 && count) {
		for (size_t o = 1; o < g->world && !bad; o++) {
			const uint32_t *src = buf[o];
			uint32_t *tmp = nullptr;
			if (g->dev[o] != g->dev[0]) { // bring it over first
				if (hipMalloc((void **)&tmp, count * 4) != hipSuccess ||
					hipMemcpyPeerAsync(tmp, g->dev[0], buf[o], g->dev[o], count * 4, st) != hipSuccess) {
					bad = g->fail("reduce by copies: device-to-device copy failed") != 0;
					if (tmp) (void)hipFree(tmp);
					break;
				}
				src = tmp;
			}
			hipLaunchKernelGGL(add_u32_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, st, buf[0], src, count);
			if (tmp) {
				(void)hipStreamSynchronize(st);
				(void)hipFree(tmp);
			}
		}
		if (!bad && hipStreamSynchronize(st) != hipSuccess) bad = g->fail("reduce by copies: stream synchronisation failed") != 0;
	}
	return g->barrier->wait(bad);
}

// contiguous blocks of genomes balanced by length
std::vector<size_t> split_by_length(const std::vector<uint64_t> &len, size_t world)
{
	const size_t n = len.size();
	double tot = 0;
	for (uint64_t l : len) tot += (double)l;
	if (tot <= 0) tot = 1;
	std::vector<size_t> b(1, 0);
	double acc = 0;
	size_t r = 1;
	for (size_t j = 0; j < n; j++) {
		acc += (double)len[j];
		while (r < world && acc >= tot * (double)r / (double)world) {
			b.push_back(j + 1);
			r++;
		}
	}
	while (b.size() < world + 1) b.push_back(n);
	b[world] = n;
	return b;
}

} // namespace

extern "C" {

const char *phylo_group_last_error(const phylo_group *g) { return g ? g->err.c_str() : g_group_error.c_str(); }

int phylo_group_create(phylo_group **out, size_t n_ranks, const int *devices)
{
	if (!out) return 1;
	*out = nullptr;
	if (n_ranks == 0 || n_ranks > 64) {
		g_group_error = "a group has 1 to 64 ranks";
		return 1;
	}
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
		g_group_error = "no usable HIP device";
		return 2;
	}
#ifdef PHY_DEV_HOOKS
	if (getenv("PHY_DEBUG_ABORT")) {
		signal(SIGABRT, abort_trace);
		signal(SIGSEGV, abort_trace);
	}
#endif
	phylo_group *g = new phylo_group();
	g->world = n_ranks;
	for (size_t r = 0; r < n_ranks; r++) g->dev.push_back(devices ? devices[r] : (int)(r % (size_t)count));
	for (int d : g->dev)
		if (d < 0 || d >= count) {
			g_group_error = "device ordinal out of range";
			delete g;
			return 3;
		}
	g->ctx.assign(n_ranks, nullptr);
	g->stream.assign(n_ranks, nullptr);
	g->threads.reset(new RankThreads(n_ranks));
	g->barrier.reset(new Barrier(n_ranks));
	g->own_total.assign(n_ranks, 0);
	g->d_all.assign(n_ranks, nullptr);
	g->d_tri.assign(n_ranks, nullptr);
	g->t_anchor.assign(n_ranks, 0);
	g->t_exchange.assign(n_ranks, 0);
	g->t_compare.assign(n_ranks, 0);
	g->t_reduce.assign(n_ranks, 0);
	// RCCL when every rank has a device of its own (it refuses two ranks on one device)
	std::vector<int> sorted = g->dev;
	std::sort(sorted.begin(), sorted.end());
	const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
	bool copies_only = false;
#ifdef PHY_DEV_HOOKS
	if (const char *force = getenv("PHYLONIUM_AMD_GROUP_BACKEND")) copies_only = !strcmp(force, "copies"); // tests
#endif
	if (n_ranks > 1 && distinct && !copies_only && g->rccl.load()) {
		g->comm.assign(n_ranks, nullptr);
		if (g->rccl.CommInitAll(g->comm.data(), (int)n_ranks, g->dev.data()) == ncclSuccess) g->use_rccl = true;
		else g->comm.clear();
	}
	// contexts and streams, every rank on its own thread (the HIP start-up of the devices runs side by side)
	g->threads->run([&](size_t r) {
		if (phylo_ctx_create(&g->ctx[r], g->dev[r])) {
			g->fail("rank %zu: %s", r, phylo_last_error(nullptr));
			return;
		}
		if (hipSetDevice(g->dev[r]) != hipSuccess || hipStreamCreateWithFlags(&g->stream[r], hipStreamNonBlocking) != hipSuccess ||
			phylo_ctx_set_stream(g->ctx[r], g->stream[r])) {
			g->fail("rank %zu: cannot create its stream", r);
			return;
		}
		// the contexts' host worker pools (sort + chain filter of tie lists, result copies) share the host's cores
		{
			const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
			(void)phylo_set_option(g->ctx[r], "host_threads", (long)std::max<size_t>(2, std::min<size_t>(48, hw) / g->world));
		}
		if (!g->use_rccl) // copies between the ranks' devices
			for (size_t o = 0; o < g->world; o++)
				if (g->dev[o] != g->dev[r]) (void)hipDeviceEnablePeerAccess(g->dev[o], 0);
		(void)hipGetLastError();
	});
	if (g->failed()) {
		g_group_error = g->err;
		phylo_group_destroy(g);
		return 4;
	}
	*out = g;
	return 0;
}

void phylo_group_destroy(phylo_group *g)
{
	if (!g) return;
	if (g->threads) {
		g->threads->run([&](size_t r) {
			(void)hipSetDevice(g->dev[r]);
			if (g->stream[r]) (void)hipStreamSynchronize(g->stream[r]);
			if (g->ctx[r]) {
				(void)phylo_ctx_set_stream(g->ctx[r], nullptr);
				phylo_ctx_destroy(g->ctx[r]);
			}
			if (g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
		});
	}
	g->release_plan();
	if (g->use_rccl)
		for (ncclComm_t c : g->comm)
			if (c) (void)g->rccl.CommDestroy(c);
	g->threads.reset();
	delete g;
}

size_t phylo_group_size(const phylo_group *g) { return g ? g->world : 0; }
size_t phylo_group_rank_begin(const phylo_group *g, size_t rank) { return g && rank < g->bounds.size() ? g->bounds[rank] : 0; }
phylo_ctx *phylo_group_ctx(phylo_group *g, size_t rank) { return g && rank < g->world ? g->ctx[rank] : nullptr; }
const char *phylo_group_backend(const phylo_group *g)
{
	if (!g) return "";
	return g->world == 1 ? "one rank" : g->use_rccl ? "rccl" : "device-to-device copies";
}

int phylo_group_set_option(phylo_group *g, const char *key, long value)
{
	if (!g) return 1;
	g->clear_error();
	if (key && !strcmp(key, "exchange_cap")) { // a host that knows its lists' sizes; a pass that outgrows it is repeated with a plan of its own
		if (value < 0) return g->fail("exchange_cap must be >= 0");
		g->forced_cap = (size_t)value;
		g->plan_valid = false;
		return 0;
	}
	for (size_t r = 0; r < g->world; r++)
		if (phylo_set_option(g->ctx[r], key, value)) return g->fail("rank %zu: %s", r, phylo_last_error(g->ctx[r]));
	return 0;
}

int phylo_group_get_stat(phylo_group *g, size_t rank, const char *key, double *out)
{
	if (!g || rank >= g->world || !key || !out) return 1;
	const std::string k = key;
	if (k == "group:ms_anchor") *out = g->t_anchor[rank];
	else if (k == "group:ms_exchange") *out = g->t_exchange[rank];
	else if (k == "group:ms_compare") *out = g->t_compare[rank];
	else if (k == "group:ms_reduce") *out = g->t_reduce[rank];
	else if (k == "group:replans") *out = (double)g->replans;
	else return phylo_get_stat(g->ctx[rank], key, out);
	return 0;
}

// The genomes as 2-bit codes (phylo_set_genomes_packed's arguments: what phylo_host_read_fasta_packed makes).  Rank r
// uploads the genomes of its block — a 1/world-th of the bytes crosses each GPU's PCIe link — into its block of a
// buffer laid out as the contexts' genome arena, one all-gather fills the other blocks over xGMI, and every context
// installs the whole (phylo_set_genomes_packed_device).
int phylo_group_set_genomes_packed(phylo_group *g, size_t n, const uint32_t *const *q2, const size_t *len, const uint32_t *const *bad_lists,
								   const size_t *nbad)
{
	if (!g) return 1;
	g->clear_error();
	if (n && (!q2 || !len || !bad_lists || !nbad)) return g->fail("null genome arrays");
	const size_t W = g->world;
	g->n = n;
	g->glen.assign(len, len + n);
	g->bounds = split_by_length(g->glen, W);
	g->plan_valid = false;
	g->lists_everywhere = false;
	// layout: rank r's block is [r * cap, (r + 1) * cap) bytes of the arena; inside it 64 bytes, then its genomes, each
	// padded to a multiple of 64 and followed by 64 zero bytes (the rules of phylo_set_genomes_device)
	std::vector<uint64_t> off(n, 0);
	uint64_t cap = 64;
	for (size_t r = 0; r < W; r++) {
		uint64_t at = 64;
		for (size_t j = g->bounds[r]; j < g->bounds[r + 1]; j++) {
			off[j] = at;
			at += (len[j] + 63) / 64 * 64 + 64;
		}
		cap = std::max(cap, at);
	}
	cap = (cap + 1023) / 1024 * 1024;
	for (size_t r = 0; r < W; r++)
		for (size_t j = g->bounds[r]; j < g->bounds[r + 1]; j++) off[j] += r * cap;
	if (W * cap / 16 + 64 >= 0xffffffffull) return g->fail("genome buffer too large for 32-bit word offsets");
	const size_t capw_bytes = (size_t)(cap / 4); // bytes of 2-bit codes per block (16 bases per 4-byte word)
	std::vector<uint64_t> lens64(len, len + n);
	std::vector<void *> d_q2(W, nullptr);
	g->threads->run([&](size_t r) {
		hipStream_t st = g->stream[r];
		bool bad = false;
		if (hipSetDevice(g->dev[r]) != hipSuccess || hipMalloc(&d_q2[r], W * capw_bytes + 512) != hipSuccess) {
			d_q2[r] = nullptr;
			bad = g->fail("rank %zu: out of device memory for the packed genomes", r) != 0;
		} else if (hipMemsetAsync(d_q2[r], 0, W * capw_bytes + 512, st) != hipSuccess) {
			bad = g->fail("rank %zu: memset failed", r) != 0;
		} else {
			for (size_t j = g->bounds[r]; j < g->bounds[r + 1] && !bad; j++)
				if (len[j] && hipMemcpyAsync((char *)d_q2[r] + off[j] / 4, q2[j], (len[j] + 15) / 16 * 4, hipMemcpyHostToDevice, st) != hipSuccess)
					bad = g->fail("rank %zu: upload of genome %zu failed", r, j) != 0;
		}
		if (W > 1) {
			if (all_gather(g, r, d_q2, capw_bytes, bad)) return;
		} else if (bad) {
			return;
		}
		if (phylo_set_genomes_packed_device(g->ctx[r], n, d_q2[r], off.data(), lens64.data(), bad_lists, nbad))
			g->fail("rank %zu: %s", r, phylo_last_error(g->ctx[r]));
	});
	for (size_t r = 0; r < W; r++)
		if (d_q2[r]) {
			(void)hipSetDevice(g->dev[r]);
			(void)hipFree(d_q2[r]);
		}
	return g->failed() ? 1 : 0;
}

// esa ref(subject) on every rank (src/process.cxx:413-417): each GPU builds its own index — the suffix array on the
// device by default, or from the caller's array
int phylo_group_set_reference(phylo_group *g, size_t ref_idx, const int64_t *sa, size_t threshold)
{
	if (!g) return 1;
	g->clear_error();
	g->lists_everywhere = false;
	g->plan_valid = false; // another subject, other lists: the next pass sizes the exchange blocks anew
	g->threads->run([&](size_t r) {
		if (phylo_set_reference(g->ctx[r], ref_idx, sa, threshold)) g->fail("rank %zu: %s", r, phylo_last_error(g->ctx[r]));
	});
	return g->failed() ? 1 : 0;
}

// Phase A sharded over the ranks by query block (the loop at src/process.cxx:433-434), then the lists to every rank.
int phylo_group_anchor(phylo_group *g)
{
	if (!g) return 1;
	g->clear_error();
	const size_t W = g->world, n = g->n;
	g->lists_everywhere = false;
	if (g->bounds.size() != W + 1) return g->fail("phylo_group_anchor: no genomes set");
	g->threads->run([&](size_t r) {
		const size_t qb = g->bounds[r], qe = g->bounds[r + 1];
		bool bad = false;
		const double t0 = now_ms();
		if (phylo_anchor(g->ctx[r], qb, qe)) bad = g->fail("rank %zu: %s", r, phylo_last_error(g->ctx[r])) != 0;
		const double t1 = now_ms();
		g->t_anchor[r] = t1 - t0;
		g->t_exchange[r] = 0;
		if (W == 1) return;
		if (!g->plan_valid) { // this pass's list lengths size the blocks (the plan is kept while the lists fit)
			std::vector<uint64_t> counts(qe - qb + 1, 0);
			size_t total = 0;
			if (!bad && phylo_export_packed_device(g->ctx[r], qb, qe, nullptr, 0, counts.data(), &total))
				bad = g->fail("rank %zu: %s", r, phylo_last_error(g->ctx[r])) != 0;
			g->own_total[r] = total;
			if (g->barrier->wait(bad)) return;
			if (r == 0) {
				size_t most = 0, mq = 0;
				for (size_t o = 0; o < W; o++) {
					most = std::max(most, g->own_total[o]);
					mq = std::max(mq, g->bounds[o + 1] - g->bounds[o]);
				}
				g->cap = g->forced_cap ? g->forced_cap : most + most / 4 + 64;
				g->maxq = std::max<size_t>(4, (mq + 3) / 4 * 4);
				g->block_bytes = phylo_exchange_block_bytes(g->maxq, g->cap);
			}
			g->barrier->wait();
			if (hipSetDevice(g->dev[r]) != hipSuccess) bad = g->fail("rank %zu: hipSetDevice failed", r) != 0;
			if (g->d_all[r]) (void)hipFree(g->d_all[r]);
			if (g->d_tri[r]) (void)hipFree(g->d_tri[r]);
			g->d_all[r] = nullptr;
			g->d_tri[r] = nullptr;
			const size_t tri_bytes = phylo_triangle_words(n) * 4;
			if (!bad && (hipMalloc(&g->d_all[r], W * g->block_bytes) != hipSuccess || hipMalloc((void **)&g->d_tri[r], tri_bytes) != hipSuccess))
				bad = g->fail("rank %zu: out of device memory for the exchange buffers", r) != 0;
		}
		// own block straight into its place of the gathered buffer; the all-gather fills the rest in place
		if (!bad && phylo_export_block_device(g->ctx[r], qb, qe, (char *)g->d_all[r] + r * g->block_bytes, g->maxq, g->cap))
			bad = g->fail("rank %zu: %s", r, phylo_last_error(g->ctx[r])) != 0;
#ifdef PHY_DEV_HOOKS
		if (!bad) {
			const char *cr = getenv("PHYLONIUM_AMD_TEST_CORRUPT_RANK");
			char *end = nullptr;
			const long want = cr && *cr ? strtol(cr, &end, 10) : -1;
			if (want >= 0 && end && !*end && (size_t)want == r) {
				fprintf(stderr, "[phylonium_amd] development hook: rank %zu sends a damaged record (PHYLONIUM_AMD_TEST_CORRUPT_RANK)\n", r);
				hipLaunchKernelGGL(corrupt_block_kernel, dim3(1), dim3(64), 0, g->stream[r], (uint32_t *)((char *)g->d_all[r] + r * g->block_bytes), (uint32_t)g->maxq);
			}
		}
#endif
		if (all_gather(g, r, g->d_all, g->block_bytes, bad)) return;
		if (phylo_attach_blocks_device(g->ctx[r], g->d_all[r], W, g->bounds.data(), g->maxq, g->cap, qb, qe))
			g->fail("rank %zu: %s", r, phylo_last_error(g->ctx[r]));
		g->t_exchange[r] = now_ms() - t1;
	});
	if (g->failed()) {
		g->plan_valid = false;
		return 1;
	}
	g->plan_valid = W > 1;
	g->lists_everywhere = true;
	return 0;
}

// Phase B sharded by reference-window range (the pair loop of src/process.cxx:524-529 re-cut so that projection and
// pair kernel both shrink with the ranks): every rank tallies all pairs over its range, one reduce adds the parts on
// rank 0, which writes the two symmetric n x n matrices process() returns.
int phylo_group_compare(phylo_group *g, uint64_t *subst, uint64_t *homologs)
{
	if (!g) return 1;
	g->clear_error();
	if (!subst || !homologs) return g->fail("null output matrix");
	const size_t W = g->world, n = g->n;
	if (W == 1) {
		const double t0 = now_ms();
		const int rc = phylo_compare_all(g->ctx[0], subst, homologs);
		g->t_compare[0] = now_ms() - t0;
		g->t_reduce[0] = 0;
		return rc ? g->fail("%s", phylo_last_error(g->ctx[0])) : 0;
	}
	if (!g->lists_everywhere) return g->fail("phylo_group_compare: call phylo_group_anchor first");
	g->threads->run([&](size_t r) {
		bool bad = false;
		const double t0 = now_ms();
		if (phylo_compare_triangle_device(g->ctx[r], r, W, g->d_tri[r])) bad = g->fail("rank %zu: %s", r, phylo_last_error(g->ctx[r])) != 0;
		const double t1 = now_ms();
		g->t_compare[r] = t1 - t0;
		if (reduce_to_rank0(g, r, g->d_tri, phylo_triangle_words(n), bad)) return; // tallies and the parts' reports alike
		if (r == 0 && phylo_triangle_to_matrices(g->ctx[0], g->d_tri[0], subst, homologs)) g->fail("rank 0: %s", phylo_last_error(g->ctx[0]));
		g->t_reduce[r] = now_ms() - t1;
	});
	if (g->failed()) {
		// lists that outgrew the planned blocks: whoever anchors next (phylo_group_process, or a caller of the two
		// calls that tries again) gets a plan made from that pass's own lists
		if (g->err.find("overflowed") != std::string::npos) {
			g->plan_valid = false;
			g->forced_cap = 0;
		}
		return 1;
	}
	return 0;
}

// process() in one call.  The exchange blocks are sized from an earlier pass's list lengths; should a pass outgrow
// them (every rank sees every block's overflow mark, so all fail together), it is repeated once with a new plan.
int phylo_group_process(phylo_group *g, uint64_t *subst, uint64_t *homologs)
{
	if (!g) return 1;
	for (int attempt = 0; attempt < 2; attempt++) {
		if (phylo_group_anchor(g)) return 1;
		if (!phylo_group_compare(g, subst, homologs)) return 0;
		if (attempt == 0 && g->err.find("overflowed") != std::string::npos) {
			g->plan_valid = false;
			g->replans++;
			continue;
		}
		return 1;
	}
	return 1;
}

} // extern "C"

// group.hip — several GPUs of one node behind one host: phylo_group_* of include/phylonium_amd.h.
//
// process() of /root/reference/src/process.cxx:408-556 shards without a data-path collective inside either phase
// (SURVEY §8e): phase A is independent per query (the OpenMP loop at process.cxx:433-434), phase B per reference
// window of the pileup (the pair loop at process.cxx:524-529, re-cut by window range so that projection and pair
// kernel both shrink with the ranks).  A group is one context (phylo_ctx) and one host thread per rank; the ranks
// meet three times:
//   genomes     every rank uploads its block of the packed genomes, one all-gather leaves all of them on every GPU
//   lists       after phase A: fixed-shape exchange blocks, one all-gather in place (the records never visit the host)
//   tallies     after phase B: the parts' u32 triangles with their report words, one all-reduce; every rank's device
//               then writes ITS rows of the two matrices into the node's page-locked home of the result, over its own
//               PCIe link (phylo_triangle_rows_to_result)
// over RCCL (xGMI) when every rank has a GPU of its own — the library is loaded when a group asks for it, a
// single-GPU host never pays for it — and by device-to-device copies when ranks share a GPU (a test box with one)
// or RCCL is not there.
//
// A rank's pass (phylo_group_process) is ONE queue on the rank's stream and the host waits once, for the rank's rows:
//   phylo_anchor_block_device -> ncclAllGather (in place) -> phylo_attach_blocks_device -> phylo_compare_triangle_device
//   -> ncclAllReduce -> phylo_triangle_rows_to_result
// What a wait in between would have told a rank — a list that needs the host's std::sort (process.cxx:438), exchange
// blocks that overflowed, more '!' than the lists hold — rides in the blocks' headers and the summed triangle's report,
// which every rank reads alike: the pass is then repeated, by all of them, the long way (repeat_how below).  The first
// pass against a reference has no plan yet (the blocks are sized from that pass's own lists) and takes phase A with a
// wait.  Written over the public C ABI and the HIP runtime: it is the host a maintainer would write.
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
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
						   hipStream_t stream);
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count);
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
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclCommCount) CommCount = nullptr;
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
		AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
		CommCount = (decltype(CommCount))dlsym(lib, "ncclCommCount");
		GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
		return CommInitAll && CommDestroy && AllGather && AllReduce && CommCount && GetErrorString;
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
	std::vector<void *> d_all;      // the ranks' exchange blocks, gathered in place (the rank's own block lies in its place)
	std::vector<uint32_t *> d_tri;  // the triangle: the rank's part, summed in place by the all-reduce
	std::vector<uint32_t *> d_part; // copies between ranks only: the rank's part, which the other ranks read while they sum
	std::vector<uint32_t *> d_tmp;  //   and where a part from another device is brought to first
	std::vector<size_t> own_total;
	bool lists_everywhere = false; // the last phylo_group_anchor left every rank with all lists
	size_t replans = 0;            // passes repeated because the lists had outgrown the planned blocks
	size_t forced_cap = 0;         // option "exchange_cap": records per exchange block of the next plan (0: from the lists' lengths)

	// the result's home: one page-locked shared-memory segment every rank's context maps and registers with its device
	// (phylo_result_open); every rank's device writes its rows of both matrices there over its own PCIe link
	bool home_open = false, home_failed = false;
	size_t home_n = 0, home_serial = 0;

	// what this data set needs (kept until the genomes or the reference change): a list with tied starts raises its
	// report on every pass — later passes start on the route that worked
	bool slow_anchor = false, valu_pairs = false;
	long user_pairs_kernel = 0; // option "pairs_kernel" as the host set it
	uint32_t report[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // the last summed triangle's report (every rank read the same)
	bool have_report = false;
	size_t passes_repeated = 0;

	std::vector<double> t_anchor, t_exchange, t_compare, t_reduce, t_queued, t_step; // ms of the last pass, per rank (host side)

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
			for (std::vector<uint32_t *> *v : {&d_tri, &d_part, &d_tmp})
				if (r < v->size() && (*v)[r]) (void)hipFree((*v)[r]);
		}
		d_all.assign(world, nullptr);
		d_tri.assign(world, nullptr);
		d_part.assign(world, nullptr);
		d_tmp.assign(world, nullptr);
		plan_valid = false;
	}
	uint32_t *part_buf(size_t r) { return use_rccl ? d_tri[r] : d_part[r]; } // where rank r's comparison leaves its part
	void new_inputs()
	{
		plan_valid = false;
		lists_everywhere = false;
		slow_anchor = valu_pairs = false;
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

// the sum of every rank's part (count u32: part_buf(r)) in every rank's d_tri[r]; `bad` and the result as above
bool all_reduce(phylo_group *g, size_t r, size_t count, bool bad)
{
	hipStream_t st = g->stream[r];
	if (g->use_rccl) {
		if (g->barrier->wait(bad)) return true;
		const ncclResult_t rc = g->rccl.AllReduce(g->d_tri[r], g->d_tri[r], count, ncclUint32, ncclSum, g->comm[r], st);
		if (rc != ncclSuccess) {
			g->fail("ncclAllReduce: %s", g->rccl.GetErrorString(rc));
			return true;
		}
		return false;
	}
	if (!bad && hipStreamSynchronize(st) != hipSuccess) bad = g->fail("all-reduce by copies: stream synchronisation failed") != 0;
	if (g->barrier->wait(bad)) return true; // every rank's part is complete
	if (count) {
		if (hipMemcpyAsync(g->d_tri[r], g->d_part[r], count * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
			bad = g->fail("all-reduce by copies: device copy failed") != 0;
		for (size_t o = 0; o < g->world && !bad; o++) {
			if (o == r) continue;
			const uint32_t *src = g->d_part[o];
			if (g->dev[o] != g->dev[r]) { // bring it over first
				if (hipMemcpyPeerAsync(g->d_tmp[r], g->dev[r], g->d_part[o], g->dev[o], count * 4, st) != hipSuccess) {
					bad = g->fail("all-reduce by copies: device-to-device copy failed") != 0;
					break;
				}
				src = g->d_tmp[r];
			}
			hipLaunchKernelGGL(add_u32_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, st, g->d_tri[r], src, count);
		}
	}
	if (!bad && hipStreamSynchronize(st) != hipSuccess) bad = g->fail("all-reduce by copies: stream synchronisation failed") != 0;
	return g->barrier->wait(bad); // nobody overwrites a part another rank is still reading
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
	g->d_part.assign(n_ranks, nullptr);
	g->d_tmp.assign(n_ranks, nullptr);
	g->t_anchor.assign(n_ranks, 0);
	g->t_exchange.assign(n_ranks, 0);
	g->t_compare.assign(n_ranks, 0);
	g->t_reduce.assign(n_ranks, 0);
	g->t_queued.assign(n_ranks, 0);
	g->t_step.assign(n_ranks, 0);
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
	if (key && !strcmp(key, "pairs_kernel")) g->user_pairs_kernel = value;
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
	else if (k == "group:ms_queued") *out = g->t_queued[rank];
	else if (k == "group:ms_step") *out = g->t_step[rank];
	else if (k == "group:replans") *out = (double)g->replans;
	else if (k == "group:passes_repeated") *out = (double)g->passes_repeated;
	else if (k == "group:shared_result") *out = g->home_open ? 1.0 : 0.0;
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
	g->new_inputs();
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
	g->new_inputs(); // another subject, other lists: the next pass sizes the exchange blocks anew
	g->threads->run([&](size_t r) {
		if (phylo_set_reference(g->ctx[r], ref_idx, sa, threshold)) g->fail("rank %zu: %s", r, phylo_last_error(g->ctx[r]));
	});
	return g->failed() ? 1 : 0;
}

// ── a pass ──

} // extern "C"

namespace {

enum : unsigned { PASS_A = 1u, PASS_B = 2u };
enum Repeat { REPEAT_NONE, REPEAT_PLAN, REPEAT_ANCHOR, REPEAT_PAIRS, REPEAT_FATAL };

// the summed triangle's report (include/phylonium_amd.h: phylo_compare_triangle_device), in the order every rank reads it
Repeat repeat_how(const uint32_t rep[8])
{
	if (rep[2]) return REPEAT_PLAN;   // the gathered lists overflowed their blocks: plan again from this pass's counts
	if (rep[4]) return REPEAT_ANCHOR; // a rank's phase A needs the host (a list with tied starts): phylo_anchor + phylo_export_block_device
	if (rep[0]) return REPEAT_PAIRS;  // more '!' than the lists hold: the vector-ALU pair kernels take the pass
	if (rep[1]) return REPEAT_FATAL;  // a gathered list is not sorted, disjoint and inside the reference
	return REPEAT_NONE;
}
const char *repeat_text(Repeat how)
{
	switch (how) {
		case REPEAT_PLAN: return "the lists gathered from the ranks overflowed their blocks' capacity";
		case REPEAT_ANCHOR: return "a rank's phase A needs the host (a list with tied starts, or scratch that overflowed)";
		case REPEAT_PAIRS: return "more '!' inside homologies than the genomes hold separators: the vector-ALU pair kernels take the pass (pairs_kernel = 1)";
		case REPEAT_FATAL: return "a gathered list is not sorted by projected start, disjoint and inside the reference";
		default: return "";
	}
}
// the library's own words for the same verdicts (the calls that wait for their flags themselves)
Repeat repeat_how(const std::string &e)
{
	if (e.find("overflow") != std::string::npos && e.find("scratch") == std::string::npos) return REPEAT_PLAN;
	if (e.find("needs the host") != std::string::npos) return REPEAT_ANCHOR;
	if (e.find("pairs_kernel = 1") != std::string::npos) return REPEAT_PAIRS;
	return REPEAT_FATAL;
}

// The exchange's shape from this pass's own list lengths (kept while the lists fit: their sizes repeat from pass to pass),
// the buffers, and the result's home.  Inside a rank's job; true when any rank failed.
bool make_plan(phylo_group *g, size_t r, bool bad)
{
	const size_t W = g->world, n = g->n, qb = g->bounds[r], qe = g->bounds[r + 1];
	std::vector<uint64_t> counts(qe - qb + 1, 0);
	size_t total = 0;
	if (!bad && phylo_export_packed_device(g->ctx[r], qb, qe, nullptr, 0, counts.data(), &total))
		bad = g->fail("rank %zu: %s", r, phylo_last_error(g->ctx[r])) != 0;
	g->own_total[r] = total;
	if (g->barrier->wait(bad)) return true;
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
	void **all = &g->d_all[r];
	uint32_t **bufs[3] = {&g->d_tri[r], &g->d_part[r], &g->d_tmp[r]};
	if (*all) (void)hipFree(*all);
	*all = nullptr;
	for (uint32_t **b : bufs) {
		if (*b) (void)hipFree(*b);
		*b = nullptr;
	}
	const size_t tri_bytes = phylo_triangle_words(n) * 4;
	if (!bad && (hipMalloc(all, W * g->block_bytes) != hipSuccess || hipMalloc((void **)&g->d_tri[r], tri_bytes) != hipSuccess ||
				 (!g->use_rccl && (hipMalloc((void **)&g->d_part[r], tri_bytes) != hipSuccess || hipMalloc((void **)&g->d_tmp[r], tri_bytes) != hipSuccess))))
		bad = g->fail("rank %zu: out of device memory for the exchange buffers", r) != 0;
	// the result's home (its size depends on n and the number of ranks only: a plan made again keeps it).  Rank 0 creates the
	// segment, the others open it, rank 0 takes the name away once all have; when a rank cannot map or register it, all
	// fall back to rank 0 fetching the result (decided together)
	if (!g->home_failed && (!g->home_open || g->home_n != n)) {
		char name[96];
		snprintf(name, sizeof name, "/phylonium_amd_g%d_%zu", (int)getpid(), g->home_serial);
		bool hbad = false;
		if (r == 0 && phylo_result_open(g->ctx[0], name, 1, n, W)) hbad = true;
		hbad = g->barrier->wait(hbad);
		if (!hbad && r != 0 && phylo_result_open(g->ctx[r], name, 0, n, W)) hbad = true;
		hbad = g->barrier->wait(hbad);
		if (r == 0) (void)phylo_result_unlink(g->ctx[0]);
		if (hbad) phylo_result_close(g->ctx[r]);
		if (r == 0) {
			g->home_open = !hbad;
			g->home_failed = hbad;
			g->home_n = n;
			g->home_serial++;
		}
	}
	return g->barrier->wait(bad);
}

// One pass of all ranks: phase A + the lists' exchange (PASS_A), phase B + the tallies' sum + the result (PASS_B), or both
// as one queue per rank.  subst / homologs: where the result is wanted (the group's own home of the result: no copy).
int run_pass(phylo_group *g, unsigned what, uint64_t *subst, uint64_t *homologs)
{
	const size_t W = g->world, n = g->n;
	g->have_report = false;
	uint64_t *home_s = nullptr, *home_h = nullptr;
	g->threads->run([&](size_t r) {
		phylo_ctx *ctx = g->ctx[r];
		const size_t qb = g->bounds[r], qe = g->bounds[r + 1];
		bool bad = false;
		const double t0 = now_ms();
		double t1 = t0;
		if (what & PASS_A) {
			// queued: phase A with its exchange block written behind it, nothing waited for — when a plan exists, phase B follows
			// in this pass (its report brings back what the wait would have told), and this data set does not need the host
			const bool queued = (what & PASS_B) && g->plan_valid && !g->slow_anchor;
			if (queued) {
				if (phylo_anchor_block_device(ctx, qb, qe, (char *)g->d_all[r] + r * g->block_bytes, g->maxq, g->cap))
					bad = g->fail("rank %zu: %s", r, phylo_last_error(ctx)) != 0;
			} else {
				if (phylo_anchor(ctx, qb, qe)) bad = g->fail("rank %zu: %s", r, phylo_last_error(ctx)) != 0;
				if (!g->plan_valid && make_plan(g, r, bad)) return;
				// own block straight into its place of the gathered buffer; the all-gather fills the rest in place
				if (!bad && phylo_export_block_device(ctx, qb, qe, (char *)g->d_all[r] + r * g->block_bytes, g->maxq, g->cap))
					bad = g->fail("rank %zu: %s", r, phylo_last_error(ctx)) != 0;
			}
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
			t1 = now_ms();
			g->t_anchor[r] = t1 - t0;
			if (all_gather(g, r, g->d_all, g->block_bytes, bad)) return;
			if (phylo_attach_blocks_device(ctx, g->d_all[r], W, g->bounds.data(), g->maxq, g->cap, qb, qe))
				bad = g->fail("rank %zu: %s", r, phylo_last_error(ctx)) != 0;
			g->t_exchange[r] = now_ms() - t1;
			t1 = now_ms();
		}
		if (what & PASS_B) {
			if (!bad && phylo_compare_triangle_device(ctx, r, W, g->part_buf(r))) bad = g->fail("rank %zu: %s", r, phylo_last_error(ctx)) != 0;
			const double t2 = now_ms();
			g->t_compare[r] = t2 - t1;
			if (all_reduce(g, r, phylo_triangle_words(n), bad)) return; // tallies and the parts' reports alike
			g->t_queued[r] = now_ms() - t0;
			if (g->home_open) {
				// the rank's rows of both matrices, written by its device over its own link; this is the pass's one wait.  Nobody
				// spins on another rank's delivery: the ranks' threads meet at the end of the job.
				uint32_t rep[8] = {0, 0, 0, 0, 0, 0, 0, 0};
				const size_t rb = n * r / W, re = n * (r + 1) / W;
				if (phylo_triangle_rows_to_result(ctx, g->d_tri[r], rb, re, r, 0, rep)) {
					g->fail("rank %zu: %s", r, phylo_last_error(ctx));
					return;
				}
				if (r == 0) {
					memcpy(g->report, rep, sizeof rep);
					g->have_report = true;
					(void)phylo_result_matrices(ctx, &home_s, &home_h);
				}
				// a caller with matrices of its own: every rank's thread copies the rows its device has just delivered (a pass
				// that will be repeated — every rank reads the same report — is not copied out)
				if (subst && homologs && repeat_how(rep) == REPEAT_NONE) {
					uint64_t *hs = nullptr, *hh = nullptr;
					if (phylo_result_matrices(ctx, &hs, &hh) == 0 && hs != subst && re > rb) {
						memcpy(subst + rb * n, hs + rb * n, (re - rb) * n * 8);
						memcpy(homologs + rb * n, hh + rb * n, (re - rb) * n * 8);
					}
				}
			} else if (r == 0) { // no shared home on this node: rank 0 fetches the whole result
				if (!subst || !homologs) g->fail("no shared home of the result on this node: the caller's matrices are needed");
				else if (phylo_triangle_to_matrices(ctx, g->d_tri[0], subst, homologs)) g->fail("rank 0: %s", phylo_last_error(ctx));
			} else if (hipStreamSynchronize(g->stream[r]) != hipSuccess) {
				g->fail("rank %zu: the device failed", r);
			}
			g->t_reduce[r] = now_ms() - t2;
		}
		g->t_step[r] = now_ms() - t0;
	});
	if (g->failed()) return 1;
	if ((what & PASS_B) && g->home_open && (!subst || !homologs) && !home_s) return g->fail("no result matrices");
	return 0;
}

// what a failed or reported pass asks for
Repeat pass_verdict(phylo_group *g, int rc)
{
	if (rc) return repeat_how(g->err);
	if (g->have_report) return repeat_how(g->report);
	return REPEAT_NONE;
}

void set_pairs_kernel(phylo_group *g, long v)
{
	for (size_t r = 0; r < g->world; r++) (void)phylo_set_option(g->ctx[r], "pairs_kernel", v);
}

} // namespace

extern "C" {

// Phase A sharded over the ranks by query block (the loop at src/process.cxx:433-434), then the lists to every rank.
int phylo_group_anchor(phylo_group *g)
{
	if (!g) return 1;
	g->clear_error();
	g->lists_everywhere = false;
	if (g->bounds.size() != g->world + 1) return g->fail("phylo_group_anchor: no genomes set");
	if (g->world == 1) {
		const double t0 = now_ms();
		const int rc = phylo_anchor(g->ctx[0], 0, g->n);
		g->t_anchor[0] = now_ms() - t0;
		g->t_exchange[0] = 0;
		if (rc) return g->fail("%s", phylo_last_error(g->ctx[0]));
		g->lists_everywhere = true;
		return 0;
	}
	if (run_pass(g, PASS_A, nullptr, nullptr)) {
		g->plan_valid = false;
		return 1;
	}
	g->plan_valid = true;
	g->lists_everywhere = true;
	return 0;
}

// Phase B sharded by reference-window range (the pair loop of src/process.cxx:524-529 re-cut so that projection and
// pair kernel both shrink with the ranks): every rank tallies all pairs over its range, one all-reduce adds the parts,
// every rank's device writes its rows of the two symmetric n x n matrices process() returns.
int phylo_group_compare(phylo_group *g, uint64_t *subst, uint64_t *homologs)
{
	if (!g) return 1;
	g->clear_error();
	const size_t W = g->world;
	if (W == 1) {
		if (!subst || !homologs) return g->fail("null output matrix");
		const double t0 = now_ms();
		const int rc = phylo_compare_all(g->ctx[0], subst, homologs);
		g->t_compare[0] = now_ms() - t0;
		g->t_reduce[0] = 0;
		return rc ? g->fail("%s", phylo_last_error(g->ctx[0])) : 0;
	}
	if (!g->lists_everywhere) return g->fail("phylo_group_compare: call phylo_group_anchor first");
	if (!g->home_open && (!subst || !homologs)) return g->fail("null output matrix");
	const int rc = run_pass(g, PASS_B, subst, homologs);
	const Repeat how = pass_verdict(g, rc);
	if (how == REPEAT_NONE) return 0;
	if (!rc) g->fail("%s", repeat_text(how));
	// lists that outgrew the planned blocks: whoever anchors next (phylo_group_process, or a caller of the two calls that
	// tries again) gets a plan made from that pass's own lists
	if (how == REPEAT_PLAN) {
		g->plan_valid = false;
		g->forced_cap = 0;
	}
	return 1;
}

// process() in one call: a rank's pass as one queue (above).  A pass that reports — exchange blocks sized from an earlier
// pass's lists that this pass outgrew, a list that needs the host's std::sort, more '!' than the lists hold — is repeated
// by all ranks the long way; what the data set needed is remembered until the genomes or the reference change.
int phylo_group_process(phylo_group *g, uint64_t *subst, uint64_t *homologs)
{
	if (!g) return 1;
	g->clear_error();
	if (g->bounds.size() != g->world + 1) return g->fail("phylo_group_process: no genomes set");
	if (g->world == 1) {
		if (!subst || !homologs) return g->fail("null output matrix");
		const double t0 = now_ms();
		const int rc = phylo_anchor_compare(g->ctx[0], subst, homologs);
		g->t_step[0] = g->t_queued[0] = now_ms() - t0;
		if (rc) return g->fail("%s", phylo_last_error(g->ctx[0]));
		g->lists_everywhere = true;
		return 0;
	}
	for (int attempt = 0; attempt < 4; attempt++) {
		g->clear_error();
		g->lists_everywhere = false;
		const bool had_plan = g->plan_valid;
		if (g->valu_pairs) set_pairs_kernel(g, 1);
		const int rc = run_pass(g, PASS_A | PASS_B, subst, homologs);
		if (g->valu_pairs) set_pairs_kernel(g, g->user_pairs_kernel);
		const Repeat how = pass_verdict(g, rc);
		if (how == REPEAT_NONE) {
			g->plan_valid = true;
			g->lists_everywhere = true;
			return 0;
		}
		const std::string why = rc ? g->err : std::string(repeat_text(how));
		if (!rc) g->plan_valid = true; // (a pass that ran to its report made, or kept, a plan)
		if (how == REPEAT_PLAN) {
			if (!had_plan && !g->forced_cap) {
				g->clear_error();
				return g->fail("phylo_group_process: the exchange blocks overflowed twice (%s)", why.c_str());
			}
			g->plan_valid = false;
			g->forced_cap = 0;
			g->replans++;
		} else if (how == REPEAT_ANCHOR && !g->slow_anchor) {
			g->slow_anchor = true;
		} else if (how == REPEAT_PAIRS && !g->valu_pairs) {
			g->valu_pairs = true;
		} else {
			if (!rc) g->fail("%s", why.c_str());
			return 1;
		}
		g->passes_repeated++;
	}
	g->clear_error();
	return g->fail("phylo_group_process: the pass was repeated three times without a result");
}

// the two n x n matrices of the result's home (valid after a pass of several ranks on a node that can share the segment;
// handed to phylo_group_process / phylo_group_compare as subst / homologs they mean "leave the result where it is")
int phylo_group_result_matrices(phylo_group *g, uint64_t **subst, uint64_t **homologs)
{
	if (!g || !subst || !homologs) return 1;
	g->clear_error();
	if (g->world == 1 || !g->home_open) return g->fail("phylo_group_result_matrices: this group has no shared home of the result");
	if (phylo_result_matrices(g->ctx[0], subst, homologs)) return g->fail("%s", phylo_last_error(g->ctx[0]));
	return 0;
}

// the ranks RCCL itself counts in rank 0's communicator (0: the ranks do not talk over RCCL)
size_t phylo_group_rccl_ranks(const phylo_group *g)
{
	int count = 0;
	if (!g || !g->use_rccl || g->comm.empty() || g->rccl.CommCount(g->comm[0], &count) != ncclSuccess) return 0;
	return (size_t)count;
}

} // extern "C"

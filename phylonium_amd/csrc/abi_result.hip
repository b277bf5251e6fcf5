// abi_result.hip — where the two N x N tally matrices process() returns (/root/reference/src/process.cxx:519-549: the
// vector<evo_model> filled by the pair loop) land on the host.
//
// The result of a pass is 2 x N x N x 8 bytes (16 MB at N = 1024): over one GPU's PCIe link a third of a millisecond,
// a tenth of a rank's whole step at eight ranks.  So the library owns a page-locked home for it —
//   private   hipHostMalloc'ed, written by this context's device directly (no registration of caller memory, no
//             staging copy): phylo_triangle_to_matrices recognises the pointers phylo_result_matrices hands out;
//   shared    a POSIX shared-memory segment that every rank of the node (one process per GPU, or one thread per GPU)
//             maps and registers: after the all-reduce of the triangles every rank's device writes ITS rows of both
//             matrices over its own PCIe link (phylo_triangle_rows_to_result), and the ranks tell each other through
//             a counter per rank in the segment's header, written after the rank's own stream has been waited for.
#include "abi_ctx.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace phy;
using namespace phyabi;

static const size_t RES_HDR = 4096, RES_MAX_RANKS = 64;
static const uint64_t RES_ABANDONED = ~0ull; // a rank's delivery counter once it has given a pass up

extern "C" {

// rows [row_begin, row_end) of both symmetric matrices from a (summed) triangle: 16-byte stores, a thread per two columns;
// the triangle's report (TRI_TAIL words) goes to tail_out — the rank's slot of the home's header — on the way
__global__ __launch_bounds__(256) void triangle_rows_kernel(uint32_t N, const uint32_t *__restrict__ tri, unsigned long long *__restrict__ s,
															 unsigned long long *__restrict__ h, uint32_t row_begin, uint32_t row_end,
															 uint32_t *__restrict__ tail_out)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, half = ((uint64_t)N + 1) / 2;
	if (t < TRI_TAIL) tail_out[t] = tri[(uint64_t)N * (N - 1) + t];
	if (t >= (uint64_t)(row_end - row_begin) * half) return;
	const uint32_t i = row_begin + (uint32_t)(t / half), j0 = (uint32_t)(t % half) * 2u;
	const uint64_t P = (uint64_t)N * (N - 1) / 2;
	unsigned long long vs[2] = {0, 0}, vh[2] = {0, 0};
#pragma unroll
	for (uint32_t e = 0; e < 2; e++) {
		const uint32_t j = j0 + e;
		if (j < N && j != i) {
			const uint32_t a = i < j ? i : j, b = i < j ? j : i;
			const uint64_t k = (uint64_t)a * (2ull * N - a - 1) / 2 + (b - a - 1);
			vs[e] = tri[k];
			vh[e] = tri[P + k];
		}
	}
	const uint64_t o = (uint64_t)i * N + j0;
	if (j0 + 1 < N && (o & 1) == 0) {
		*(ulonglong2 *)(s + o) = ulonglong2{vs[0], vs[1]};
		*(ulonglong2 *)(h + o) = ulonglong2{vh[0], vh[1]};
	} else {
		s[o] = vs[0];
		h[o] = vh[0];
		if (j0 + 1 < N) {
			s[o + 1] = vs[1];
			h[o + 1] = vh[1];
		}
	}
}

void phylo_result_close(phylo_ctx *c)
{
	if (!c || !c->res.map) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	if (c->res.shared) {
		(void)hipHostUnregister(c->res.map);
		(void)munmap(c->res.map, c->res.bytes);
		if (c->res.creator && c->res.linked) (void)shm_unlink(c->res.name.c_str());
	} else {
		(void)hipHostFree(c->res.map);
	}
	(void)hipGetLastError();
	c->res = phylo_ctx::ResultHome();
}

int phylo_result_open(phylo_ctx *c, const char *shm_name, int create, size_t n, size_t ranks)
{
	if (!c) return 1;
	if (!n || !ranks || ranks > RES_MAX_RANKS) return c->fail("phylo_result_open: 1 to %zu ranks, n > 0", RES_MAX_RANKS);
	phylo_result_close(c);
	HIPOK(c, hipSetDevice(c->device));
	phylo_ctx::ResultHome R;
	R.n = n;
	R.ranks = ranks;
	R.bytes = (RES_HDR + 2 * R.matrix_words() * 8 + 4095) / 4096 * 4096;
	if (shm_name && *shm_name) {
		const int fd = shm_open(shm_name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
		if (fd < 0) return c->fail("phylo_result_open: shm_open(%s) failed", shm_name);
		if (create && ftruncate(fd, (off_t)R.bytes) != 0) {
			close(fd);
			shm_unlink(shm_name);
			return c->fail("phylo_result_open: cannot size the segment (%zu bytes)", R.bytes);
		}
		struct stat sb;
		if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < R.bytes) {
			close(fd);
			return c->fail("phylo_result_open: the segment %s is smaller than %zu genomes need", shm_name, n);
		}
		void *m = mmap(nullptr, R.bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
		close(fd);
		if (m == MAP_FAILED) {
			if (create) shm_unlink(shm_name);
			return c->fail("phylo_result_open: mmap failed");
		}
		void *d = nullptr;
		if (hipHostRegister(m, R.bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess || hipHostGetDevicePointer(&d, m, 0) != hipSuccess || !d) {
			(void)hipGetLastError();
			munmap(m, R.bytes);
			if (create) shm_unlink(shm_name);
			return c->fail("phylo_result_open: the HIP runtime cannot register the shared segment");
		}
		R.map = m;
		R.dev = d;
		R.shared = true;
		R.creator = create != 0;
		R.linked = create != 0;
		R.name = shm_name;
	} else {
		void *m = nullptr, *d = nullptr;
		HIPOK(c, hipHostMalloc(&m, R.bytes, hipHostMallocMapped | hipHostMallocPortable));
		if (hipHostGetDevicePointer(&d, m, 0) != hipSuccess || !d) {
			(void)hipHostFree(m);
			return c->fail("phylo_result_open: no device address for the page-locked buffer");
		}
		memset(m, 0, RES_HDR);
		R.map = m;
		R.dev = d;
	}
	c->res = R;
	return 0;
}

// the creator, once every rank has opened the segment: the name goes, the memory stays until the last mapping does
int phylo_result_unlink(phylo_ctx *c)
{
	if (!c) return 1;
	if (c->res.shared && c->res.creator && c->res.linked) {
		(void)shm_unlink(c->res.name.c_str());
		c->res.linked = false;
	}
	return 0;
}

// A rank that fails between the all-reduce and its delivery says so in its slot of the header: the ranks waiting for it in
// phylo_triangle_rows_to_result return with an error at once instead of after their time-out.  The segment is of no use
// afterwards (every later wait for this rank fails the same way): the ranks open a new one.
int phylo_result_abandon(phylo_ctx *c, size_t rank)
{
	if (!c) return 1;
	if (!c->res.map || rank >= c->res.ranks) return 0;
	__atomic_store_n((uint64_t *)((char *)c->res.map + 64 * rank), RES_ABANDONED, __ATOMIC_RELEASE);
	return 0;
}

int phylo_result_matrices(phylo_ctx *c, uint64_t **subst, uint64_t **homologs)
{
	if (!c || !subst || !homologs) return 1;
	if (!c->res.map || c->res.n != c->n) return c->fail("phylo_result_matrices: no result home for %zu genomes (phylo_result_open)", c->n);
	*subst = c->res.subst();
	*homologs = c->res.homologs();
	return 0;
}

// Queues the rows [row_begin, row_end) of both matrices behind whatever made dev_tri (the all-reduce of the parts'
// triangles), waits for this context's stream, says so in the segment's header, and — wait_ranks > 0 — waits until ranks
// 0 .. wait_ranks - 1 have said the same for this delivery.  report: the TRI_TAIL words behind the triangle.
int phylo_triangle_rows_to_result(phylo_ctx *c, const uint32_t *dev_tri, size_t row_begin, size_t row_end, size_t rank, size_t wait_ranks,
								  uint32_t *report)
{
	if (!c) return 1;
	const size_t N = c->n;
	if (!dev_tri || row_begin > row_end || row_end > N) return c->fail("phylo_triangle_rows_to_result: bad arguments");
	if (!c->res.map || c->res.n != N) return c->fail("phylo_triangle_rows_to_result: no result home for %zu genomes (phylo_result_open)", N);
	if (rank >= c->res.ranks || wait_ranks > c->res.ranks) return c->fail("phylo_triangle_rows_to_result: rank out of range");
	HIPOK(c, hipSetDevice(c->device));
	const double t0 = now_ms();
	// (the header: a 64-byte slot per rank for its deliveries, and from byte 2048 on 32 bytes per rank for the report it read)
	const uint32_t *tail = (const uint32_t *)((const char *)c->res.map + 2048 + 32 * rank);
	{
		unsigned long long *ds = (unsigned long long *)((char *)c->res.dev + RES_HDR), *dh = ds + c->res.matrix_words();
		const uint64_t threads = std::max<uint64_t>(TRI_TAIL, (uint64_t)(row_end - row_begin) * ((N + 1) / 2));
		hipLaunchKernelGGL(triangle_rows_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->stream, (uint32_t)N, dev_tri, ds, dh,
						   (uint32_t)row_begin, (uint32_t)row_end, (uint32_t *)((char *)c->res.dev + 2048 + 32 * rank));
		HIPOK(c, hipGetLastError());
	}
	if (sync_stream(c)) return 1;
	const uint64_t step = ++c->res.step;
	__atomic_store_n((uint64_t *)((char *)c->res.map + 64 * rank), step, __ATOMIC_RELEASE);
	const double t1 = now_ms();
	for (size_t r = 0; r < wait_ranks; r++) {
		const uint64_t *f = (const uint64_t *)((const char *)c->res.map + 64 * r);
		uint64_t spins = 0;
		uint64_t seen;
		while ((seen = __atomic_load_n(f, __ATOMIC_ACQUIRE)) < step) {
			if ((++spins & 0xfffff) == 0 && now_ms() - t1 > 60000.0) return c->fail("phylo_triangle_rows_to_result: rank %zu has not delivered its rows", r);
			__builtin_ia32_pause();
		}
		if (seen == RES_ABANDONED) return c->fail("phylo_triangle_rows_to_result: rank %zu gave the pass up (phylo_result_abandon)", r);
	}
	if (report) memcpy(report, tail, TRI_TAIL * 4);
	c->stats["ms:result_rows"] += t1 - t0;
	c->stats["ms:result_wait_for_ranks"] += now_ms() - t1;
	if (settle_anchor(c)) return 1; // (a phase A queued by phylo_anchor_block_device: its statistics)
	return 0;
}

} // extern "C"

// seqcmp_bw — achieved bandwidth of the B0 byte kernels (seqcmp / revseqcmp over device-resident strings) through the
// public C ABI (phylo_seqcmp_batch), for profiles/r05_seqcmp_bw.json:
//   long    one segment of 64 MiB per call (the shape of one seqcmp() of a whole genome pair), forward and reverse;
//           the calls rotate over 8 pairs of buffers (1 GiB) so that no call finds its strings in the 256 MB Infinity Cache
//   batch   100,000 segments of 0.1-10 kbp (mean ~2.9 kbp: the calls of a pair grid at d = 0.1) between
//           random places of the same buffers, a tenth of them reverse
// The kernel's time is the library's own HIP-event span around the launch (option "profile"); run under
// `rocprofv3 --kernel-trace --stats` for the profiler's figure of the same launches.  Counts are checked on the host for
// one long call and a sample of the batch.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/microbench/seqcmp_bw.hip -L phylonium_amd -lphylonium_amd -Wl,-rpath,$PWD/phylonium_amd -o build/seqcmp_bw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "phylonium_amd.h"

#define CK(x)                                                                          \
	do {                                                                               \
		hipError_t e = (x);                                                            \
		if (e != hipSuccess) {                                                         \
			fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
			exit(1);                                                                   \
		}                                                                              \
	} while (0)
#define PK(ctx, x)                                                               \
	do {                                                                         \
		if (x) {                                                                 \
			fprintf(stderr, "error at line %d: %s\n", __LINE__, phylo_last_error(ctx)); \
			exit(1);                                                             \
		}                                                                        \
	} while (0)

// random ACGT; string 2k + 1 is string 2k with every `every`-th-or-so byte changed
__global__ void fill_kernel(unsigned char *p, size_t n, unsigned long long seed)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull ^ seed;
	x ^= x >> 29;
	x *= 0xBF58476D1CE4E5B9ull;
	x ^= x >> 32;
	p[i] = "ACGT"[x & 3];
}

static double stat(phylo_ctx *c, const char *k)
{
	double v = 0;
	return phylo_get_stat(c, k, &v) ? 0.0 : v;
}

int main(int argc, char **argv)
{
	const size_t LEN = (argc > 1 ? atol(argv[1]) : 64) << 20, PAIRS = 8, REPS = argc > 2 ? atol(argv[2]) : 40;
	const uint64_t ALIGN = argc > 3 ? (uint64_t)atol(argv[3]) : 1; // experiments: the batch's segments start on multiples of this
	phylo_ctx *c = nullptr;
	if (phylo_ctx_create(&c, 0)) {
		fprintf(stderr, "%s\n", phylo_last_error(nullptr));
		return 1;
	}
	const size_t n = 2 * PAIRS, stride = LEN + 64;
	unsigned char *d = nullptr;
	CK(hipMalloc((void **)&d, 64 + n * stride + 512));
	CK(hipMemset(d, 0, 64 + n * stride + 512));
	std::vector<uint64_t> off(n), len(n, LEN);
	for (size_t g = 0; g < n; g++) {
		off[g] = 64 + g * stride;
		hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((LEN + 255) / 256)), dim3(256), 0, 0, d + off[g], LEN, 1234567ull * (g + 1));
	}
	CK(hipDeviceSynchronize());
	PK(c, phylo_set_genomes_device(c, n, d, off.data(), len.data()));
	PK(c, phylo_set_option(c, "profile", 1));

	// host copies of pair 0 for the check
	std::vector<unsigned char> ha(LEN), hb(LEN);
	CK(hipMemcpy(ha.data(), d + off[0], LEN, hipMemcpyDeviceToHost));
	CK(hipMemcpy(hb.data(), d + off[1], LEN, hipMemcpyDeviceToHost));

	printf("[\n");
	for (int rev = 0; rev < 2; rev++) {
		uint64_t want = 0;
		for (size_t i = 0; i < LEN; i++) want += rev ? (((ha[i] ^ hb[LEN - 1 - i]) & 6) != 4) : (ha[i] != hb[i]);
		PK(c, phylo_reset_stats(c));
		bool ok = true;
		for (size_t r = 0; r < REPS + 2; r++) {
			if (r == 2) PK(c, phylo_reset_stats(c)); // two warm-up calls
			const uint32_t ga = (uint32_t)(2 * (r % PAIRS)), gb = ga + 1;
			const uint64_t o = 0, l = LEN;
			const uint8_t rv = (uint8_t)rev;
			uint64_t out = 0;
			PK(c, phylo_seqcmp_batch(c, 1, &ga, &o, &gb, &o, &l, &rv, &out));
			if (ga == 0 && out != want) ok = false;
		}
		const double ms = stat(c, "ms:seqcmp_batch") / stat(c, "n:seqcmp_batch");
		printf(" {\"shape\": \"long\", \"direction\": \"%s\", \"bytes_per_string\": %zu, \"launches\": %.0f, \"kernel_ms\": %.5f, "
			   "\"alg_GBps\": %.1f, \"frac_of_8TBps\": %.4f, \"sites_per_s\": %.4g, \"count_ok\": %s, \"buffers_rotated_GiB\": %.2f},\n",
			   rev ? "revseqcmp" : "seqcmp", LEN, stat(c, "n:seqcmp_batch"), ms, 2.0 * LEN / (ms * 1e-3) / 1e9, 2.0 * LEN / (ms * 1e-3) / 8e12,
			   LEN / (ms * 1e-3), ok ? "true" : "false", (double)(n * LEN) / (1u << 30));
	}
	// the batch three times: a tenth of the segments reverse (what a pair grid holds), all forward, all reverse
	for (int mode = 0; mode < 3; mode++) {
		const size_t NS = 100000;
		std::mt19937_64 rng(99);
		std::vector<uint32_t> ga(NS), gb(NS);
		std::vector<uint64_t> oa(NS), ob(NS), ln(NS), out(NS);
		std::vector<uint8_t> rv(NS);
		double tot = 0;
		for (size_t s = 0; s < NS; s++) {
			// lengths: exponential-ish with mean ~2.9 kbp, clipped to 0.1-10 kbp
			double u = std::generate_canonical<double, 53>(rng);
			uint64_t l = (uint64_t)std::min(10000.0, std::max(100.0, -2900.0 * std::log(1.0 - 0.97 * u)));
			ln[s] = l;
			ga[s] = (uint32_t)(rng() % n);
			gb[s] = (uint32_t)(rng() % n);
			oa[s] = rng() % (LEN - l) / ALIGN * ALIGN;
			ob[s] = rng() % (LEN - l) / ALIGN * ALIGN;
			const bool r10 = (rng() % 10) == 0;
			rv[s] = mode == 0 ? r10 : mode == 2;
			tot += (double)l;
		}
		PK(c, phylo_reset_stats(c));
		for (size_t r = 0; r < REPS / 2 + 2; r++) {
			if (r == 2) PK(c, phylo_reset_stats(c));
			PK(c, phylo_seqcmp_batch(c, NS, ga.data(), oa.data(), gb.data(), ob.data(), ln.data(), rv.data(), out.data()));
		}
		bool ok = true;
		std::vector<unsigned char> x(10000), y(10000);
		for (size_t s = 0; s < NS; s += 997) {
			CK(hipMemcpy(x.data(), d + off[ga[s]] + oa[s], ln[s], hipMemcpyDeviceToHost));
			CK(hipMemcpy(y.data(), d + off[gb[s]] + ob[s], ln[s], hipMemcpyDeviceToHost));
			uint64_t w = 0;
			for (size_t i = 0; i < ln[s]; i++) w += rv[s] ? (((x[i] ^ y[ln[s] - 1 - i]) & 6) != 4) : (x[i] != y[i]);
			if (w != out[s]) ok = false;
		}
		const double ms = stat(c, "ms:seqcmp_batch") / stat(c, "n:seqcmp_batch");
		printf(" {\"shape\": \"batch\", \"segments\": %zu, \"mean_length\": %.0f, \"reverse_share\": %s, \"launches\": %.0f, \"kernel_ms\": %.5f, "
			   "\"alg_GBps\": %.1f, \"frac_of_8TBps\": %.4f, \"sites_per_s\": %.4g, \"count_ok\": %s}%s\n",
			   NS, tot / NS, mode == 0 ? "0.1" : mode == 1 ? "0" : "1", stat(c, "n:seqcmp_batch"), ms, 2.0 * tot / (ms * 1e-3) / 1e9, 2.0 * tot / (ms * 1e-3) / 8e12, tot / (ms * 1e-3),
			   ok ? "true" : "false", mode == 2 ? "\n]" : ",");
	}
	phylo_ctx_destroy(c);
	CK(hipFree(d));
	return 0;
}

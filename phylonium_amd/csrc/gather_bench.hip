// gather_bench.hip — micro-benchmark (not on the product path): the ceiling for what the
// anchor kernel does to memory — uniformly random rows fetched out of a table — as a
// function of the table's size (L2 / Infinity Cache / HBM / translation reach), the row's
// length (8 … 128 bytes, read whole) and the lanes in flight.  Two modes:
//   rate  — every lookup's index comes from a generator in registers: the throughput ceiling;
//   chase — the next index depends on the loaded row (a dependent chain per lane, as a
//           chain step's slot fetch is): the time of one hop under full load.
// usage: gather_bench <table MB> [<table MB> ...]   (one JSON line per size)
// DESIGN.md §3.1; profiles/r01_gather_bench.jsonl (round 1: 1–16 GB, 64/128-byte rows),
// profiles/r04_gather_bench.jsonl (32 MB–16 GB, 8–128-byte rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// ROW bytes per row, read whole with the widest loads (8: one dwordx2; 16 and more: dwordx4s)
template <int ROW, bool CHASE>
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t *__restrict__ table, uint64_t rows, uint32_t iters,
													 uint32_t *__restrict__ out)
{
	uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
	x *= 0x9E3779B97F4A7C15ull;
	uint32_t acc = 0;
	for (uint32_t it = 0; it < iters; it++) {
		x ^= x << 13;
		x ^= x >> 7;
		x ^= x << 17;
		const uint64_t r = (x + (CHASE ? acc : 0u)) % rows;
		const uint8_t *p = table + r * ROW;
		if constexpr (ROW == 8) {
			const uint2 v = *(const uint2 *)p;
			acc ^= v.x ^ v.y;
		} else {
			uint4 v[ROW / 16];
#pragma unroll
			for (int j = 0; j < ROW / 16; j++) v[j] = ((const uint4 *)p)[j];
#pragma unroll
			for (int j = 0; j < ROW / 16; j++) acc ^= v[j].x ^ v[j].w;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// the cache-policy bits of a load (gfx950: sc0, sc1, nt) decide how a miss is fetched: does any of them bring a 16-byte
// row in with less than a 128-byte line, and does the row rate rise with it?  Four independent 16-byte rows a trip.
#define POLICY_LOAD(bits)                                                                                               \
	asm volatile("global_load_dwordx4 %0, %4, off " bits "\n\tglobal_load_dwordx4 %1, %5, off " bits                   \
				 "\n\tglobal_load_dwordx4 %2, %6, off " bits "\n\tglobal_load_dwordx4 %3, %7, off " bits               \
				 "\n\ts_waitcnt vmcnt(0)"                                                                               \
				 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)                                                            \
				 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)                                                                    \
				 : "memory")
template <int POL>
__global__ __launch_bounds__(256) void policy_kernel(const uint8_t *__restrict__ table, uint64_t rows, uint32_t iters,
													 uint32_t *__restrict__ out)
{
	uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
	x *= 0x9E3779B97F4A7C15ull;
	uint32_t acc = 0;
	for (uint32_t it = 0; it < iters; it += 4) {
		const uint8_t *pp[4];
		for (int j = 0; j < 4; j++) {
			x ^= x << 13;
			x ^= x >> 7;
			x ^= x << 17;
			pp[j] = table + (x % rows) * 16;
		}
		const uint8_t *p0 = pp[0], *p1 = pp[1], *p2 = pp[2], *p3 = pp[3];
		uint4 v0, v1, v2, v3;
		if constexpr (POL == 0) POLICY_LOAD("");
		if constexpr (POL == 1) POLICY_LOAD("nt");
		if constexpr (POL == 2) POLICY_LOAD("sc0");
		if constexpr (POL == 3) POLICY_LOAD("sc1");
		if constexpr (POL == 4) POLICY_LOAD("sc0 sc1");
		if constexpr (POL == 5) POLICY_LOAD("sc0 sc1 nt");
		if constexpr (POL == 6) POLICY_LOAD("sc1 nt");
		if constexpr (POL == 7) POLICY_LOAD("sc0 nt");
		acc ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// The same rows through the scalar path: a lane's address goes to SGPRs (readlane), the row comes in by s_load_dwordx4
// — the scalar cache is a 64-byte client of the L2: does a miss then cost a 64-byte fetch, and what is the rate?
__global__ __launch_bounds__(256) void scalar_kernel(const uint8_t *__restrict__ table, uint64_t rows, uint32_t iters,
													 uint32_t *__restrict__ out)
{
	typedef uint32_t v4u __attribute__((ext_vector_type(4)));
	uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
	x *= 0x9E3779B97F4A7C15ull;
	uint32_t acc = 0;
	for (uint32_t it = 0; it < iters; it++) {
		x ^= x << 13;
		x ^= x >> 7;
		x ^= x << 17;
		const uint64_t off = (x % rows) * 16;
		const uint32_t lo = (uint32_t)off, hi = (uint32_t)(off >> 32);
		uint32_t sacc = 0;
#pragma unroll
		for (int b = 0; b < 4; b++) {
			v4u v[16];
#pragma unroll
			for (int l = 0; l < 16; l++) {
				const uint64_t o = (uint64_t)__builtin_amdgcn_readlane((int)lo, b * 16 + l) |
								   (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, b * 16 + l) << 32;
				v[l] = *(const v4u __attribute__((address_space(4))) *)(uintptr_t)(table + o);
			}
#pragma unroll
			for (int l = 0; l < 16; l++) sacc ^= v[l].x ^ v[l].w;
		}
		acc ^= sacc;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
static float run_scalar(const uint8_t *table, uint64_t bytes, int blocks, uint32_t iters, uint32_t *out)
{
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	float best = 1e30f;
	for (int rep = 0; rep < 4; rep++) {
		hipEventRecord(a, 0);
		hipLaunchKernelGGL(scalar_kernel, dim3(blocks), dim3(256), 0, 0, table, bytes / 16, iters, out);
		hipEventRecord(b, 0);
		hipEventSynchronize(b);
		float ms;
		hipEventElapsedTime(&ms, a, b);
		if (rep && ms < best) best = ms;
	}
	hipEventDestroy(a);
	hipEventDestroy(b);
	return best;
}

template <int POL> static float run_policy(const uint8_t *table, uint64_t bytes, int blocks, uint32_t iters, uint32_t *out)
{
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	float best = 1e30f;
	for (int rep = 0; rep < 4; rep++) {
		hipEventRecord(a, 0);
		hipLaunchKernelGGL((policy_kernel<POL>), dim3(blocks), dim3(256), 0, 0, table, bytes / 16, iters, out);
		hipEventRecord(b, 0);
		hipEventSynchronize(b);
		float ms;
		hipEventElapsedTime(&ms, a, b);
		if (rep && ms < best) best = ms;
	}
	hipEventDestroy(a);
	hipEventDestroy(b);
	return best;
}

template <int ROW, bool CHASE> static float run(const uint8_t *table, uint64_t bytes, int blocks, uint32_t iters, uint32_t *out)
{
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	float best = 1e30f;
	for (int rep = 0; rep < 4; rep++) { // the first pass warms the caches and the translations
		hipEventRecord(a, 0);
		hipLaunchKernelGGL((gather_kernel<ROW, CHASE>), dim3(blocks), dim3(256), 0, 0, table, bytes / ROW, iters, out);
		hipEventRecord(b, 0);
		hipEventSynchronize(b);
		float ms;
		hipEventElapsedTime(&ms, a, b);
		if (rep && ms < best) best = ms;
	}
	hipEventDestroy(a);
	hipEventDestroy(b);
	return best;
}

int main(int argc, char **argv)
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	uint32_t *out;
	hipMalloc((void **)&out, (size_t)cus * 8 * 256 * 4);
	const bool policy = argc > 1 && !strcmp(argv[1], "policy"); // gather_bench policy <MB>...: the cache-policy experiment
	for (int ai = policy ? 2 : 1; ai < argc; ai++) {
		const uint64_t bytes = strtoull(argv[ai], 0, 10) << 20;
		uint8_t *table;
		if (hipMalloc((void **)&table, bytes) != hipSuccess) {
			fprintf(stderr, "hipMalloc(%llu MB) failed\n", (unsigned long long)(bytes >> 20));
			continue;
		}
		hipMemset(table, 1, bytes);
		hipDeviceSynchronize();
		printf("{\"table_MB\": %llu", (unsigned long long)(bytes >> 20));
		if (policy) {
			const int blocks = cus * 4;
			const uint32_t iters = 512;
			const double n = (double)blocks * 256 * iters;
			static const char *names[8] = {"plain", "nt", "sc0", "sc1", "sc0_sc1", "sc0_sc1_nt", "sc1_nt", "sc0_nt"};
#define POL_(P)                                                                                                         \
	{                                                                                                                   \
		const float ms = run_policy<P>(table, bytes, blocks, iters, out);                                               \
		printf(", \"row16_x4_%s_Grows_s\": %.2f", names[P], n / (ms * 1e-3) / 1e9);                                     \
	}
			POL_(0) POL_(1) POL_(2) POL_(3) POL_(4) POL_(5) POL_(6) POL_(7)
			for (int bpc = 2; bpc <= 8; bpc *= 2) {
				const uint32_t it2 = 128;
				const float ms = run_scalar(table, bytes, cus * bpc, it2, out);
				printf(", \"row16_scalar_b%d_Grows_s\": %.2f", bpc, (double)cus * bpc * 256 * it2 / (ms * 1e-3) / 1e9);
			}
			printf("}\n");
			fflush(stdout);
			hipFree(table);
			continue;
		}
		const int bpcs[] = {3, 4, 8};
		for (int bi = 0; bi < 3; bi++) {
			const int bpc = bpcs[bi], blocks = cus * bpc;
			const uint32_t iters = 512;
			const double n = (double)blocks * 256 * iters;
#define RATE(ROW)                                                                                                       \
	{                                                                                                                   \
		const float ms = run<ROW, false>(table, bytes, blocks, iters, out);                                             \
		printf(", \"rate_row%d_b%d_Grows_s\": %.2f", ROW, bpc, n / (ms * 1e-3) / 1e9);                                  \
	}
			RATE(8) RATE(16) RATE(32) RATE(64) RATE(128)
#define CHASE_(ROW)                                                                                                     \
	{                                                                                                                   \
		const float ms = run<ROW, true>(table, bytes, blocks, iters, out);                                              \
		printf(", \"chase_row%d_b%d_us_per_hop\": %.3f, \"chase_row%d_b%d_Grows_s\": %.2f", ROW, bpc, ms * 1e3 / iters, ROW, \
			   bpc, n / (ms * 1e-3) / 1e9);                                                                             \
	}
			if (bpc != 8) { CHASE_(16) CHASE_(64) }
		}
		printf("}\n");
		fflush(stdout);
		hipFree(table);
	}
	return 0;
}

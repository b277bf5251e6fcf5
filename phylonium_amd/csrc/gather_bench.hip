// gather_bench.hip — micro-benchmark (not on the product path): the ceiling for
// what the anchor kernel does to memory, i.e. uniformly random lines fetched out of
// a table far larger than the Infinity Cache (128-byte lines with 5 x 16 bytes read,
// or 64-byte lines read whole), `ILP` independent lookups in flight per lane.  Gives
// the "peak" that the anchor kernel's random-line rate is compared with, and shows
// what that peak depends on — the pages in play (argv[1] = table bytes: 53 G lines/s
// up to ~3 GB, ~20 G/s beyond), not the bytes per line.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

// LINE_U4: line length in 16-byte units (8 = 128 B, 4 = 64 B); READS of them are loaded
template <int ILP, int LINE_U4 = 8, int READS = 5>
__global__ __launch_bounds__(256) void gather_kernel(const uint4 *__restrict__ table, uint64_t lines, uint32_t iters,
													 uint32_t *__restrict__ out)
{
	uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
	x *= 0x9E3779B97F4A7C15ull;
	uint32_t acc = 0;
	for (uint32_t it = 0; it < iters; it++) {
		uint4 v[ILP][READS];
#pragma unroll
		for (int k = 0; k < ILP; k++) {
			x ^= x << 13;
			x ^= x >> 7;
			x ^= x << 17;
			const uint4 *p = table + (x % lines) * LINE_U4;
#pragma unroll
			for (int j = 0; j < READS; j++) v[k][j] = p[j];
		}
#pragma unroll
		for (int k = 0; k < ILP; k++)
#pragma unroll
			for (int j = 0; j < READS; j++) acc ^= v[k][j].x ^ v[k][j].w;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
	uint64_t table_bytes = argc > 1 ? strtoull(argv[1], 0, 10) : (2ull << 30);
	uint64_t lines = table_bytes / 128;
	uint4 *table;
	uint32_t *out;
	hipMalloc((void **)&table, lines * 128);
	hipMemset(table, 1, lines * 128);
	int blocks_per_cu[] = {1, 2, 4, 8};
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	hipMalloc((void **)&out, (size_t)prop.multiProcessorCount * 8 * 256 * 4);
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	printf("{\"table_MB\": %llu", (unsigned long long)(table_bytes >> 20));
	for (int bi = 0; bi < 4; bi++) {
		int blocks = prop.multiProcessorCount * blocks_per_cu[bi];
		uint32_t iters = 256;
		for (int ilp = 1; ilp <= 4; ilp *= 2) {
			float best = 1e30f;
			for (int rep = 0; rep < 3; rep++) {
				hipEventRecord(a, 0);
				if (ilp == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, table, lines, iters, out);
				if (ilp == 2) hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, table, lines, iters, out);
				if (ilp == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(256), 0, 0, table, lines, iters, out);
				hipEventRecord(b, 0);
				hipEventSynchronize(b);
				float ms;
				hipEventElapsedTime(&ms, a, b);
				if (ms < best) best = ms;
			}
			double n = (double)blocks * 256 * iters * ilp;
			printf(", \"b%d_ilp%d_Glines_s\": %.2f", blocks_per_cu[bi], ilp, n / (best * 1e-3) / 1e9);
		}
	}
	// the same with 64-byte lines (4 x 16 bytes read per line): what a slot half the size would cost
	{
		const uint64_t lines64 = table_bytes / 64;
		for (int bi = 2; bi < 4; bi++) {
			int blocks = prop.multiProcessorCount * blocks_per_cu[bi];
			uint32_t iters = 256;
			float best = 1e30f;
			for (int rep = 0; rep < 3; rep++) {
				hipEventRecord(a, 0);
				hipLaunchKernelGGL((gather_kernel<4, 4, 4>), dim3(blocks), dim3(256), 0, 0, table, lines64, iters, out);
				hipEventRecord(b, 0);
				hipEventSynchronize(b);
				float ms;
				hipEventElapsedTime(&ms, a, b);
				if (ms < best) best = ms;
			}
			double n = (double)blocks * 256 * iters * 4;
			printf(", \"line64_b%d_ilp4_Glines_s\": %.2f", blocks_per_cu[bi], n / (best * 1e-3) / 1e9);
		}
	}
	printf("}\n");
	return 0;
}

#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// 16 independent accumulators, ITER x 16 x (ops) VALU instructions per wave
template <int KIND> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters)
{
	uint32_t a[16];
#pragma unroll
	for (int i = 0; i < 16; i++) a[i] = seed * (i + 1) + threadIdx.x;
	uint32_t x = seed ^ threadIdx.x;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 16; i++) {
			if (KIND == 0) a[i] = (a[i] ^ x) + 0x9e3779b9u;           // xor + add
			if (KIND == 1) a[i] += __popc(a[i] ^ x);                   // xor + bcnt(acc)
			if (KIND == 2) { float f = __uint_as_float(a[i]); f = f * 1.0001f + 0.5f; a[i] = __float_as_uint(f); } // fma
			if (KIND == 3) a[i] = (a[i] & x) | (a[i] >> 1);            // and_or / shifts
		}
		x += 0x1234567u;
	}
	uint32_t r = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) r ^= a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND> void run(const char *name, int ops_per_elem)
{
	uint32_t *d;
	const int blocks = 256 * 8, iters = 20000; // 8 blocks of 4 waves per CU = 8 waves per SIMD
	CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 7u, 100);
	CK(hipEventRecord(e0));
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 7u, iters);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	double winst = (double)blocks * 4 * iters * 16 * ops_per_elem;
	printf("%-18s %.2f ms  %.0f G wave-instr/s (counting %d per element)  -> %.2f cycles per wave-instr per SIMD at 2.4 GHz\n", name, ms,
		   winst / (ms * 1e-3) / 1e9, ops_per_elem, 1024 * 2.4e9 / (winst / (ms * 1e-3)));
	CK(hipFree(d));
}
int main()
{
	run<0>("xor+add", 2);
	run<1>("xor+bcnt(acc)", 2);
	run<2>("fma", 1);
	run<3>("and_or(shift)", 2);
	return 0;
}

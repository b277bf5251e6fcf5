// unaligned — what a misaligned 16-byte load per lane costs a streaming kernel on gfx950 (for seqcmp_kernels.hip: a batch's
// segments start anywhere).  Every wavefront reads pieces of 4 KiB (four global_load_dwordx4 per lane, issued together) of
// a 1 GiB buffer, starting `off` bytes into it: off = 0 (16-byte aligned), 4, 8 (dword aligned), 1, 5 (byte aligned);
// then the same bytes fetched as ALIGNED chunks with the neighbour lane's chunk brought in over the cross-lane network
// (ds_bpermute) and the wanted 16 bytes cut out by v_alignbyte.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench/unaligned.hip -o build/unaligned
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                          \
	do {                                                                               \
		hipError_t e = (x);                                                            \
		if (e != hipSuccess) {                                                         \
			fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
			exit(1);                                                                   \
		}                                                                              \
	} while (0)

static __device__ __forceinline__ uint4 ld16(const uint8_t *p)
{
	uint4 v;
	__builtin_memcpy(&v, p, 16);
	return v;
}

template <int MODE> // 0: direct loads at any alignment; 1: aligned loads + neighbour's chunk + alignbyte
__global__ __launch_bounds__(256) void stream_kernel(const uint8_t *__restrict__ buf, size_t npieces, uint32_t off, unsigned long long *__restrict__ out)
{
	const uint32_t lane = threadIdx.x & 63u;
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	uint32_t acc = 0;
	for (size_t g = wave; g < npieces; g += nwaves) {
		uint4 x[4];
		if (MODE == 0) {
#pragma unroll
			for (int u = 0; u < 4; u++) x[u] = ld16(buf + g * 4096 + off + (lane + 64u * u) * 16u);
		} else {
			const uint32_t q = off >> 2 & 3u, r = off & 3u; // (wave-uniform)
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const uint4 lo = ld16(buf + g * 4096 + (off & ~15u) + (lane + 64u * u) * 16u);
				uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, 0, 0, 0, 0};
#pragma unroll
				for (int k = 0; k < 4; k++) w[4 + k] = (uint32_t)__shfl_down((int)w[k], 1, 64); // (lane 63 gets its own: a 64th of the bytes wrong, timing only)
				uint32_t o[4];
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const uint32_t a = q == 0 ? w[k] : q == 1 ? w[k + 1] : q == 2 ? w[k + 2] : w[k + 3];
					const uint32_t b = q == 0 ? w[k + 1] : q == 1 ? w[k + 2] : q == 2 ? w[k + 3] : w[k + 4 > 7 ? 7 : k + 4];
					o[k] = __builtin_amdgcn_alignbyte(b, a, r);
				}
				x[u] = make_uint4(o[0], o[1], o[2], o[3]);
			}
		}
#pragma unroll
		for (int u = 0; u < 4; u++) acc += __popc(x[u].x) + __popc(x[u].y) + __popc(x[u].z) + __popc(x[u].w);
	}
	if (acc == 0xffffffffu) out[0] = acc; // (never: keeps the loads alive)
	if (lane == 0) atomicAdd(&out[1 + (wave & 7)], (unsigned long long)acc);
}

int main(int argc, char **argv)
{
	const size_t BYTES = (size_t)(argc > 1 ? atol(argv[1]) : 1024) << 20;
	const int per_cu = argc > 2 ? atoi(argv[2]) : 8;
	uint8_t *d = nullptr;
	unsigned long long *out = nullptr;
	CK(hipMalloc((void **)&d, BYTES + 8192));
	CK(hipMemset(d, 0x5a, BYTES + 8192));
	CK(hipMalloc((void **)&out, 128));
	CK(hipMemset(out, 0, 128));
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const dim3 grid(prop.multiProcessorCount * per_cu), block(256);
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const size_t npieces = BYTES / 4096;
	printf("[\n");
	for (int mode = 0; mode < 2; mode++)
		for (uint32_t off : {0u, 4u, 8u, 1u, 5u, 13u}) {
			float best = 1e9f;
			for (int rep = 0; rep < 6; rep++) {
				CK(hipEventRecord(e0));
				if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, grid, block, 0, 0, d, npieces, off, out);
				else hipLaunchKernelGGL(stream_kernel<1>, grid, block, 0, 0, d, npieces, off, out);
				CK(hipEventRecord(e1));
				CK(hipEventSynchronize(e1));
				float ms = 0;
				CK(hipEventElapsedTime(&ms, e0, e1));
				if (rep && ms < best) best = ms;
			}
			printf(" {\"mode\": \"%s\", \"offset\": %u, \"MiB\": %zu, \"blocks_per_cu\": %d, \"ms\": %.4f, \"GBps\": %.1f}%s\n",
				   mode == 0 ? "direct" : "aligned + neighbour + alignbyte", off, BYTES >> 20, per_cu, best, (double)BYTES / (best * 1e-3) / 1e9,
				   mode == 1 && off == 13u ? "" : ",");
		}
	printf("]\n");
	return 0;
}

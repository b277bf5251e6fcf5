#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void touch(char *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i * 4096 < n) p[i * 4096] = 1; }
int main(int argc, char **argv)
{
	size_t gb = argc > 1 ? atol(argv[1]) : 17;
	CK(hipSetDevice(0));
	CK(hipFree(0));
	double t0 = now();
	char *d;
	CK(hipMalloc(&d, gb << 30));
	double t1 = now();
	size_t pages = (gb << 30) / 4096;
	hipLaunchKernelGGL(touch, dim3((unsigned)((pages + 255) / 256)), dim3(256), 0, 0, d, gb << 30);
	CK(hipDeviceSynchronize());
	double t2 = now();
	CK(hipMemset(d, 0, gb << 30));
	CK(hipDeviceSynchronize());
	double t3 = now();
	printf("%zu GB: hipMalloc %.3f s, first touch %.3f s, memset %.3f s\n", gb, t1 - t0, t2 - t1, t3 - t2);
	return 0;
}

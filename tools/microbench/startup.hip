// startup — what any HIP program pays on this box before and after its own work: hipInit, a stream, hipMalloc of 1 MB,
// one empty kernel, a wait, then either an orderly exit (runtime teardown) or _exit.  Timed from outside, start to exit,
// beside phylonium-amd (tools/tools_wallclock.py -> profiles/r05_wallclock_*.json): the floor of the wall-clock metric.
// usage: startup [quick]     ("quick": leave through _exit(0) once the kernel has run)
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/startup.hip -o build/startup
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <unistd.h>

__global__ void nothing(int *p)
{
	if (p && threadIdx.x == 1234567) *p = 1;
}

static double now()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
	const double t0 = now();
	if (hipInit(0) != hipSuccess) return 1;
	const double t1 = now();
	int *d = nullptr;
	hipStream_t st;
	if (hipSetDevice(0) != hipSuccess || hipStreamCreate(&st) != hipSuccess || hipMalloc((void **)&d, 1 << 20) != hipSuccess) return 2;
	const double t2 = now();
	hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, st, d);
	if (hipStreamSynchronize(st) != hipSuccess) return 3;
	const double t3 = now();
	fprintf(stderr, "startup: main reached; hipInit %.3f s, device + stream + hipMalloc %.3f s, first kernel %.3f s\n", t1 - t0, t2 - t1, t3 - t2);
	if (argc > 1 && !strcmp(argv[1], "quick")) {
		fflush(stderr);
		_exit(0);
	}
	(void)hipFree(d);
	(void)hipStreamDestroy(st);
	return 0;
}

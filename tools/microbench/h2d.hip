#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char **argv)
{
	size_t n = argc > 1 ? atoi(argv[1]) : 256, sz = argc > 2 ? atol(argv[2]) : 1250000;
	double t0 = now();
	CK(hipSetDevice(0));
	CK(hipFree(0));
	printf("hip init %.3f\n", now() - t0);
	std::vector<char *> buf(n);
	for (auto &b : buf) { b = (char *)malloc(sz); memset(b, 1, sz); }
	char *d;
	CK(hipMalloc(&d, n * sz));
	CK(hipMemset(d, 0, n * sz));
	CK(hipDeviceSynchronize());
	hipStream_t st;
	CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	for (int rep = 0; rep < 2; rep++) {
		t0 = now();
		for (size_t i = 0; i < n; i++) CK(hipMemcpyAsync(d + i * sz, buf[i], sz, hipMemcpyHostToDevice, st));
		CK(hipStreamSynchronize(st));
		double t = now() - t0;
		printf("a. 1 thread pageable async: %.3f s  %.1f GB/s\n", t, n * sz / t / 1e9);
	}
	for (int nt : {4, 8, 16}) {
		t0 = now();
		std::atomic<size_t> next{0};
		std::vector<std::thread> th;
		for (int t = 0; t < nt; t++)
			th.emplace_back([&] {
				hipStream_t s;
				CK(hipSetDevice(0));
				CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
				for (size_t i; (i = next.fetch_add(1)) < n;) CK(hipMemcpyAsync(d + i * sz, buf[i], sz, hipMemcpyHostToDevice, s));
				CK(hipStreamSynchronize(s));
				CK(hipStreamDestroy(s));
			});
		for (auto &t : th) t.join();
		double t = now() - t0;
		printf("b. %d threads pageable async: %.3f s  %.1f GB/s\n", nt, t, n * sz / t / 1e9);
	}
	{
		t0 = now();
		char *pin;
		size_t ring = 64 << 20;
		CK(hipHostMalloc(&pin, ring, hipHostMallocDefault));
		printf("   hipHostMalloc 64 MB %.3f\n", now() - t0);
		t0 = now();
		size_t per = ring / 4, off = 0; // 4 slots
		hipEvent_t ev[4];
		for (auto &ee : ev) CK(hipEventCreateWithFlags(&ee, hipEventDisableTiming));
		size_t slot = 0; bool used[4] = {0,0,0,0};
		for (size_t i = 0; i < n; i++) {
			if (off + sz > per) { CK(hipEventRecord(ev[slot], st)); used[slot] = true; slot = (slot + 1) % 4; off = 0; if (used[slot]) CK(hipEventSynchronize(ev[slot])); }
			memcpy(pin + slot * per + off, buf[i], sz);
			CK(hipMemcpyAsync(d + i * sz, pin + slot * per + off, sz, hipMemcpyHostToDevice, st));
			off += sz;
		}
		CK(hipStreamSynchronize(st));
		double t = now() - t0;
		printf("c. 1 thread memcpy to pinned ring + DMA: %.3f s  %.1f GB/s\n", t, n * sz / t / 1e9);
	}
	{
		t0 = now();
		for (size_t i = 0; i < n; i++) CK(hipHostRegister(buf[i], sz, hipHostRegisterDefault));
		double tr = now() - t0;
		for (size_t i = 0; i < n; i++) CK(hipMemcpyAsync(d + i * sz, buf[i], sz, hipMemcpyHostToDevice, st));
		CK(hipStreamSynchronize(st));
		double t = now() - t0;
		printf("d. register each (%.3f s) + DMA: %.3f s  %.1f GB/s\n", tr, t, n * sz / t / 1e9);
		t0 = now();
		for (size_t i = 0; i < n; i++) CK(hipHostUnregister(buf[i]));
		printf("   unregister %.3f\n", now() - t0);
	}
	{
		char *big = (char *)malloc(n * sz);
		memset(big, 2, n * sz);
		t0 = now();
		CK(hipMemcpy(d, big, n * sz, hipMemcpyHostToDevice));
		double t = now() - t0;
		printf("e. one big pageable hipMemcpy: %.3f s  %.1f GB/s\n", t, n * sz / t / 1e9);
		t0 = now();
		CK(hipHostRegister(big, n * sz, hipHostRegisterDefault));
		double tr = now() - t0;
		CK(hipMemcpy(d, big, n * sz, hipMemcpyHostToDevice));
		t = now() - t0;
		printf("f. register big (%.3f s) + one copy: %.3f s  %.1f GB/s\n", tr, t, n * sz / t / 1e9);
		t0 = now();
		CK(hipMemcpy(d, big, n * sz, hipMemcpyHostToDevice));
		t = now() - t0;
		printf("g. registered big copy again: %.3f s  %.1f GB/s\n", t, n * sz / t / 1e9);
	}
	{
		t0 = now();
		char *pin;
		CK(hipHostMalloc(&pin, n * sz, hipHostMallocDefault));
		printf("h. hipHostMalloc %zu MB %.3f s\n", n * sz >> 20, now() - t0);
		t0 = now();
		memset(pin, 3, n * sz);
		printf("   first touch memset %.3f s\n", now() - t0);
		t0 = now();
		CK(hipMemcpy(d, pin, n * sz, hipMemcpyHostToDevice));
		double t = now() - t0;
		printf("   pinned copy %.3f s %.1f GB/s\n", t, n * sz / t / 1e9);
	}
	return 0;
}

// dev: how long rocPRIM's segmented radix sort takes on the long-list filter's shape (64 lists of ~33 k keys in slots of 2^17)
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
int main(int argc, char **argv)
{
	const unsigned segs = argc > 1 ? atoi(argv[1]) : 64, n = argc > 2 ? atoi(argv[2]) : 33000, slot = 1u << 17, slots = 96;
	const unsigned size = slots * slot;
	std::vector<uint64_t> h((size_t)size, ~0ull);
	std::vector<unsigned> b(slots), e(slots);
	srand(1);
	for (unsigned s = 0; s < slots; s++) {
		b[s] = s * slot;
		e[s] = s < segs ? b[s] + n : b[s];
		for (unsigned t = 0; t < (s < segs ? n : 0); t++) h[(size_t)b[s] + t] = ((uint64_t)(((unsigned)rand() << 8) ^ (unsigned)rand()) << 32) | t;
	}
	uint64_t *din, *dout;
	unsigned *db, *de;
	hipMalloc(&din, (size_t)size * 8);
	hipMalloc(&dout, (size_t)size * 8);
	hipMalloc(&db, slots * 4);
	hipMalloc(&de, slots * 4);
	hipMemcpy(din, h.data(), (size_t)size * 8, hipMemcpyHostToDevice);
	hipMemcpy(db, b.data(), slots * 4, hipMemcpyHostToDevice);
	hipMemcpy(de, e.data(), slots * 4, hipMemcpyHostToDevice);
	size_t bytes = 0;
	rocprim::segmented_radix_sort_keys(nullptr, bytes, din, dout, size, slots, db, de, 32u, 64u);
	void *tmp;
	hipMalloc(&tmp, bytes);
	printf("temporary storage %.1f MB\n", bytes / 1e6);
	hipEvent_t a, c;
	hipEventCreate(&a);
	hipEventCreate(&c);
	for (int rep = 0; rep < 5; rep++) {
		hipEventRecord(a, 0);
		rocprim::segmented_radix_sort_keys(tmp, bytes, din, dout, size, slots, db, de, 32u, 64u, 0);
		hipEventRecord(c, 0);
		hipEventSynchronize(c);
		float ms;
		hipEventElapsedTime(&ms, a, c);
		printf("%u lists of %u keys: %.3f ms\n", segs, n, ms);
	}
	std::vector<uint64_t> o((size_t)n);
	hipMemcpy(o.data(), dout, (size_t)n * 8, hipMemcpyDeviceToHost);
	std::vector<uint64_t> w(h.begin(), h.begin() + n);
	std::stable_sort(w.begin(), w.end(), [](uint64_t x, uint64_t y) { return (x >> 32) < (y >> 32); });
	printf("list 0 %s\n", w == o ? "sorted as std::stable_sort does" : "DIFFERS");
	return 0;
}

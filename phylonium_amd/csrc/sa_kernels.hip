// sa_kernels.hip — the suffix array of S on the device.
//
// Replaces the one call the reference makes into libdivsufsort (/root/reference/src/esa.cxx:74,
// `divsufsort64(S, SA, n)`) for callers that do not bring a suffix array of their own.  The result is
// the suffix array — unique for a given string (unsigned-byte order, a suffix that is a prefix of
// another comes first) — so it is checked against the host builders, entry by entry.
//
// Prefix doubling (Manber & Myers; the "discard what is already unique" refinement of Larsson &
// Sadakane), arranged for a GPU:
//   round 0   every suffix's first 21 bytes as one 63-bit key (S holds six byte values: '!', '#',
//             A, C, G, T — 3 bits each, 0 for "past the end"), one radix sort of (key, position);
//             groups of equal keys get the rank of their first member;
//   round h   only suffixes whose group still has more than one member take part: key =
//             (group's rank, rank of the suffix h bytes further on), one radix sort of the active
//             set, new sub-groups, new ranks; h doubles.
// A genome against its reverse complement shares few long strings, so after round 0 a few per cent
// of the suffixes are still active and the later rounds are small; a string full of repeats (the
// tests' 70 kbp copies, a run of one letter) takes log2(n / 21) rounds over what stays tied.
// Sorting and scans are rocPRIM's (library primitives, as a GEMM would be hipBLASLt's).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.h"

namespace phy {

static const uint32_t SA_H0 = 21; // bytes in the first key

static __device__ __forceinline__ uint32_t sa_code(uint8_t b)
{
	// '!' 0x21, '#' 0x23, 'A' 0x41, 'C' 0x43, 'G' 0x47, 'T' 0x54 -> 1..6 in byte order; anything else 7
	return b == '!' ? 1u : b == '#' ? 2u : b == 'A' ? 3u : b == 'C' ? 4u : b == 'G' ? 5u : b == 'T' ? 6u : 7u;
}

__global__ __launch_bounds__(256) void sa_pack_kernel(const uint8_t *__restrict__ S, uint32_t n, uint64_t *__restrict__ key,
													   uint32_t *__restrict__ val, uint32_t *__restrict__ bad)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint64_t k = 0;
	uint32_t wrong = 0;
#pragma unroll
	for (uint32_t t = 0; t < SA_H0; t++) {
		uint32_t c = 0;
		if (i + t < n) {
			c = sa_code(S[i + t]);
			wrong |= c == 7u;
		}
		k = (k << 3) | c;
	}
	key[i] = k;
	val[i] = i;
	if (wrong) *bad = 1;
}

// head value of position t: its own index (through `pos`, or t itself) where a new group starts, 0 elsewhere
__global__ __launch_bounds__(256) void sa_heads_kernel(const uint64_t *__restrict__ key, uint32_t m, const uint32_t *__restrict__ pos,
														uint32_t *__restrict__ headv)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= m) return;
	const bool head = t == 0 || key[t] != key[t - 1];
	headv[t] = head ? (pos ? pos[t] : t) : 0u;
}

// after the max-scan headv[t] is the rank of t's group (the position of its first member).  Writes the ranks
// where they live (rank_at[position], ISA[suffix]) and flags the members of groups larger than one.
__global__ __launch_bounds__(256) void sa_update_kernel(const uint64_t *__restrict__ key, const uint32_t *__restrict__ suf, uint32_t m,
														 const uint32_t *__restrict__ pos, const uint32_t *__restrict__ rank,
														 uint32_t *__restrict__ SA, uint32_t *__restrict__ rank_at,
														 uint32_t *__restrict__ ISA, uint8_t *__restrict__ active, uint32_t n_all)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= m) return;
	const uint32_t p = pos ? pos[t] : t, s = suf[t];
	// (entries out of range — a sort that went wrong — are left out rather than followed: the array's consumer checks it, index_kernels.hip: lcp_kernel)
	if (p < n_all && s < n_all) {
		SA[p] = s;
		rank_at[p] = rank[t];
		ISA[s] = rank[t];
	}
	const bool head = t == 0 || key[t] != key[t - 1];
	const bool last = t + 1 == m || key[t + 1] != key[t];
	active[t] = !(head && last);
}

// keys of a doubling round for the active positions
__global__ __launch_bounds__(256) void sa_round_keys_kernel(const uint32_t *__restrict__ pos, uint32_t m, const uint32_t *__restrict__ SA,
															 const uint32_t *__restrict__ rank_at, const uint32_t *__restrict__ ISA, uint32_t h,
															 uint32_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= m) return;
	const uint32_t p = pos[t] < n ? pos[t] : 0u, s = SA[p] < n ? SA[p] : 0u; // (see sa_update_kernel)
	const uint64_t next = (uint64_t)s + h;
	const uint32_t k2 = next < n ? ISA[next] + 1u : 0u; // a suffix that ends here is smaller than any that goes on
	key[t] = ((uint64_t)rank_at[p] << 32) | k2;
	val[t] = s;
}

struct SaMax {
	__device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

static uint32_t bits_for(uint64_t v)
{
	uint32_t b = 1;
	while (b < 64 && (v >> b)) b++;
	return b;
}

// Working memory: two key buffers, two value buffers, two position lists, the ranks by position, the ranks by
// suffix, the scan values, the flags, and rocPRIM's own temporary storage (sized for n).
struct SaPlan {
	size_t key_a, key_b, val_a, val_b, pos_a, pos_b, rank_at, isa, headv, active, count, tmp, tmp_bytes, total;
};
static SaPlan sa_plan(uint32_t n)
{
	SaPlan P{};
	size_t off = 0;
	auto take = [&](size_t bytes) {
		const size_t at = off;
		off += (bytes + 255) / 256 * 256;
		return at;
	};
	const size_t N = (size_t)n + 4;
	P.key_a = take(N * 8);
	P.key_b = take(N * 8);
	P.val_a = take(N * 4);
	P.val_b = take(N * 4);
	P.pos_a = take(N * 4);
	P.pos_b = take(N * 4);
	P.rank_at = take(N * 4);
	P.isa = take(N * 4);
	P.headv = take(N * 4);
	P.active = take(N);
	P.count = take(256);
	size_t t_sort = 0, t_scan = 0, t_sel = 0;
	rocprim::double_buffer<uint64_t> kd((uint64_t *)nullptr, (uint64_t *)nullptr);
	rocprim::double_buffer<uint32_t> vd((uint32_t *)nullptr, (uint32_t *)nullptr);
	(void)rocprim::radix_sort_pairs(nullptr, t_sort, kd, vd, (size_t)n, 0u, 64u);
	(void)rocprim::inclusive_scan(nullptr, t_scan, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, SaMax());
	(void)rocprim::select(nullptr, t_sel, (uint32_t *)nullptr, (uint8_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n);
	P.tmp_bytes = std::max(t_sort, std::max(t_scan, t_sel)) + 256;
	P.tmp = take(P.tmp_bytes);
	P.total = off;
	return P;
}
size_t suffix_array_scratch_bytes(uint32_t n) { return sa_plan(n).total; }

#define SA_OK(x)                        \
	do {                                \
		hipError_t e_ = (x);            \
		if (e_ != hipSuccess) return 2; \
	} while (0)

// SA[0..n) of S[0..n) on the device.  Returns 0, 1 when S holds a byte other than ! # A C G T (the caller builds
// the array on the host then), 2 on a HIP error.  *rounds_out: doubling rounds after the first sort.
int device_suffix_array(const uint8_t *S, uint32_t n, uint32_t *SA, void *scratch, uint32_t *rounds_out, hipStream_t st)
{
	if (rounds_out) *rounds_out = 0;
	if (n == 0) return 0;
	const SaPlan P = sa_plan(n);
	uint8_t *base = (uint8_t *)scratch;
	uint64_t *key_a = (uint64_t *)(base + P.key_a), *key_b = (uint64_t *)(base + P.key_b);
	uint32_t *val_a = (uint32_t *)(base + P.val_a), *val_b = (uint32_t *)(base + P.val_b);
	uint32_t *pos_a = (uint32_t *)(base + P.pos_a), *pos_b = (uint32_t *)(base + P.pos_b);
	uint32_t *rank_at = (uint32_t *)(base + P.rank_at), *ISA = (uint32_t *)(base + P.isa), *headv = (uint32_t *)(base + P.headv);
	uint8_t *active = base + P.active;
	uint32_t *count = (uint32_t *)(base + P.count);
	void *tmp = base + P.tmp;
	size_t tmp_bytes = P.tmp_bytes;
	auto grid = [](uint32_t m) { return dim3((m + 255u) / 256u); };

	SA_OK(hipMemsetAsync(count, 0, 8, st));
	hipLaunchKernelGGL(sa_pack_kernel, grid(n), dim3(256), 0, st, S, n, key_a, val_a, count + 1);
	rocprim::double_buffer<uint64_t> kd(key_a, key_b);
	rocprim::double_buffer<uint32_t> vd(val_a, val_b);
	tmp_bytes = P.tmp_bytes;
	SA_OK(rocprim::radix_sort_pairs(tmp, tmp_bytes, kd, vd, (size_t)n, 0u, 3u * SA_H0, st));
	hipLaunchKernelGGL(sa_heads_kernel, grid(n), dim3(256), 0, st, kd.current(), n, (const uint32_t *)nullptr, headv);
	tmp_bytes = P.tmp_bytes;
	SA_OK(rocprim::inclusive_scan(tmp, tmp_bytes, headv, headv, (size_t)n, SaMax(), st));
	hipLaunchKernelGGL(sa_update_kernel, grid(n), dim3(256), 0, st, kd.current(), vd.current(), n, (const uint32_t *)nullptr, headv, SA,
					   rank_at, ISA, active, n);
	tmp_bytes = P.tmp_bytes;
	SA_OK(rocprim::select(tmp, tmp_bytes, rocprim::counting_iterator<uint32_t>(0), active, pos_a, count, (size_t)n, st));
	uint32_t host[2] = {0, 0};
	SA_OK(hipMemcpyAsync(host, count, 8, hipMemcpyDeviceToHost, st));
	SA_OK(hipStreamSynchronize(st));
	if (host[1]) return 1;
	uint32_t m = host[0];
	uint32_t *pos = pos_a, *pos_next = pos_b;
	const uint32_t rank_bits = bits_for((uint64_t)n + 1);
	uint32_t rounds = 0;
	for (uint64_t h = SA_H0; m > 0 && h < n; h *= 2) {
		rounds++;
		hipLaunchKernelGGL(sa_round_keys_kernel, grid(m), dim3(256), 0, st, pos, m, SA, rank_at, ISA, (uint32_t)h, n, key_a, val_a);
		rocprim::double_buffer<uint64_t> kr(key_a, key_b);
		rocprim::double_buffer<uint32_t> vr(val_a, val_b);
		tmp_bytes = P.tmp_bytes;
		// the low half holds a rank + 1 (rank_bits), the high half a rank: the bits between are zero, but a radix
		// pass over them costs as much as any other — sort the low part, then (stable) the high part
		SA_OK(rocprim::radix_sort_pairs(tmp, tmp_bytes, kr, vr, (size_t)m, 0u, rank_bits, st));
		tmp_bytes = P.tmp_bytes;
		SA_OK(rocprim::radix_sort_pairs(tmp, tmp_bytes, kr, vr, (size_t)m, 32u, 32u + rank_bits, st));
		hipLaunchKernelGGL(sa_heads_kernel, grid(m), dim3(256), 0, st, kr.current(), m, pos, headv);
		tmp_bytes = P.tmp_bytes;
		SA_OK(rocprim::inclusive_scan(tmp, tmp_bytes, headv, headv, (size_t)m, SaMax(), st));
		hipLaunchKernelGGL(sa_update_kernel, grid(m), dim3(256), 0, st, kr.current(), vr.current(), m, pos, headv, SA, rank_at, ISA, active, n);
		tmp_bytes = P.tmp_bytes;
		SA_OK(rocprim::select(tmp, tmp_bytes, pos, active, pos_next, count, (size_t)m, st));
		SA_OK(hipMemcpyAsync(host, count, 4, hipMemcpyDeviceToHost, st));
		SA_OK(hipStreamSynchronize(st));
		m = host[0];
		std::swap(pos, pos_next);
	}
	SA_OK(hipGetLastError());
	if (rounds_out) *rounds_out = rounds;
	return 0;
}

} // namespace phy

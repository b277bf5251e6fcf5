// index_kernels.hip — the reference index on the device.  The suffix array comes
// from the caller, from the host cores (north star: "the libdivsufsort ESA build stays
// on the host cores") or from sa_kernels.hip; everything else is built here:
//   S     subject + '#' + reverse complement (esa.cxx:72-75, sequence.cxx:73-103), with the
//         subject's GC count (sequence.cxx:152-165) and, for the 6-mer cache check, the set of
//         bytes seen after every nucleotide string of up to 5 letters
//   LCP   (the reference's init_LCP, /root/reference/src/esa.cxx:305-347) — by direct
//         comparison of neighbouring suffixes, capped; exact values beyond the cap
//         are only needed for repeats >= 64 kbp and then come from the host (Kasai)
//   T     k-mer bucket bounds (replaces the 6-mer interval cache, esa.cxx:90-228)
//   SAX   one 16-byte record per rank (anchor_core.h: sax_record)
// SLOT is assembled from T and SAX by build_slots_kernel in abi_reference.hip.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "anchor_core.h"
#include "kernels.h"

namespace phy {

// S[0..L) = ref, S[L] = '#', S[L+1..2L+1) = reverse complement, zeros behind (total bytes written: ns + 64).
// *gc += bytes of ref that count for gc_content (hostlogic.hpp: the same predicate).
__global__ __launch_bounds__(256) void build_subject_kernel(const uint8_t *__restrict__ ref, uint32_t L, uint8_t *__restrict__ S,
															 unsigned long long *__restrict__ gc)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t ns = 2ull * L + 1;
	uint32_t is_gc = 0;
	if (i < ns + 64) {
		uint8_t out = 0;
		if (i < L) {
			out = ref[i];
			is_gc = (out & 'G' & 'C') == ('G' & 'C');
		} else if (i == L) {
			out = '#';
		} else if (i < ns) {
			const uint8_t c = ref[2ull * L - i];
			out = (c < 'A') ? c : (uint8_t)(c ^ ((c & 2) ? 4 : 21));
		}
		S[i] = out;
	}
	const unsigned long long m = __ballot(is_gc);
	if ((threadIdx.x & 63u) == 0 && m) atomicAdd(gc, (unsigned long long)__popcll(m));
}
void launch_build_subject(const uint8_t *ref, uint32_t L, uint8_t *S, unsigned long long *gc, hipStream_t st)
{
	const uint64_t total = 2ull * L + 1 + 64;
	hipLaunchKernelGGL(build_subject_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, ref, L, S, gc);
}

// For every nucleotide string w of 1..5 letters (entry: NEXT_BASE[|w|] + code of w) the set of "next bytes" seen
// behind its occurrences in S: bits 0..3 a nucleotide A C G T, bit 4 anything else or the end of S.  The
// reference's 6-mer cache can only go wrong (hostlogic.hpp: esa_cache_quirk) below a string all of whose
// occurrences go on with the same byte; when every string that occurs is followed by at least two different
// bytes the exact walk over the suffix array is not needed — the case of every genome beyond a few kbp.
static const uint32_t NEXT_ENTRIES = 4 + 16 + 64 + 256 + 1024;
__global__ __launch_bounds__(256) void next_bytes_kernel(const uint8_t *__restrict__ S, uint32_t n, uint32_t *__restrict__ masks)
{
	__shared__ uint32_t loc[NEXT_ENTRIES];
	for (uint32_t t = threadIdx.x; t < NEXT_ENTRIES; t += blockDim.x) loc[t] = 0;
	__syncthreads();
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t code = 0, base = 0, span = 4;
		for (uint32_t k = 1; k <= 5; k++) {
			const uint32_t v = nuc_code(S[i + k - 1]); // S is followed by zero bytes: 4 past the end
			if (v > 3) break;
			code = (code << 2) | v;
			const uint32_t nx = (i + k < n) ? nuc_code(S[i + k]) : 4u;
			const uint32_t bit = 1u << nx;
			if (!(loc[base + code] & bit)) atomicOr(&loc[base + code], bit);
			base += span;
			span <<= 2;
		}
	}
	__syncthreads();
	for (uint32_t t = threadIdx.x; t < NEXT_ENTRIES; t += blockDim.x)
		if (loc[t]) atomicOr(&masks[t], loc[t]);
}
uint32_t next_bytes_entries() { return NEXT_ENTRIES; }
void launch_next_bytes(const uint8_t *S, uint32_t n, uint32_t *masks, hipStream_t st)
{
	const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 255) / 256, 4096);
	hipLaunchKernelGGL(next_bytes_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, S, n, masks);
}

// LCP[r] = lcp(suffix SA[r-1], suffix SA[r]) for 1 <= r < n, min'ed with `cap`;
// LCP[0] = LCP[n] = 0.  capped[0] counts the ranks that reached the cap.  The same comparison checks the array it is
// given: every entry inside S and every suffix smaller than its successor (unsigned bytes, the shorter one first) — which
// together say "a permutation, sorted", i.e. THE suffix array; capped[1] is raised otherwise, and nothing outside S is
// read whatever the entries hold.  (A pair that reaches the cap is not decided here: repeats of >= 64 kbp.)
__global__ __launch_bounds__(256) void lcp_kernel(const uint8_t *__restrict__ S, const uint32_t *__restrict__ SA,
												   uint32_t n, uint32_t cap, uint32_t *__restrict__ LCP,
												   uint32_t *__restrict__ capped)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n) return;
	uint32_t l = 0;
	if (r >= 1 && r < n) {
		const uint32_t a = SA[r - 1], b = SA[r];
		if (a >= n || b >= n || a == b) {
			capped[1] = 1;
		} else {
			// both suffixes end in S's zero padding at different offsets, and S itself never
			// contains a zero byte, so the scan stops at the shorter suffix's end by itself
			const uint32_t lim = min(cap, n - max(a, b));
			while (l < lim) {
				uint64_t x, y;
				__builtin_memcpy(&x, S + a + l, 8);
				__builtin_memcpy(&y, S + b + l, 8);
				const uint64_t d = x ^ y;
				if (d) {
					l += (uint32_t)(__ffsll((unsigned long long)d) - 1) >> 3;
					break;
				}
				l += 8;
			}
			if (l > lim) l = lim;
			if (l >= cap) atomicAdd(capped, 1u);
			else if (!(S[a + l] < S[b + l])) capped[1] = 1; // (the shorter suffix's next byte is the padding's zero)
		}
	} else if (r == 0 && n && SA[0] >= n) {
		capped[1] = 1;
	}
	LCP[r] = l;
}

// T histogram: every suffix is counted in the first bucket whose k-mer is greater
// than it (hostlogic.hpp: kmer_table).  T must be zeroed; an inclusive scan follows.
__global__ __launch_bounds__(256) void kmer_hist_kernel(const uint8_t *__restrict__ S, uint32_t n, uint32_t k,
														 uint32_t *__restrict__ T)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t code = 0, run = 0;
	for (; run < k; run++) {
		const uint32_t v = (i + run < n) ? nuc_code(S[i + run]) : 4u;
		if (v > 3) break;
		code = (code << 2) | v;
	}
	// run == k: a k-mer, counted one past its own bucket; otherwise the ACGT prefix
	// of `run` bytes followed by a byte < 'A' (or the end): before every k-mer with that prefix
	const uint32_t b = (run == k) ? code + 1u : code << (2u * (k - run));
	atomicAdd(&T[b], 1u);
}

// inclusive scan of `m` uint32 in place: per-block sums, scan of the sums, add back
static const uint32_t SCAN_ELEMS = 2048; // per block of 256 threads

__global__ __launch_bounds__(256) void scan_block_kernel(uint32_t *__restrict__ data, uint64_t m,
														  uint32_t *__restrict__ block_sums)
{
	__shared__ uint32_t wsum[4];
	const uint64_t base = (uint64_t)blockIdx.x * SCAN_ELEMS + (uint64_t)threadIdx.x * 8;
	uint32_t v[8], s = 0;
#pragma unroll
	for (int e = 0; e < 8; e++) {
		v[e] = (base + e < m) ? data[base + e] : 0u;
		s += v[e];
		v[e] = s;
	}
	uint32_t incl = s; // wave-inclusive scan of the per-thread sums
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
		if ((int)lane >= d) incl += t;
	}
	if (lane == 63) wsum[wave] = incl;
	__syncthreads();
	uint32_t off = incl - s;
	for (uint32_t w = 0; w < wave; w++) off += wsum[w];
#pragma unroll
	for (int e = 0; e < 8; e++)
		if (base + e < m) data[base + e] = v[e] + off;
	if (threadIdx.x == 255) block_sums[blockIdx.x] = off + s;
}

__global__ __launch_bounds__(256) void scan_sums_kernel(uint32_t *__restrict__ sums, uint32_t nb)
{
	// one block: serial over chunks of 256 with a running carry (nb is a few thousand)
	__shared__ uint32_t wsum[4];
	__shared__ uint32_t carry_s;
	if (threadIdx.x == 0) carry_s = 0;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	for (uint32_t base = 0; base < nb; base += 256) {
		const uint32_t i = base + threadIdx.x;
		uint32_t x = i < nb ? sums[i] : 0u, incl = x;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
			if ((int)lane >= d) incl += t;
		}
		if (lane == 63) wsum[wave] = incl;
		__syncthreads();
		uint32_t off = carry_s;
		for (uint32_t w = 0; w < wave; w++) off += wsum[w];
		if (i < nb) sums[i] = incl + off; // inclusive
		__syncthreads();
		if (threadIdx.x == 255) carry_s = incl + off;
		__syncthreads();
	}
}

__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t *__restrict__ data, uint64_t m,
														const uint32_t *__restrict__ block_sums)
{
	if (blockIdx.x == 0) return;
	const uint32_t add = block_sums[blockIdx.x - 1];
	const uint64_t base = (uint64_t)blockIdx.x * SCAN_ELEMS + (uint64_t)threadIdx.x * 8;
#pragma unroll
	for (int e = 0; e < 8; e++)
		if (base + e < m) data[base + e] += add;
}

__global__ __launch_bounds__(256) void sax_kernel(const uint8_t *__restrict__ S, const uint32_t *__restrict__ SA,
												   const uint32_t *__restrict__ LCP, uint32_t n, U4 *__restrict__ SAX)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= (uint64_t)n + 4) return;
	U4 rec = {0, 0, 0, 0};
	if (r < n) rec = sax_record(S, SA[r], LCP[r], LCP[r + 1]);
	SAX[r] = rec;
}

void launch_lcp(const uint8_t *S, const uint32_t *SA, uint32_t n, uint32_t cap, uint32_t *LCP, uint32_t *capped,
				hipStream_t st)
{
	uint64_t threads = (uint64_t)n + 1;
	hipLaunchKernelGGL(lcp_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, st, S, SA, n, cap, LCP, capped);
}

// T: 4^k + 1 counts → inclusive prefix sums (T[c] = #suffixes < k-mer c), then 4 pad entries = n
void launch_kmer_table(const uint8_t *S, uint32_t n, uint32_t k, uint32_t *T, uint32_t *scratch_sums, hipStream_t st)
{
	const uint64_t m = ((uint64_t)1 << (2 * k)) + 1;
	(void)hipMemsetAsync(T, 0, (m + 4) * sizeof(uint32_t), st);
	hipLaunchKernelGGL(kmer_hist_kernel, dim3((n + 255) / 256), dim3(256), 0, st, S, n, k, T);
	const uint32_t nb = (uint32_t)((m + SCAN_ELEMS - 1) / SCAN_ELEMS);
	hipLaunchKernelGGL(scan_block_kernel, dim3(nb), dim3(256), 0, st, T, m, scratch_sums);
	hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, st, scratch_sums, nb);
	hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(256), 0, st, T, m, scratch_sums);
}
size_t kmer_table_scratch(uint32_t k) { return (size_t)((((uint64_t)1 << (2 * k)) + 1 + SCAN_ELEMS - 1) / SCAN_ELEMS) + 1; }

void launch_sax(const uint8_t *S, const uint32_t *SA, const uint32_t *LCP, uint32_t n, U4 *SAX, hipStream_t st)
{
	uint64_t threads = (uint64_t)n + 4;
	hipLaunchKernelGGL(sax_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, st, S, SA, LCP, n, SAX);
}

} // namespace phy

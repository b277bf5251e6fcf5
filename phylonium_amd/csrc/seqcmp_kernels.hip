// seqcmp_kernels.hip — batched mismatch counters over device-resident genomes.
//
// Replaces seqcmp / revseqcmp (/root/reference/libs/seqcmp.c:13-28,
// libs/revseqcmp.c:15-30 and their SSE2/AVX2/AVX-512 bodies,
// libs/seqcmp_avx2.c:23-58, libs/revseqcmp_avx2.c:24-46) as called from
// evo_model::account / account_rev (src/evo_model.cxx:53-75).  The CPU bodies
// compare 32 bytes per instruction and popcount a movemask; here a wavefront
// works on pieces of 4 KiB — every lane four 16-byte chunks of either string,
// all eight loads issued before the first is looked at —, counts differing bytes
// with a SWAR mask + v_bcnt, and the 64 lane tallies are summed with a wave
// reduction.  Two ways of dealing the pieces out:
//   per segment   a batch of many short segments (the calls of one pair grid, ~3 kbp each): a wavefront per segment
//   split         fewer segments than wavefronts (one seqcmp() of megabytes): the pieces of all segments, numbered
//                 through, are dealt round-robin over all wavefronts of the launch, which add their tallies up with
//                 one atomic per block and segment
//
// Roofline: HBM streaming, 2 algorithmic bytes per compared site;
// tools/microbench/seqcmp_bw.hip measures it (profiles/r05_seqcmp_bw.json).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace phy {

// number of nonzero bytes in x (any byte values)
static __device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x)
{
	uint32_t t = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
	return (uint32_t)__popc(t);
}

static __device__ __forceinline__ uint32_t diff16(const uint4 &a, const uint4 &b)
{
	return nonzero_bytes(a.x ^ b.x) + nonzero_bytes(a.y ^ b.y) + nonzero_bytes(a.z ^ b.z) +
		   nonzero_bytes(a.w ^ b.w);
}

// is_complement(c,d) = ((c^d)&6)==4  (libs/revseqcmp.h:19-23); counts the failures
static __device__ __forceinline__ uint32_t noncomp4(uint32_t a, uint32_t b)
{
	return nonzero_bytes(((a ^ b) & 0x06060606u) ^ 0x04040404u);
}

static __device__ __forceinline__ uint4 reverse16(const uint4 &v)
{
	return make_uint4(__builtin_bswap32(v.w), __builtin_bswap32(v.z), __builtin_bswap32(v.y),
					  __builtin_bswap32(v.x));
}

static __device__ __forceinline__ uint64_t wave_sum(uint64_t v)
{
	for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, o, 64);
	return v;
}

static __device__ __forceinline__ uint4 ld_chunk(const uint8_t *p)
{
	uint4 v;
	__builtin_memcpy(&v, p, 16); // (any alignment: one global_load_dwordx4)
	return v;
}

// A piece [o, o + m) of a segment of n bytes (m <= SEQCMP_PIECE; a, b: the segment's strings) in two halves: its loads —
// every lane four 16-byte chunks of either string, issued together — and the count over what they brought.
// Forward: a[i] against b[i].  Reverse: a[i] against b[n - 1 - i] — chunk [i, i + 16) of a meets [n - 16 - i, n - i)
// of b, byte-reversed.  The ragged end of the segment's last piece goes a byte per lane.
struct PieceRegs {
	uint4 x[SEQCMP_PIECE / 1024], y[SEQCMP_PIECE / 1024];
};
// U: rounds of 64 chunks (1 KiB) the piece can have — SEQCMP_PIECE / 1024 for any piece; a batch of short segments says how
// many its segment needs (a 2.6 kbp segment: three), so that no round is loaded and counted for nothing
template <int U = SEQCMP_PIECE / 1024>
static __device__ __forceinline__ void piece_load(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint32_t n, uint32_t o, uint32_t m,
												   bool rev, uint32_t lane, PieceRegs &R)
{
	const uint32_t full = m & ~15u;
	// No branch around a load: a chunk beyond the piece's whole ones reads the piece's first chunk again and is left out
	// of the count (a load under a branch makes the compiler wait for every load in flight at each of them).  A piece of
	// fewer than 16 bytes reads up to 15 bytes beyond its strings — inside the padding every buffer these kernels are
	// given has on both sides (include/phylonium_amd.h: phylo_set_genomes_device).
#pragma unroll
	for (int u = 0; u < U; u++) {
		uint32_t i = (lane + 64u * (uint32_t)u) * 16u;
		i = i < full ? i : 0u;
		R.x[u] = ld_chunk(a + o + i);
		R.y[u] = ld_chunk(rev ? b + ((int64_t)n - 16 - (int64_t)(o + i)) : b + o + i); // (signed: a string of fewer than 16 bytes starts before b)
	}
}
// a byte of the reversed string pairs with its complement unless ((c ^ d) & 6) != 4: y = ((c ^ d) & 6) ^ 4 is one of
// 0, 2, 4, 6 in every byte, and y + 0x7e carries into bit 7 exactly when y != 0
static __device__ __forceinline__ uint32_t noncomp4_fast(uint32_t a, uint32_t b)
{
	const uint32_t y = ((a ^ b) & 0x06060606u) ^ 0x04040404u;
	return (uint32_t)__popc((y + 0x7e7e7e7eu) & 0x80808080u);
}
template <int U = SEQCMP_PIECE / 1024>
static __device__ __forceinline__ uint32_t piece_count(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint32_t n, uint32_t o, uint32_t m,
														bool rev, uint32_t lane, const PieceRegs &R)
{
	const uint32_t full = m & ~15u;
	uint32_t cnt = 0;
	if (rev) {
#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t i = (lane + 64u * (uint32_t)u) * 16u;
			const uint4 r = reverse16(R.y[u]);
			const uint32_t d = noncomp4_fast(R.x[u].x, r.x) + noncomp4_fast(R.x[u].y, r.y) + noncomp4_fast(R.x[u].z, r.z) + noncomp4_fast(R.x[u].w, r.w);
			cnt += i < full ? d : 0u;
		}
	} else {
#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t i = (lane + 64u * (uint32_t)u) * 16u;
			const uint32_t d = diff16(R.x[u], R.y[u]);
			cnt += i < full ? d : 0u;
		}
	}
	if (full + lane < m) {
		const uint32_t i = o + full + lane;
		cnt += rev ? ((((uint32_t)a[i] ^ (uint32_t)b[n - 1u - i]) & 6u) != 4u) : (a[i] != b[i]);
	}
	return cnt;
}

__global__ __launch_bounds__(256) void seqcmp_batch_kernel(const uint8_t *__restrict__ base,
															const Segment *__restrict__ segs, uint32_t nseg,
															uint64_t *__restrict__ out)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	for (uint32_t s = wave; s < nseg; s += nwaves) {
		const Segment sg = segs[s];
		const uint8_t *a = base + sg.a;
		const uint8_t *b = base + sg.b;
		const uint32_t n = sg.len;
		const bool rev = sg.rev != 0;
		uint64_t cnt = 0;
		for (uint32_t o = 0; o < n; o += SEQCMP_PIECE) {
			const uint32_t m = n - o < SEQCMP_PIECE ? n - o : SEQCMP_PIECE;
			PieceRegs R;
			switch ((m + 1023u) >> 10) { // (wave-uniform)
				case 1:
					piece_load<1>(a, b, n, o, m, rev, lane, R);
					cnt += piece_count<1>(a, b, n, o, m, rev, lane, R);
					break;
				case 2:
					piece_load<2>(a, b, n, o, m, rev, lane, R);
					cnt += piece_count<2>(a, b, n, o, m, rev, lane, R);
					break;
				case 3:
					piece_load<3>(a, b, n, o, m, rev, lane, R);
					cnt += piece_count<3>(a, b, n, o, m, rev, lane, R);
					break;
				default:
					piece_load<4>(a, b, n, o, m, rev, lane, R);
					cnt += piece_count<4>(a, b, n, o, m, rev, lane, R);
			}
		}
		cnt = wave_sum(cnt);
		if (lane == 0) out[s] = cnt;
	}
}

// piece0[s]: the number of pieces of the segments before s (piece0[nseg] = all pieces); out[] zeroed by the caller.
// Wavefront w takes the pieces w, w + nwaves, ...: neighbouring wavefronts read neighbouring 4 KiB at the same time.
// A wavefront per SIMD: a launch of tens of microseconds is over before thousands of blocks have been dealt out
// (measured, 2 x 64 MiB: 8 blocks per CU 40.4 us, 4: 32.5, 2: 29.4, 1: 27.9 — profiles/r05_seqcmp_bw.json).
struct PieceAt {
	const uint8_t *a, *b;
	uint32_t n, o, m, seg;
	bool rev;
};
// ONE: a batch of one segment (one seqcmp() / revseqcmp() call): it comes with the kernel's arguments, nothing is looked up
template <bool ONE>
__global__ __launch_bounds__(256) void seqcmp_split_kernel(const uint8_t *__restrict__ base, const Segment *__restrict__ segs, uint32_t nseg,
															const uint32_t *__restrict__ piece0, unsigned long long *__restrict__ out, Segment one)
{
	const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t npieces = ONE ? (one.len + SEQCMP_PIECE - 1) / SEQCMP_PIECE : piece0[nseg];
	__shared__ unsigned long long blk_cnt[4];
	__shared__ uint32_t blk_seg[4];
	uint32_t s = 0, s_p0 = 0, s_p1 = ONE ? npieces : 0; // the segment of the piece looked up last and its pieces [s_p0, s_p1)
	bool have = ONE;
	Segment sg = one;
	auto locate = [&](uint32_t g) -> PieceAt { // (g ascends from call to call)
		if (!ONE && (!have || g >= s_p1)) {
			uint32_t lo = have ? s + 1 : 0, hi = nseg; // the last segment with piece0[s] <= g (g < npieces: there is one)
			while (hi - lo > 1) {
				const uint32_t mid = (lo + hi) >> 1;
				if (piece0[mid] <= g) lo = mid;
				else hi = mid;
			}
			s = lo;
			s_p0 = piece0[s];
			s_p1 = piece0[s + 1];
			sg = segs[s];
			have = true;
		}
		const uint32_t o = (g - s_p0) * SEQCMP_PIECE;
		return PieceAt{base + sg.a, base + sg.b, sg.len, o, sg.len - o < SEQCMP_PIECE ? sg.len - o : SEQCMP_PIECE, s, sg.rev != 0};
	};
	uint64_t cnt = 0;
	uint32_t cnt_seg = 0xffffffffu; // the segment the running tally belongs to
	auto account = [&](const PieceAt &p, const PieceRegs &R) {
		if (p.seg != cnt_seg) {
			if (cnt_seg != 0xffffffffu) { // on to another segment: this one's tally leaves
				const uint64_t tot = wave_sum(cnt);
				if (lane == 0 && tot) atomicAdd(&out[cnt_seg], (unsigned long long)tot);
			}
			cnt = 0;
			cnt_seg = p.seg;
		}
		cnt += piece_count(p.a, p.b, p.n, p.o, p.m, p.rev, lane, R);
	};
	// (the next piece's loads issued ahead and waited for by count were measured: 2 x 64 MiB 27.7 us against 28.2, 2 x 256 MiB 92.3
	// against 89.3 — profiles/r05_ab_seqcmp_pipe.txt: a launch this short is ramp and bandwidth, not a piece's latency)
	for (uint32_t g = wave; g < npieces; g += nwaves) {
		PieceRegs R;
		const PieceAt p = locate(g);
		piece_load(p.a, p.b, p.n, p.o, p.m, p.rev, lane, R);
		account(p, R);
	}
	// the block's wavefronts mostly end inside the same segment: one atomic for the four of them
	const uint64_t tot = wave_sum(cnt);
	if (lane == 0) {
		blk_cnt[wib] = cnt_seg != 0xffffffffu ? (unsigned long long)tot : 0ull;
		blk_seg[wib] = cnt_seg;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (uint32_t w = 0; w < 4; w++) {
			if (blk_seg[w] == 0xffffffffu) continue;
			unsigned long long v = blk_cnt[w];
			for (uint32_t u = w + 1; u < 4; u++)
				if (blk_seg[u] == blk_seg[w]) {
					v += blk_cnt[u];
					blk_seg[u] = 0xffffffffu;
				}
			if (v) atomicAdd(&out[blk_seg[w]], v);
		}
	}
}

uint32_t seqcmp_split_waves(int n_cu) { return (uint32_t)n_cu * 8u * 4u; } // (what the per-segment launch has)

void launch_seqcmp_batch(const uint8_t *base, const Segment *segs, uint32_t nseg, const uint32_t *piece0, uint32_t npieces, uint64_t *out,
						 int n_cu, hipStream_t st, const Segment *one)
{
	if (!nseg) return;
	if (piece0) { // split (the caller decided: fewer segments than wavefronts, and more pieces than segments; it has zeroed out[])
		uint32_t per_cu = 1;
#ifdef PHY_DEV_HOOKS
		if (const char *e = getenv("PHY_SEQCMP_BPC")) per_cu = (uint32_t)std::max(1, atoi(e)); // experiments
#endif
		const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)n_cu * per_cu, (npieces + 3u) / 4u));
		unsigned long long *o64 = (unsigned long long *)out;
		if (one) hipLaunchKernelGGL((seqcmp_split_kernel<true>), dim3(blocks), dim3(256), 0, st, base, segs, nseg, piece0, o64, *one);
		else hipLaunchKernelGGL((seqcmp_split_kernel<false>), dim3(blocks), dim3(256), 0, st, base, segs, nseg, piece0, o64, Segment{});
		return;
	}
	const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)n_cu * 8u, (nseg + 3u) / 4u));
	hipLaunchKernelGGL(seqcmp_batch_kernel, dim3(blocks), dim3(256), 0, st, base, segs, nseg, out);
}

} // namespace phy

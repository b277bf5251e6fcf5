// seqcmp_kernels.hip — batched mismatch counters over device-resident genomes.
//
// Replaces seqcmp / revseqcmp (/root/reference/libs/seqcmp.c:13-28,
// libs/revseqcmp.c:15-30 and their SSE2/AVX2/AVX-512 bodies,
// libs/seqcmp_avx2.c:23-58, libs/revseqcmp_avx2.c:24-46) as called from
// evo_model::account / account_rev (src/evo_model.cxx:53-75).  The CPU bodies
// compare 32 bytes per instruction and popcount a movemask; here a wavefront
// works on 4 KiB of either string at a time — every lane four 16-byte chunks of
// each, all eight loads issued before the first is looked at —, counts differing
// bytes with a SWAR mask + v_bcnt, and the 64 lane tallies are summed with a wave
// reduction.  Two ways of dealing the work out:
//   rounds   a batch of segments (the calls of one pair grid, ~3 kbp each, or a handful of long homologies): every
//            segment is cut into rounds of 63 chunks, a small kernel writes a 32-byte descriptor per round, and a
//            wavefront's pass takes FOUR consecutive rounds — of one segment or of four — so that every lane has its
//            eight loads in flight whatever the segments' lengths; the next pass's descriptors are fetched meanwhile.
//            Segments start at any byte: a 16-byte load at a byte-misaligned address costs this chip 13-19 % of a
//            stream's rate, one at any multiple of four nothing (tools/microbench/unaligned.hip) — so every lane loads
//            the four dwords its chunk begins in, takes the fifth from its neighbour over the cross-lane network
//            (lane 63 is there to provide it: 63 chunks a round) and cuts its 16 bytes out with v_alignbyte
//   one      a single seqcmp() / revseqcmp() of megabytes: the segment comes with the kernel's arguments, its 4 KiB
//            pieces are dealt round-robin over a wavefront per SIMD, one atomic per block
//
// Roofline: HBM streaming, 2 algorithmic bytes per compared site;
// tools/microbench/seqcmp_bw.hip measures it (profiles/r06_seqcmp_bw.json).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace phy {

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

// bit 7 of every byte in which the strings differ (seqcmp: the bytes differ, libs/seqcmp.c:13-28)
static __device__ __forceinline__ uint32_t differ4(uint32_t a, uint32_t b)
{
	const uint32_t x = a ^ b;
	return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}
// bit 7 of every byte that is not the other's complement: is_complement(c, d) = ((c ^ d) & 6) == 4 (libs/revseqcmp.h:19-23);
// y = ((c ^ d) & 6) ^ 4 is one of 0, 2, 4, 6 in every byte, and y + 0x7e carries into bit 7 exactly when y != 0
static __device__ __forceinline__ uint32_t noncomp4(uint32_t a, uint32_t b)
{
	const uint32_t y = ((a ^ b) & 0x06060606u) ^ 0x04040404u;
	return (y + 0x7e7e7e7eu) & 0x80808080u;
}
// bit 7 of the first `valid` bytes (0..4 and beyond) of a dword
static __device__ __forceinline__ uint32_t first_bytes(int valid)
{
	return valid >= 4 ? 0x80808080u : valid <= 0 ? 0u : (0x80808080u & ((1u << (8 * valid)) - 1u));
}
// the count over one whole 16-byte chunk of either string; reverse: y holds the 16 bytes of the second string that meet
// x's, last byte first
static __device__ __forceinline__ uint32_t count16(const uint4 &x, const uint4 &y, bool rev)
{
	const uint4 r = reverse16(y);
	const uint32_t f0 = rev ? noncomp4(x.x, r.x) : differ4(x.x, y.x), f1 = rev ? noncomp4(x.y, r.y) : differ4(x.y, y.y);
	const uint32_t f2 = rev ? noncomp4(x.z, r.z) : differ4(x.z, y.z), f3 = rev ? noncomp4(x.w, r.w) : differ4(x.w, y.w);
	return (uint32_t)(__popc(f0) + __popc(f1) + __popc(f2) + __popc(f3));
}
// the same over the chunk's first `valid` bytes (a segment's last chunk; <= 0: none)
static __device__ __forceinline__ uint32_t count16(const uint4 &x, const uint4 &y, bool rev, int valid)
{
	const uint4 r = reverse16(y);
	const uint32_t f0 = rev ? noncomp4(x.x, r.x) : differ4(x.x, y.x), f1 = rev ? noncomp4(x.y, r.y) : differ4(x.y, y.y);
	const uint32_t f2 = rev ? noncomp4(x.z, r.z) : differ4(x.z, y.z), f3 = rev ? noncomp4(x.w, r.w) : differ4(x.w, y.w);
	return (uint32_t)(__popc(f0 & first_bytes(valid)) + __popc(f1 & first_bytes(valid - 4)) + __popc(f2 & first_bytes(valid - 8)) +
					  __popc(f3 & first_bytes(valid - 12)));
}

// ── rounds ──
// A round: up to SEQCMP_ROUND bytes (63 chunks) of a segment.  a: the first string's bytes of the round (an offset into
// the genome buffer); b: the second string's 16 bytes that meet a's first 16 — forward the same offset into the segment,
// reverse the 16 bytes that END where the round's part of the reversed window ends: chunk i of a meets b + 16 i forward
// and b - 16 i reverse.  A segment's last chunk is read whole and the chunk behind it as well (its first dword is the
// last chunk's fifth) — up to 19 bytes beyond the strings' ends (before the second one's begin when reversed), inside
// the padding every buffer these kernels are given has on both sides (include/phylonium_amd.h: phylo_set_genomes_device) —
// and the bytes beyond the segment are masked out: no byte-wise tail, no load under a branch.
struct Round {
	uint64_t a, b;
	uint32_t m;   // bytes (0: an empty round, padding of the last pass)
	uint32_t rev; // 0 seqcmp, 1 revseqcmp
	uint32_t seg; // whose tally it adds to
	uint32_t pad;
};
static_assert(sizeof(Round) == 32, "a round's descriptor is 32 bytes");
static const uint32_t ROUNDS_PER_PASS = 4;
static_assert(SEQCMP_ROUND == 63 * 16, "a round is 63 chunks: lane 63 provides lane 62's fifth dword");

// round0[s]: the rounds of the segments before s (round0[nseg]: all of them); hint[p]: the segment round 4 p lies in (made
// by the host beside round0: no search here); a thread per round of the padded list
__global__ __launch_bounds__(256) void seqcmp_rounds_kernel(const Segment *__restrict__ segs, const uint32_t *__restrict__ round0, const uint32_t *__restrict__ hint,
															 uint32_t nrounds, uint32_t npadded, Round *__restrict__ rounds)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= npadded) return;
	Round r = {64, 64, 0, 0, 0, 0};
	if (t < nrounds) {
		uint32_t s = hint[t / ROUNDS_PER_PASS];
		while (round0[s + 1] <= t) s++; // (at most three segments on, and over the empty ones between)
		const Segment sg = segs[s];
		const uint32_t o = (t - round0[s]) * SEQCMP_ROUND;
		r.a = sg.a + o;
		r.b = sg.rev ? sg.b + sg.len - 16u - o : sg.b + o; // (a segment of fewer than 16 bytes: before its string, inside the padding)
		r.m = sg.len - o < SEQCMP_ROUND ? sg.len - o : SEQCMP_ROUND;
		r.rev = sg.rev;
		r.seg = s;
	}
	rounds[t] = r;
}

// the four dwords a chunk begins in (p: any byte address)
static __device__ __forceinline__ uint4 ld_dwords(const uint8_t *p)
{
	uint4 v;
	// (a multiple of four: one global_load_dwordx4 at full rate; the address space said again — an integer turned pointer would be FLAT)
	__builtin_memcpy(&v, (const uint32_t __attribute__((address_space(1))) *)((uintptr_t)p & ~(uintptr_t)3), 16);
	return v;
}
// the chunk's 16 bytes out of its four dwords and the one behind them; r = the chunk's address modulo 4
static __device__ __forceinline__ uint4 cut16(const uint4 &v, uint32_t next, uint32_t r)
{
	return make_uint4(__builtin_amdgcn_alignbyte(v.y, v.x, r), __builtin_amdgcn_alignbyte(v.z, v.y, r), __builtin_amdgcn_alignbyte(v.w, v.z, r),
					  __builtin_amdgcn_alignbyte(next, v.w, r));
}

// A wavefront's pass: four rounds, lane l the l-th chunk of each.  Descriptors are wave-uniform (scalar loads); the next
// pass's are on their way while this pass's chunks are.  The four tallies travel through one wave reduction as 16-bit
// fields (a round's tally is at most 1008) and leave as up to four atomics of one instruction.
__global__ __launch_bounds__(256) void seqcmp_pass_kernel(const uint8_t *__restrict__ base, const Round *__restrict__ rounds, uint32_t npasses,
														   unsigned long long *__restrict__ out)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	uint32_t p = wave;
	if (p >= npasses) return;
	Round d[ROUNDS_PER_PASS];
#pragma unroll
	for (uint32_t u = 0; u < ROUNDS_PER_PASS; u++) d[u] = rounds[(size_t)p * ROUNDS_PER_PASS + u];
	const uint32_t i = lane * 16u;
	const uint32_t up = ((lane + 1u) & 63u) * 4u, down = ((lane + 63u) & 63u) * 4u; // ds_bpermute addresses of the neighbours
	for (;;) {
		uint4 x[ROUNDS_PER_PASS], y[ROUNDS_PER_PASS];
#pragma unroll
		for (uint32_t u = 0; u < ROUNDS_PER_PASS; u++) {
			// the lane's chunk, or — one chunk beyond the round's bytes — the neighbour's fifth dword; further out the round's
			// first chunk again (no load under a branch).  Reverse: the second string's chunks descend, the dword behind a
			// chunk is the lane BELOW's first, and lane 63 fetches lane 0's: the chunk above the round's first.
			const uint32_t at = i < d[u].m + 16u ? i : 0u;
			x[u] = ld_dwords(base + d[u].a + at);
			const uint8_t *pb = base + d[u].b;
			y[u] = ld_dwords(d[u].rev ? (lane == 63u ? pb + 16 : pb - at) : pb + at);
		}
		const uint32_t pn = p + nwaves;
		const bool more = pn < npasses;
		Round nd[ROUNDS_PER_PASS];
#pragma unroll
		for (uint32_t u = 0; u < ROUNDS_PER_PASS; u++) nd[u] = rounds[(size_t)(more ? pn : p) * ROUNDS_PER_PASS + u];
		uint64_t packed = 0;
#pragma unroll
		for (uint32_t u = 0; u < ROUNDS_PER_PASS; u++) {
			const bool rev = d[u].rev != 0;
			const uint32_t ra = (uint32_t)(uintptr_t)(base + d[u].a) & 3u, rb = (uint32_t)(uintptr_t)(base + d[u].b) & 3u;
			const uint32_t xn = (uint32_t)__builtin_amdgcn_ds_bpermute((int)up, (int)x[u].x);
			const uint32_t yn = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rev ? down : up), (int)y[u].x);
			const uint4 cx = cut16(x[u], xn, ra), cy = cut16(y[u], yn, rb);
			uint32_t cnt;
			if (d[u].m == SEQCMP_ROUND) { // (wave-uniform) a whole round: 63 whole chunks, nothing to mask
				cnt = lane < 63u ? count16(cx, cy, rev) : 0u;
			} else {
				const int valid = (int)d[u].m - (int)i; // > 0 for the lanes that have a chunk of this round (never lane 63)
				cnt = count16(cx, cy, rev, valid > 16 ? 16 : valid);
			}
			packed |= (uint64_t)cnt << (16u * u);
		}
		packed = wave_sum(packed);
		if (lane < ROUNDS_PER_PASS) {
			const uint32_t mine = (uint32_t)(packed >> (16u * lane)) & 0xffffu;
			const uint32_t seg = lane == 0 ? d[0].seg : lane == 1 ? d[1].seg : lane == 2 ? d[2].seg : d[3].seg;
			if (mine) atomicAdd(&out[seg], (unsigned long long)mine);
		}
		if (!more) break;
		p = pn;
#pragma unroll
		for (uint32_t u = 0; u < ROUNDS_PER_PASS; u++) d[u] = nd[u];
	}
}

// ── one long segment ──
// A piece [o, o + m) of the segment (m <= SEQCMP_PIECE): every lane four chunks of either string, issued together.
struct PieceRegs {
	uint4 x[SEQCMP_PIECE / 1024], y[SEQCMP_PIECE / 1024];
};
static __device__ __forceinline__ void piece_load(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint32_t n, uint32_t o, uint32_t m,
												   bool rev, uint32_t lane, PieceRegs &R)
{
#pragma unroll
	for (int u = 0; u < (int)(SEQCMP_PIECE / 1024); u++) {
		uint32_t i = (lane + 64u * (uint32_t)u) * 16u;
		i = i < m ? i : 0u;
		R.x[u] = ld_chunk(a + o + i);
		R.y[u] = ld_chunk(rev ? b + ((int64_t)n - 16 - (int64_t)(o + i)) : b + o + i); // (signed: a string of fewer than 16 bytes starts before b)
	}
}
static __device__ __forceinline__ uint32_t piece_count(uint32_t m, bool rev, uint32_t lane, const PieceRegs &R)
{
	uint32_t cnt = 0;
	if (m == SEQCMP_PIECE) { // (wave-uniform) all but a segment's last piece: whole chunks, nothing to mask
#pragma unroll
		for (int u = 0; u < (int)(SEQCMP_PIECE / 1024); u++) cnt += count16(R.x[u], R.y[u], rev);
		return cnt;
	}
#pragma unroll
	for (int u = 0; u < (int)(SEQCMP_PIECE / 1024); u++) {
		const int valid = (int)m - (int)((lane + 64u * (uint32_t)u) * 16u);
		cnt += count16(R.x[u], R.y[u], rev, valid > 16 ? 16 : valid);
	}
	return cnt;
}

// Wavefront w takes the pieces w, w + nwaves, ...: neighbouring wavefronts read neighbouring 4 KiB at the same time.
// A wavefront per SIMD: a launch of tens of microseconds is over before thousands of blocks have been dealt out
// (measured, 2 x 64 MiB: 8 blocks per CU 40.4 us, 4: 32.5, 2: 29.4, 1: 27.9 — profiles/r05_seqcmp_bw.json; the next
// piece's loads issued ahead and waited for by count: 27.7 us against 28.2, 2 x 256 MiB 92.3 against 89.3 —
// profiles/r05_ab_seqcmp_pipe.txt: a launch this short is ramp and bandwidth, not a piece's latency).
__global__ __launch_bounds__(256) void seqcmp_one_kernel(const uint8_t *__restrict__ base, unsigned long long *__restrict__ out, Segment one)
{
	const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t npieces = (one.len + SEQCMP_PIECE - 1) / SEQCMP_PIECE;
	__shared__ unsigned long long blk_cnt[4];
	const uint8_t *a = base + one.a, *b = base + one.b;
	const bool rev = one.rev != 0;
	uint64_t cnt = 0;
	for (uint32_t g = wave; g < npieces; g += nwaves) {
		PieceRegs R;
		const uint32_t o = g * SEQCMP_PIECE, m = one.len - o < SEQCMP_PIECE ? one.len - o : SEQCMP_PIECE;
		piece_load(a, b, one.len, o, m, rev, lane, R);
		cnt += piece_count(m, rev, lane, R);
	}
	const uint64_t tot = wave_sum(cnt);
	if (lane == 0) blk_cnt[wib] = (unsigned long long)tot;
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned long long v = blk_cnt[0] + blk_cnt[1] + blk_cnt[2] + blk_cnt[3];
		if (v) atomicAdd(&out[0], v);
	}
}

uint32_t seqcmp_rounds_per_pass() { return ROUNDS_PER_PASS; }
size_t seqcmp_rounds_bytes(uint64_t nrounds) { return (size_t)((nrounds + ROUNDS_PER_PASS - 1) / ROUNDS_PER_PASS * ROUNDS_PER_PASS) * sizeof(Round); }

void launch_seqcmp_batch(const uint8_t *base, const Segment *segs, uint32_t nseg, const uint32_t *round0, const uint32_t *hint, uint32_t nrounds,
						 void *rounds, uint64_t *out, int n_cu, hipStream_t st, const Segment *one)
{
	if (!nseg) return;
	unsigned long long *o64 = (unsigned long long *)out;
	if (one) { // (the caller has zeroed out[])
		uint32_t per_cu = 1;
#ifdef PHY_DEV_HOOKS
		if (const char *e = getenv("PHY_SEQCMP_BPC")) per_cu = (uint32_t)std::max(1, atoi(e)); // experiments
#endif
		const uint32_t npieces = (one->len + SEQCMP_PIECE - 1) / SEQCMP_PIECE;
		const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)n_cu * per_cu, (npieces + 3u) / 4u));
		hipLaunchKernelGGL(seqcmp_one_kernel, dim3(blocks), dim3(256), 0, st, base, o64, *one);
		return;
	}
	if (!nrounds) return;
	const uint32_t npasses = (nrounds + ROUNDS_PER_PASS - 1) / ROUNDS_PER_PASS, npadded = npasses * ROUNDS_PER_PASS;
	hipLaunchKernelGGL(seqcmp_rounds_kernel, dim3((npadded + 255) / 256), dim3(256), 0, st, segs, round0, hint, nrounds, npadded, (Round *)rounds);
	uint32_t per_cu = 3; // blocks per CU: a wavefront has eight 16-byte loads per lane in flight; measured on the 100 k x 2.6 kbp batch: 2: 0.110 ms, 3: 0.0968, 4: 0.0993, 6: 0.0989, 8: 0.0999
#ifdef PHY_DEV_HOOKS
	if (const char *e = getenv("PHY_SEQCMP_WPC")) per_cu = (uint32_t)std::max(1, atoi(e)); // experiments
#endif
	const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)n_cu * per_cu, (npasses + 3u) / 4u));
	hipLaunchKernelGGL(seqcmp_pass_kernel, dim3(blocks), dim3(256), 0, st, base, (const Round *)rounds, npasses, o64);
}

} // namespace phy

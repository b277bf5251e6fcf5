// pileup_kernels.hip — phase B on gfx950: the N×N pair grid.
//
// Replaces the OpenMP pair loop, the list×list merge-join and the per-overlap
// account / account_rev calls (/root/reference/src/process.cxx:517-549, 566-611,
// 620-658; src/evo_model.cxx:53-87).
//
// The reference visits, for every pair (i,j), every overlap of a homology of i
// with a homology of j and runs seqcmp/revseqcmp over it: 2 bytes read per
// compared site.  After filter_overlaps_max each genome's homologies are
// disjoint on the reference, so the same tallies are
//     homologs(i,j)      = #reference positions covered by both genomes
//     substitutions(i,j) = #those positions where the two projected bases differ
// (SURVEY §3.4).  So each genome is projected ONCE onto reference coordinates
// as bit planes (32 positions per word):
//     V  covered            N0,N1  base on the forward strand (A0 C1 T2 G3; reverse
//     D  reverse homology          hits store the complement, n^2)
//     B  raw byte is '!'
// and a pair is  both = Vi&Vj;  diff = (N0i^N0j)|(N1i^N1j);  popcount.  '!' needs
// the two extra planes because seqcmp compares bytes ('!' != 'A') while
// revseqcmp's ((c^d)&6)==4 test sees '!' as 'A' (libs/revseqcmp.h:19-23):
//     diff |= ~(Di^Dj) & (Bi^Bj)
// The five-plane variant only runs when some projected position holds '!'.
//
// Layout: plane[w][g] (word-major, genome-minor), so the 64 lanes of a wave
// read 64 genomes' words of one reference window with one coalesced 256-byte
// load, and genome i's word is wave-uniform (scalar load).  HBM traffic is
// 3/8 (5/8) byte per genome per reference position, re-used from L2 across the
// whole pair grid, against 2 bytes per compared site for the reference layout.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace phy {

// movemask of bit `b` of each of the 4 bytes of x → 4 bits
static __device__ __forceinline__ uint32_t gather4(uint32_t x, uint32_t b)
{
	return ((((x >> b) & 0x01010101u) * 0x01020408u) >> 24) & 0xfu;
}

// bit `b` of each of the 32 bytes of (lo16,hi16) → 32 bits, byte 0 → bit 0
static __device__ __forceinline__ uint32_t gather32(const uint4 &lo, const uint4 &hi, uint32_t b)
{
	return gather4(lo.x, b) | (gather4(lo.y, b) << 4) | (gather4(lo.z, b) << 8) | (gather4(lo.w, b) << 12) |
		   (gather4(hi.x, b) << 16) | (gather4(hi.y, b) << 20) | (gather4(hi.z, b) << 24) | (gather4(hi.w, b) << 28);
}

// 1 where the byte equals '!' (0x21): among {A,C,G,T,!} only '!' has bit 5 set and bit 6 clear
static __device__ __forceinline__ uint32_t bang32(const uint4 &lo, const uint4 &hi)
{
	return gather32(lo, hi, 5) & ~gather32(lo, hi, 6);
}

static const uint32_t PROJ_TW = 64; // words per tile
static const uint32_t PROJ_TG = 32; // genomes per tile (LDS: 5*64*33*4 = 42 KB → 3 blocks per CU)

// Projection: one block per tile of 64 reference windows × 32 genomes.
// Reading side: a wavefront takes one genome and its 64 lanes take the 64
// consecutive windows, so the query bytes of a homology are read as contiguous
// 32-byte pieces (2 KiB per wave, coalesced).  Writing side: the tile is
// transposed through LDS so that every plane row [w][g0..g0+31] leaves as one
// full 128-byte line.
// first[g*ntw + tw] = first homology of genome g that ends beyond the first
// position of window tile tw (lists are sorted and disjoint).  One thread per
// entry, so the binary searches' latencies overlap instead of serialising inside
// the projection's wavefronts.
__global__ __launch_bounds__(256) void tile_index_kernel(Pileup P, const DevHom *__restrict__ homs,
														  const uint32_t *__restrict__ hom_off,
														  uint32_t *__restrict__ first)
{
	const uint32_t ntw = (P.W + PROJ_TW - 1) / PROJ_TW;
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (tid >= (uint64_t)P.N * ntw) return;
	const uint32_t g = (uint32_t)(tid / ntw), tw = (uint32_t)(tid % ntw);
	const uint32_t span0 = (P.w0 + tw * PROJ_TW) * 32u;
	uint32_t lo = hom_off[g], hi = hom_off[g + 1];
	while (lo < hi) {
		uint32_t mid = lo + ((hi - lo) >> 1);
		if (homs[mid].start + homs[mid].len <= span0) lo = mid + 1;
		else hi = mid;
	}
	first[tid] = lo;
}

static const uint32_t PROJ_HM = 8; // homology descriptors cached per genome and tile

__global__ __launch_bounds__(256) void project_kernel(Pileup P, const uint8_t *__restrict__ gbase,
													   const uint64_t *__restrict__ goff,
													   const DevHom *__restrict__ homs,
													   const uint32_t *__restrict__ hom_off,
													   const uint32_t *__restrict__ first,
													   uint32_t *__restrict__ bang_flag)
{
	__shared__ uint32_t tile[5][PROJ_TW][PROJ_TG + 1];
	__shared__ DevHom hcache[PROJ_TG][PROJ_HM];
	__shared__ uint32_t hlo[PROJ_TG], hend[PROJ_TG];
	const uint32_t ntw = (P.W + PROJ_TW - 1) / PROJ_TW;
	const uint32_t tw = blockIdx.x % ntw, tg = blockIdx.x / ntw;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	// the tile's homology descriptors: two dependent rounds for the whole block
	{
		const uint32_t gl = threadIdx.x >> 3, e = threadIdx.x & 7u; // PROJ_TG * PROJ_HM == 256
		const uint32_t g = tg * PROJ_TG + gl;
		DevHom hm = {0xffffffffu, 0, 0, 0};
		uint32_t lo = 0, h1 = 0;
		if (g < P.N) {
			lo = first[(size_t)g * ntw + tw];
			h1 = hom_off[g + 1];
			if (lo + e < h1) hm = homs[lo + e];
		}
		hcache[gl][e] = hm;
		if (e == 0) {
			hlo[gl] = lo;
			hend[gl] = h1;
		}
	}
	__syncthreads();
	const uint32_t w = tw * PROJ_TW + lane; // row of this part; reference window P.w0 + w
	const uint32_t x0 = (P.w0 + w) * 32u, x1 = x0 + 32u;
	uint32_t any_bang = 0;
	for (uint32_t gi = wave; gi < PROJ_TG; gi += 4) {
		const uint32_t g = tg * PROJ_TG + gi;
		uint32_t V = 0, N0 = 0, N1 = 0, D = 0, B = 0;
		if (g < P.N && w < P.W) {
			const uint32_t lo = hlo[gi], h1 = hend[gi];
			const uint8_t *q = gbase + goff[g];
			for (uint32_t h = lo; h < h1; h++) {
				DevHom hm; // two loads: one through a selected pointer would be a FLAT load
				if (h - lo < PROJ_HM)
					hm = hcache[gi][h - lo];
				else
					__builtin_memcpy(&hm, (const uint8_t __attribute__((address_space(1))) *)(uintptr_t)(homs + h), sizeof(DevHom));
				if (hm.start >= x1) break;
				const uint32_t he = hm.start + hm.len;
				if (he <= x0) continue;
				// covered part of this window; the 32 bytes are fetched whole (genomes are
				// padded on both sides) and the bits outside the homology masked off
				const uint32_t s = hm.start > x0 ? hm.start - x0 : 0u;
				const uint32_t e = he < x1 ? he - x0 : 32u;
				const uint32_t mask = (e >= 32u ? 0xffffffffu : ((1u << e) - 1u)) & ~((1u << s) - 1u);
				uint4 a, b;
				uint32_t n0, n1, bg;
				if (!hm.rev) {
					// position x ↔ query index iq + (x - start)
					const uint8_t *src = q + ((int64_t)hm.iq + (int64_t)x0 - (int64_t)hm.start);
					__builtin_memcpy(&a, src, 16);
					__builtin_memcpy(&b, src + 16, 16);
					n0 = gather32(a, b, 1);
					n1 = gather32(a, b, 2);
					bg = bang32(a, b);
				} else {
					// position x ↔ query index iq + (he-1-x): bytes run backwards, complemented
					const uint8_t *src = q + ((int64_t)hm.iq + (int64_t)he - (int64_t)x1);
					__builtin_memcpy(&a, src, 16);
					__builtin_memcpy(&b, src + 16, 16);
					n0 = __brev(gather32(a, b, 1));
					n1 = ~__brev(gather32(a, b, 2)); // complement: n ^ 2
					bg = __brev(bang32(a, b));
					D |= mask;
				}
				V |= mask;
				N0 |= n0 & mask;
				N1 |= n1 & mask;
				B |= bg & mask;
			}
		}
		any_bang |= B;
		tile[0][lane][gi] = V;
		tile[1][lane][gi] = N0;
		tile[2][lane][gi] = N1;
		tile[3][lane][gi] = D;
		tile[4][lane][gi] = B;
	}
	if (any_bang) atomicOr(bang_flag, 1u);
	__syncthreads();
	// rows [w][g0..g0+31] out: 128 contiguous bytes per row
	for (uint32_t e = threadIdx.x; e < 5 * PROJ_TW * PROJ_TG; e += 256) {
		const uint32_t gl = e % PROJ_TG, wl = (e / PROJ_TG) % PROJ_TW, p = e / (PROJ_TG * PROJ_TW);
		const uint32_t ww = tw * PROJ_TW + wl, g = tg * PROJ_TG + gl;
		if (ww < P.W) P.plane[p][(size_t)ww * P.Npad + g] = tile[p][wl][gl];
	}
}

// One wavefront per (tile, window chunk): lane = genome j of the tile's 64,
// the tile's 16 genomes i are wave-uniform.  Tallies stay in registers over the
// chunk and leave with one 64-bit atomic per pair.
template <bool BANG>
__global__ __launch_bounds__(64) void pairs_kernel(Pileup P, const uint32_t *__restrict__ tiles, uint32_t ntiles,
													uint32_t wchunk, uint32_t nwc, unsigned long long *__restrict__ subst,
													unsigned long long *__restrict__ homologs)
{
	// XCD-aware order: blocks are dealt round-robin over the 8 XCDs (b and b+8
	// share one), so give every XCD its own window chunks and run all the pair
	// tiles of a chunk back to back there — the chunk's plane rows (sized to fit
	// the XCD's 4 MiB L2) are then fetched from HBM once, not once per tile.
	// Placement only affects speed, never the result.
	const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
	const uint32_t tile = local % ntiles;
	const uint32_t wc = (local / ntiles) * 8u + xcd;
	if (wc >= nwc) return;
	const uint32_t ig = tiles[tile] >> 16, jt = tiles[tile] & 0xffffu;
	const uint32_t i0 = ig * PAIR_IG;
	const uint32_t j = jt * PAIR_JT + (threadIdx.x & 63u);
	const uint32_t w0 = wc * wchunk;
	const uint32_t w1 = (w0 + wchunk < P.W) ? w0 + wchunk : P.W;
	const uint32_t *__restrict__ pV = P.plane[0];
	const uint32_t *__restrict__ p0 = P.plane[1];
	const uint32_t *__restrict__ p1 = P.plane[2];
	const uint32_t *__restrict__ pD = P.plane[3];
	const uint32_t *__restrict__ pB = P.plane[4];
	uint32_t acc_h[PAIR_IG], acc_s[PAIR_IG];
#pragma unroll
	for (uint32_t t = 0; t < PAIR_IG; t++) acc_h[t] = acc_s[t] = 0;

	uint32_t w = w0;
	for (; w < w1; w++) {
		const size_t row = (size_t)w * P.Npad;
		const uint32_t vj = pV[row + j], aj = p0[row + j], bj = p1[row + j];
		uint32_t dj = 0, gj = 0;
		if (BANG) {
			dj = pD[row + j];
			gj = pB[row + j];
		}
#pragma unroll
		for (uint32_t t = 0; t < PAIR_IG; t++) {
			const size_t oi = row + i0 + t; // wave-uniform → scalar loads
			const uint32_t both = pV[oi] & vj;
			uint32_t diff = (p0[oi] ^ aj) | (p1[oi] ^ bj);
			if (BANG) diff |= ~(pD[oi] ^ dj) & (pB[oi] ^ gj);
			acc_h[t] += (uint32_t)__popc(both);
			acc_s[t] += (uint32_t)__popc(both & diff);
		}
	}
	if (j < P.N) {
#pragma unroll
		for (uint32_t t = 0; t < PAIR_IG; t++) {
			const uint32_t i = i0 + t;
			if (i < j && acc_h[t]) {
				atomicAdd(&homologs[(size_t)i * P.N + j], (unsigned long long)acc_h[t]);
				if (acc_s[t]) atomicAdd(&subst[(size_t)i * P.N + j], (unsigned long long)acc_s[t]);
			}
		}
	}
}

// tallies are accumulated for i<j only; mirror them so the matrices leave symmetric
__global__ __launch_bounds__(256) void symmetrise_kernel(uint32_t N, unsigned long long *__restrict__ a,
														  unsigned long long *__restrict__ b)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= (uint64_t)N * N) return;
	const uint32_t i = (uint32_t)(t / N), j = (uint32_t)(t % N);
	if (i > j) {
		a[t] = a[(size_t)j * N + i];
		b[t] = b[(size_t)j * N + i];
	}
}
void launch_symmetrise(uint32_t N, unsigned long long *a, unsigned long long *b, hipStream_t st)
{
	uint64_t n = (uint64_t)N * N;
	if (n) hipLaunchKernelGGL(symmetrise_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, N, a, b);
}

void launch_project(const Pileup &P, const uint8_t *gbase, const uint64_t *goff, const DevHom *homs,
					const uint32_t *hom_off, uint32_t *first, uint32_t *bang_flag, hipStream_t st)
{
	uint32_t ntw = (P.W + PROJ_TW - 1) / PROJ_TW, ntg = P.Npad / PROJ_TG;
	if (!ntw || !ntg) return;
	uint64_t entries = (uint64_t)P.N * ntw;
	hipLaunchKernelGGL(tile_index_kernel, dim3((uint32_t)((entries + 255) / 256)), dim3(256), 0, st, P, homs, hom_off, first);
	hipLaunchKernelGGL(project_kernel, dim3(ntw * ntg), dim3(256), 0, st, P, gbase, goff, homs, hom_off, first, bang_flag);
}
size_t project_index_entries(const Pileup &P) { return (size_t)P.N * ((P.W + PROJ_TW - 1) / PROJ_TW); }

void launch_pairs(const Pileup &P, bool with_bang, const uint32_t *tiles, uint32_t ntiles, uint32_t wchunk,
				  unsigned long long *subst, unsigned long long *homologs, hipStream_t st)
{
	if (!ntiles || !P.W) return;
	uint32_t nwc = (P.W + wchunk - 1) / wchunk;
	dim3 grid(((nwc + 7) / 8) * 8 * ntiles);
	if (with_bang)
		hipLaunchKernelGGL(pairs_kernel<true>, grid, dim3(64), 0, st, P, tiles, ntiles, wchunk, nwc, subst, homologs);
	else
		hipLaunchKernelGGL(pairs_kernel<false>, grid, dim3(64), 0, st, P, tiles, ntiles, wchunk, nwc, subst, homologs);
}

} // namespace phy

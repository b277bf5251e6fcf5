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

// One thread per (word w, genome g), g fastest.
__global__ __launch_bounds__(256) void project_kernel(Pileup P, const uint8_t *__restrict__ gbase,
													   const uint64_t *__restrict__ goff,
													   const DevHom *__restrict__ homs,
													   const uint32_t *__restrict__ hom_off,
													   uint32_t *__restrict__ bang_flag)
{
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t g = (uint32_t)(tid % P.Npad);
	const uint32_t w = (uint32_t)(tid / P.Npad);
	if (w >= P.W) return;
	uint32_t V = 0, N0 = 0, N1 = 0, D = 0, B = 0;
	if (g < P.N) {
		const uint32_t h0 = hom_off[g], h1 = hom_off[g + 1];
		const uint32_t x0 = w * 32u, x1 = x0 + 32u;
		// first homology whose end is beyond x0 (lists are sorted and disjoint)
		uint32_t lo = h0, hi = h1;
		while (lo < hi) {
			uint32_t mid = lo + ((hi - lo) >> 1);
			if (homs[mid].start + homs[mid].len <= x0) lo = mid + 1;
			else hi = mid;
		}
		const uint8_t *q = gbase + goff[g];
		for (uint32_t h = lo; h < h1; h++) {
			const DevHom hm = homs[h];
			if (hm.start >= x1) break;
			uint32_t s = hm.start > x0 ? hm.start : x0;
			uint32_t e = hm.start + hm.len < x1 ? hm.start + hm.len : x1;
			for (uint32_t x = s; x < e; x++) {
				// forward: query index iq + (x - start); reverse: iq + (start + len - 1 - x)
				uint32_t qi = hm.rev ? hm.iq + (hm.start + hm.len - 1u - x) : hm.iq + (x - hm.start);
				uint32_t c = q[qi];
				uint32_t n = (c >> 1) & 3u;
				if (hm.rev) n ^= 2u;
				uint32_t bit = 1u << (x - x0);
				V |= bit;
				if (n & 1u) N0 |= bit;
				if (n & 2u) N1 |= bit;
				if (hm.rev) D |= bit;
				if (c == '!') B |= bit;
			}
		}
		if (B) atomicOr(bang_flag, 1u);
	}
	const size_t o = (size_t)w * P.Npad + g;
	P.plane[0][o] = V;
	P.plane[1][o] = N0;
	P.plane[2][o] = N1;
	P.plane[3][o] = D;
	P.plane[4][o] = B;
}

// One wavefront per (tile, window chunk): lane = genome j of the tile's 64,
// the tile's 16 genomes i are wave-uniform.  Tallies stay in registers over the
// chunk and leave with one 64-bit atomic per pair.
template <bool BANG>
__global__ __launch_bounds__(64) void pairs_kernel(Pileup P, const uint32_t *__restrict__ tiles, uint32_t ntiles,
													uint32_t wchunk, unsigned long long *__restrict__ subst,
													unsigned long long *__restrict__ homologs)
{
	const uint32_t tile = blockIdx.x % ntiles;
	const uint32_t wc = blockIdx.x / ntiles;
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

	for (uint32_t w = w0; w < w1; w++) {
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

void launch_project(const Pileup &P, const uint8_t *gbase, const uint64_t *goff, const DevHom *homs,
					const uint32_t *hom_off, uint32_t *bang_flag, hipStream_t st)
{
	uint64_t threads = (uint64_t)P.W * P.Npad;
	uint32_t blocks = (uint32_t)((threads + 255) / 256);
	if (!blocks) return;
	hipLaunchKernelGGL(project_kernel, dim3(blocks), dim3(256), 0, st, P, gbase, goff, homs, hom_off, bang_flag);
}

void launch_pairs(const Pileup &P, bool with_bang, const uint32_t *tiles, uint32_t ntiles, uint32_t wchunk,
				  unsigned long long *subst, unsigned long long *homologs, hipStream_t st)
{
	if (!ntiles || !P.W) return;
	uint32_t nwc = (P.W + wchunk - 1) / wchunk;
	dim3 grid(ntiles * nwc);
	if (with_bang)
		hipLaunchKernelGGL(pairs_kernel<true>, grid, dim3(64), 0, st, P, tiles, ntiles, wchunk, subst, homologs);
	else
		hipLaunchKernelGGL(pairs_kernel<false>, grid, dim3(64), 0, st, P, tiles, ntiles, wchunk, subst, homologs);
}

} // namespace phy

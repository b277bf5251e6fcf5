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
//     V  covered            N0,N1  the 2-bit code of the base on the forward strand (A 00, C 01,
//     D  reverse homology          G 10, T 11 — the genomes' packed form, lean_core.h; reverse hits
//     B  raw byte is '!'           store the complement, both bits flipped); '!' has code 00
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
// The projection itself reads the genomes as 2-bit codes: 12 bytes per window and genome.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace phy {

// Bit order inside a plane word.  A window is 32 reference positions; the query side arrives as
// 2-bit codes, 16 bases per dword with the first base in bits 31..30 (lean_core.h: Q2), so the two
// dwords A (positions 0..15) and B (16..31) of a window interleave into one word with two masks and
// a shift: position t < 16 sits at bit 31 - 2t, position 16 + t at bit 30 - 2t.  Every plane of every
// genome uses the same order and the pair kernel only ANDs, XORs and counts bits, so the order is free
// to choose.  The code's high bit is plane N1, its low bit N0 (A 00, C 01, G 10, T 11: the complement
// flips both).
static __device__ __forceinline__ uint32_t plane_bit(uint32_t p) { return p < 16u ? 31u - 2u * p : 30u - 2u * (p - 16u); }

// the high (HI) or low bits of the 32 codes in A, B → one word in plane order
template <bool HI> static __device__ __forceinline__ uint32_t code_plane(uint32_t a, uint32_t b)
{
	return HI ? (a & 0xaaaaaaaau) | ((b & 0xaaaaaaaau) >> 1) : ((a & 0x55555555u) << 1) | (b & 0x55555555u);
}

// an explicitly global load (a pointer selected between LDS and global memory would be FLAT)
static __device__ __forceinline__ DevHom gload_hom(const DevHom *p)
{
	DevHom hm;
	__builtin_memcpy(&hm, (const uint8_t __attribute__((address_space(1))) *)(uintptr_t)p, sizeof(DevHom));
	return hm;
}

static const uint32_t PROJ_TW = 64; // words per tile
// Branches the projection is better off without (same-box A/B, profiles/r05_ab_projection_micro*.txt; 0 restores them):
// (measured and left: the second piece without its test +1...5 %; v_alignbit instead of the 64-bit shifts 0...+2 %; the '!'
// look-up's test as a scalar branch 0 %, or hoisted out of the pieces into one test per genome 0...+1 %)
static const uint32_t PROJ_TG = 32; // genomes per tile (LDS with three planes: 3*64*(TG+1)*4 = 25 KB at 32 → 5 blocks per CU)
static const uint32_t PROJ_GPW = PROJ_TG / 4; // genomes per wavefront

// Projection: one block per tile of 64 reference windows × 32 genomes.
// Reading side: a wavefront takes one genome and its 64 lanes take the 64
// consecutive windows, so the query bytes of a homology are read as contiguous
// 32-byte pieces (2 KiB per wave, coalesced).  Writing side: the tile is
// transposed through LDS so that every plane row [w][g0..g0+31] leaves as one
// full 128-byte line.
// first[g*ntw + tw] = first homology of genome g that ends beyond the first
// position of window tile tw (lists are sorted and disjoint).  One thread per
// entry, so the binary searches' latencies overlap instead of serialising inside
// the projection's wavefronts.  Bit 31 (TILE_BAD) says that a homology of g carries a
// non-ACGT query position ('!') into this tile's 2048 reference positions: only there does the
// projection look positions up in the genome's list (QBAD) to form plane B.
static const uint32_t TILE_BAD = 0x80000000u;
static __device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *__restrict__ a, uint32_t n, uint32_t key)
{
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		const uint32_t mid = lo + ((hi - lo) >> 1);
		if (a[mid] < key) lo = mid + 1;
		else hi = mid;
	}
	return lo;
}
__global__ __launch_bounds__(256) void tile_index_kernel(Pileup P, QuerySrc Q, const DevHom *__restrict__ homs,
														  const uint32_t *__restrict__ hom_rng,
														  uint32_t *__restrict__ first, uint32_t g0, uint32_t g1, uint32_t *__restrict__ zero_flags)
{
	const uint32_t ntw = (P.W + PROJ_TW - 1) / PROJ_TW;
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (zero_flags && t == 0) zero_flags[0] = zero_flags[3] = 0; // the projection's '!' flag and its count of listed '!' (words 1, 2 are the attach's)
	if (t >= (uint64_t)(g1 - g0) * ntw) return;
	const uint32_t g = g0 + (uint32_t)(t / ntw), tw = (uint32_t)(t % ntw);
	const uint64_t tid = (uint64_t)g * ntw + tw;
	const uint32_t span0 = (P.w0 + tw * PROJ_TW) * 32u;
	const uint32_t h1 = hom_rng[2 * g + 1];
	uint32_t lo = hom_rng[2 * g], hi = h1; // genome g's list is homs[lo, hi)
	while (lo < hi) {
		uint32_t mid = lo + ((hi - lo) >> 1);
		if (homs[mid].start + homs[mid].len <= span0) lo = mid + 1;
		else hi = mid;
	}
	uint32_t flag = 0;
	const uint32_t b0 = Q.qbad_off[g], nb = Q.qbad_off[g + 1] - b0;
	if (nb) {
		const uint64_t span1 = (uint64_t)span0 + PROJ_TW * 32u;
		for (uint32_t h = lo; h < h1 && !flag; h++) {
			const DevHom hm = homs[h];
			if (hm.start >= span1) break;
			const uint64_t he = (uint64_t)hm.start + hm.len;
			const uint64_t xs = hm.start > span0 ? hm.start : span0, xe = he < span1 ? he : span1;
			// the query positions behind [xs, xe)
			const uint64_t qa = hm.rev ? hm.iq + (he - xe) : hm.iq + (xs - hm.start), qb = qa + (xe - xs);
			const uint32_t k = lower_bound_u32(Q.qbad + b0, nb, (uint32_t)qa);
			if (k < nb && Q.qbad[b0 + k] < qb) flag = TILE_BAD;
		}
	}
	first[tid] = lo | flag;
}

static const uint32_t PROJ_HM = 256 / PROJ_TG; // homology descriptors cached per genome and tile (one per thread)

// FIVE = false writes V, N0, N1 only (and still raises bang_flag when a projected
// position holds '!'): the host then repeats the projection with all five planes.
// bang_list (FIVE = false only, may be null): every projected '!' as {genome | reverse << 31, reference position},
// appended through the counter bang_flag[3]; more than bang_cap of them raise bit 1 of bang_flag[0].
template <bool FIVE>
__global__ __launch_bounds__(256) void project_kernel(Pileup P, QuerySrc Q,
													   const DevHom *__restrict__ homs,
													   const uint32_t *__restrict__ hom_rng,
													   const uint32_t *__restrict__ first,
													   uint32_t *__restrict__ bang_flag, uint32_t tg0, uint32_t ntiles,
													   uint32_t *__restrict__ bang_list, uint32_t bang_cap)
{
	constexpr uint32_t NP = FIVE ? 5u : 3u;
	__shared__ uint32_t tile[NP][PROJ_TW][PROJ_TG + 1];
	__shared__ DevHom hcache[PROJ_TG][PROJ_HM];
	__shared__ uint32_t hends[PROJ_TG][PROJ_HM]; // where the cached homologies end on the reference (0xffffffff: none)
	__shared__ uint32_t hlo[PROJ_TG], hend[PROJ_TG], tbad[PROJ_TG];
	__shared__ uint32_t below[33]; // below[t] = plane-order mask of the positions < t
	const uint32_t ntw = (P.W + PROJ_TW - 1) / PROJ_TW;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (threadIdx.x < 33) {
		uint32_t m = 0;
		for (uint32_t pp = 0; pp < threadIdx.x; pp++) m |= 1u << plane_bit(pp);
		below[threadIdx.x] = m;
	}
	// Round 5: a block takes tile after tile (tile t, t + gridDim.x, ...: the launch is as many blocks as the chip holds at once)
	// and keeps the NEXT tile's homology descriptors — and the tile index entry of the one after that — on their way while
	// it works on this one: what used to be three dependent fetches per tile (tile index, descriptors, codes) is one.
	// A thread's part in that: descriptor pe of genome pgl of the tile.
	const uint32_t pgl = threadIdx.x / PROJ_HM, pe = threadIdx.x % PROJ_HM; // PROJ_TG * PROJ_HM == 256
	struct Ahead {
		uint32_t f, h1; // the tile index entry (first homology | TILE_BAD) and the end of the genome's list
	};
	auto index_of = [&](uint32_t t) {
		Ahead a = {0u, 0u};
		if (t < ntiles) {
			const uint32_t g = (tg0 + t / ntw) * PROJ_TG + pgl;
			if (g < P.N) {
				a.f = first[(size_t)g * ntw + t % ntw];
				a.h1 = hom_rng[2 * g + 1];
			}
		}
		return a;
	};
	auto descriptor_of = [&](uint32_t t, const Ahead &a) {
		DevHom hm = {0xffffffffu, 0, 0, 0};
		const uint32_t lo = a.f & ~TILE_BAD;
		if (t < ntiles && lo + pe < a.h1) hm = homs[lo + pe];
		return hm;
	};
	uint32_t any_bang = 0;
	const uint32_t stride = gridDim.x;
	Ahead a_cur = index_of(blockIdx.x), a_next = index_of(blockIdx.x + stride);
	DevHom hm_cur = descriptor_of(blockIdx.x, a_cur);
	for (uint32_t t = blockIdx.x; t < ntiles; t += stride) {
	const uint32_t tw = t % ntw, tg = tg0 + t / ntw;
	if (t != blockIdx.x) __syncthreads(); // the tile before has left LDS
	{
		hcache[pgl][pe] = hm_cur;
		hends[pgl][pe] = hm_cur.start == 0xffffffffu ? 0xffffffffu : hm_cur.start + hm_cur.len;
		if (pe == 0) {
			hlo[pgl] = a_cur.f & ~TILE_BAD;
			hend[pgl] = a_cur.h1;
			tbad[pgl] = a_cur.f & TILE_BAD;
		}
	}
	// on their way while this tile is worked on: the next tile's descriptors (its index entry came in a tile ago) and the
	// index entry of the tile after it
	const DevHom hm_next = descriptor_of(t + stride, a_next);
	const Ahead a_next2 = index_of(t + 2 * stride);
	__syncthreads();
	const uint32_t w = tw * PROJ_TW + lane; // row of this part; reference window P.w0 + w
	const uint32_t x0 = (P.w0 + w) * 32u, x1 = x0 + 32u;
	// A wavefront owns PROJ_GPW genomes of the tile.  The covering homology of each is
	// looked up first (LDS only), then the PROJ_GPW 12-byte reads (32 codes at any 2-bit
	// offset) are issued together — one read per genome in flight at a time left the kernel
	// waiting on HBM latency three quarters of its cycles — then the planes are formed.  A
	// window that a second homology also touches (list boundaries) is finished by the loop below.
	struct Piece {
		int64_t pos; // genome-buffer position (in bases, = byte offset of the byte form) of the lowest query index read
		uint32_t mask, rev, next;
	};
	auto load_hom = [&](uint32_t gi, uint32_t h, uint32_t lo) {
		DevHom hm; // two loads: one through a selected pointer would be a FLAT load
		if (h - lo < PROJ_HM)
			hm = hcache[gi][h - lo];
		else
			hm = gload_hom(homs + h);
		return hm;
	};
	// first homology of genome gi at or after h that overlaps the window; next = h1 if none
	auto find_piece = [&](uint32_t gi, int64_t q, uint32_t h, uint32_t lo, uint32_t h1) {
		Piece pc = {q, 0u, 0u, h1};
		for (; h < h1; h++) {
			const DevHom hm = load_hom(gi, h, lo);
			if (hm.start >= x1) break;
			const uint32_t he = hm.start + hm.len;
			if (he <= x0) continue;
			// covered part of this window; the 32 codes are fetched whole (genomes are
			// padded on both sides) and the bits outside the homology masked off
			pc.mask = 0xffffffffu;
			if (hm.start > x0 || he < x1) {
				const uint32_t s = hm.start > x0 ? hm.start - x0 : 0u;
				const uint32_t e = he < x1 ? he - x0 : 32u;
				pc.mask = below[e] & ~below[s];
			}
			// forward: position x ↔ query index iq + (x - start)
			// reverse: position x ↔ query index iq + (he-1-x): codes run backwards, complemented
			pc.pos = hm.rev ? q + ((int64_t)hm.iq + (int64_t)he - (int64_t)x1)
							: q + ((int64_t)hm.iq + (int64_t)x0 - (int64_t)hm.start);
			pc.rev = hm.rev;
			pc.next = he < x1 ? h + 1 : h1; // lists are sorted and disjoint: nothing else if this one reaches the end
			break;
		}
		return pc;
	};
	// the 32 codes at pc.pos: three dwords of Q2, shifted into two
	auto fetch = [&](const Piece &pc, uint32_t &w0, uint32_t &w1, uint32_t &w2) {
		const uint32_t *src = Q.q2 + (pc.pos >> 4);
		struct { uint32_t a, b, c; } v;
		__builtin_memcpy(&v, (const uint8_t __attribute__((address_space(1))) *)(uintptr_t)src, 12);
		w0 = v.a;
		w1 = v.b;
		w2 = v.c;
	};
	auto add_piece = [&](const Piece &pc, uint32_t gi, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t &V, uint32_t &N0,
						 uint32_t &N1, uint32_t &D, uint32_t &B) {
		const uint32_t sh = 2u * ((uint32_t)pc.pos & 15u);
		uint32_t a = (uint32_t)(((((uint64_t)w0 << 32) | w1) << sh) >> 32);
		uint32_t b = (uint32_t)(((((uint64_t)w1 << 32) | w2) << sh) >> 32);
		uint32_t hi, lo;
		{
			// a reverse piece: all 64 bits reversed — the codes run backwards and every code's two bits have changed places —
			// and complemented.  Without a branch (nearly every wavefront holds a reverse piece somewhere and then runs both
			// sides of one): the operands selected, both planes formed, swapped and flipped by the flag.
			const bool rv = pc.rev != 0;
			const uint32_t x = rv ? __builtin_bitreverse32(b) : a, y = rv ? __builtin_bitreverse32(a) : b;
			const uint32_t pt = code_plane<true>(x, y), pf = code_plane<false>(x, y), flip = rv ? 0xffffffffu : 0u;
			hi = (rv ? pf : pt) ^ flip;
			lo = (rv ? pt : pf) ^ flip;
			if (FIVE) D |= rv ? pc.mask : 0u;
		}
		V |= pc.mask;
		N0 |= lo & pc.mask;
		N1 |= hi & pc.mask;
		if (tbad[gi]) {
			// '!' among the 32 query positions [pc.pos, pc.pos + 32)?  (the tile index found one in this tile)
			const uint32_t g = tg * PROJ_TG + gi;
			const uint32_t b0 = Q.qbad_off[g], nb = Q.qbad_off[g + 1] - b0;
			const int64_t rel = pc.pos - (int64_t)Q.goff[g]; // query index of the lowest position read (may be < 0)
			const uint32_t from = rel < 0 ? 0u : (uint32_t)rel;
			for (uint32_t k = lower_bound_u32(Q.qbad + b0, nb, from); k < nb; k++) {
				const int64_t u = (int64_t)Q.qbad[b0 + k] - rel;
				if (u >= 32) break;
				const uint32_t wp = pc.rev ? 31u - (uint32_t)u : (uint32_t)u; // position inside the window
				const uint32_t bit = (1u << plane_bit(wp)) & pc.mask;
				B |= bit;
				if (!FIVE && bit && bang_list) {
					const uint32_t idx = atomicAdd(bang_flag + 3, 1u);
					if (idx < bang_cap) {
						bang_list[2 * idx] = g | (pc.rev ? 0x80000000u : 0u);
						bang_list[2 * idx + 1] = x0 + wp;
					} else {
						atomicOr(bang_flag, 2u);
					}
				}
			}
		}
	};
	// Round 5.  The covering homology of a window without a loop: among the tile's cached descriptors (sorted, disjoint) it
	// is the one whose index is the number of cached homologies that end at or before the window's first position — eight
	// compares against wave-uniform LDS words.  A window takes that homology AND the one behind it (a window with a list
	// boundary inside — one in fifty, but nearly every wavefront holds such a window — needs both; the second one's
	// mask is empty otherwise and its load repeats the first's address), both loads of all the wavefront's genomes are
	// issued together, and only what neither covers — a third piece in 32 positions, homologies beyond the cache —
	// goes through the searching loop of rounds 1-4 below.
	struct Fast {
		uint32_t mask, rev, more; // more: first homology the searching loop still has to look at (h1: none)
		int32_t rel;              // query index of the lowest position read (may be < 0)
	};
	auto fast_piece = [&](const DevHom &hm, bool usable, Fast &f) { // -> does this homology end inside the window?
		const uint32_t he = hm.start + hm.len;
		const bool ov = usable && hm.start < x1 && he > x0;
		const uint32_t s = ov && hm.start > x0 ? hm.start - x0 : 0u, e = !ov ? 0u : he < x1 ? he - x0 : 32u;
		f.mask = below[e] & ~below[s];
		f.rev = hm.rev;
		f.rel = (int32_t)(hm.rev ? hm.iq + he - x1 : hm.iq + x0 - hm.start);
		return ov && he < x1;
	};
	// (the wavefront's genomes in two rounds of PROJ_GPW / 2: eight loads in flight per lane, and registers for five blocks per CU)
	constexpr uint32_t HG = PROJ_GPW / 2;
#pragma unroll 1
	for (uint32_t half = 0; half < 2; half++) {
	Fast fa[HG], fb[HG];
	uint32_t a0[HG], a1[HG], a2[HG], b0[HG], b1[HG], b2[HG];
#pragma unroll
	for (uint32_t u = 0; u < HG; u++) {
		const uint32_t gi = wave + 4 * (u + half * HG), g = tg * PROJ_TG + gi;
		const bool valid = g < P.N && w < P.W;
		const uint32_t lo = hlo[gi], h1 = hend[gi];
		uint32_t idx = 0;
#pragma unroll
		for (uint32_t e = 0; e < PROJ_HM; e++) idx += hends[gi][e] <= x0 ? 1u : 0u;
		const DevHom ha = hcache[gi][idx < PROJ_HM ? idx : PROJ_HM - 1], hb = hcache[gi][idx + 1 < PROJ_HM ? idx + 1 : PROJ_HM - 1];
		const bool a_ends = fast_piece(ha, valid && idx < PROJ_HM, fa[u]);
		const bool b_ends = fast_piece(hb, a_ends && idx + 1 < PROJ_HM, fb[u]);
		// what is left for the searching loop: everything when the cache holds nothing that reaches the window; the homologies
		// behind the second piece when that one ends inside the window too; the ones behind the cache
		uint32_t more = h1;
		if (valid) {
			if (idx >= PROJ_HM) more = lo + PROJ_HM;
			else if (a_ends && idx + 1 >= PROJ_HM) more = lo + PROJ_HM;
			else if (b_ends) more = lo + idx + 2;
		}
		fa[u].more = more < h1 ? more : h1;
		if (!fa[u].mask) fa[u].rel = 0;
		if (!fb[u].mask) fb[u].rel = fa[u].rel;
		const uint32_t *base = Q.q2 + (Q.goff[g < P.N ? g : 0] >> 4);
		struct { uint32_t a, b, c; } v;
		__builtin_memcpy(&v, (const uint8_t __attribute__((address_space(1))) *)(uintptr_t)(base + (fa[u].rel >> 4)), 12);
		a0[u] = v.a, a1[u] = v.b, a2[u] = v.c;
		__builtin_memcpy(&v, (const uint8_t __attribute__((address_space(1))) *)(uintptr_t)(base + (fb[u].rel >> 4)), 12);
		b0[u] = v.a, b1[u] = v.b, b2[u] = v.c;
	}
#pragma unroll
	for (uint32_t u = 0; u < HG; u++) {
		const uint32_t gi = wave + 4 * (u + half * HG), g = tg * PROJ_TG + gi;
		uint32_t V = 0, N0 = 0, N1 = 0, D = 0, B = 0;
		const int64_t q = (int64_t)Q.goff[g < P.N ? g : 0];
		add_piece(Piece{q + fa[u].rel, fa[u].mask, fa[u].rev, 0u}, gi, a0[u], a1[u], a2[u], V, N0, N1, D, B); // (an empty mask adds nothing)
		if (fb[u].mask) add_piece(Piece{q + fb[u].rel, fb[u].mask, fb[u].rev, 0u}, gi, b0[u], b1[u], b2[u], V, N0, N1, D, B);
		{
			const uint32_t lo = hlo[gi], h1 = hend[gi];
			uint32_t h = fa[u].more;
			while (h < h1) {
				const Piece more = find_piece(gi, q, h, lo, h1);
				if (!more.mask) break;
				uint32_t e0, e1, e2;
				fetch(more, e0, e1, e2);
				add_piece(more, gi, e0, e1, e2, V, N0, N1, D, B);
				h = more.next;
			}
		}
		any_bang |= B;
		tile[0][lane][gi] = V;
		tile[1][lane][gi] = N0;
		tile[2][lane][gi] = N1;
		if (FIVE) {
			tile[NP - 2][lane][gi] = D;
			tile[NP - 1][lane][gi] = B;
		}
	}
	} // half
	__syncthreads();
	// rows [w][g0..g0+31] out: 128 contiguous bytes per row; a thread keeps its genome
	// column and walks down the rows
	{
		constexpr uint32_t RPP = 256 / PROJ_TG; // rows per pass of the block
		const uint32_t gl = threadIdx.x % PROJ_TG, wl0 = threadIdx.x / PROJ_TG;
		const uint32_t g = tg * PROJ_TG + gl;
#pragma unroll
		for (uint32_t p = 0; p < NP; p++) {
			uint32_t *dst = P.plane[p] + (size_t)(tw * PROJ_TW + wl0) * P.Npad + g;
#pragma unroll
			for (uint32_t k = 0; k < PROJ_TW / RPP; k++)
				if (tw * PROJ_TW + wl0 + RPP * k < P.W) dst[(size_t)(RPP * k) * P.Npad] = tile[p][wl0 + RPP * k][gl];
		}
	}
	hm_cur = hm_next;
	a_cur = a_next;
	a_next = a_next2;
	} // tiles
	if (any_bang) atomicOr(bang_flag, 1u);
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

	// the lane's own words of the next window are requested before this window's pairs are
	// counted (the loop body is ~110 VALU instructions: plenty to cover the load)
	uint32_t vj_n = 0, aj_n = 0, bj_n = 0, dj_n = 0, gj_n = 0;
	if (w0 < w1) {
		const size_t row = (size_t)w0 * P.Npad;
		vj_n = pV[row + j];
		aj_n = p0[row + j];
		bj_n = p1[row + j];
		if (BANG) {
			dj_n = pD[row + j];
			gj_n = pB[row + j];
		}
	}
	uint32_t w = w0;
	for (; w < w1; w++) {
		const size_t row = (size_t)w * P.Npad;
		const uint32_t vj = vj_n, aj = aj_n, bj = bj_n, dj = dj_n, gj = gj_n;
		{
			const size_t rn = (size_t)(w + 1 < w1 ? w + 1 : w) * P.Npad;
			vj_n = pV[rn + j];
			aj_n = p0[rn + j];
			bj_n = p1[rn + j];
			if (BANG) {
				dj_n = pD[rn + j];
				gj_n = pB[rn + j];
			}
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

// ───────────────── the pair tallies as a matrix-core contraction (the default without '!') ─────────────────
//
// Over the projected planes the two tallies are contractions over reference positions:
//     homologs(i,j) = sum_p V_i V_j
//     matches(i,j)  = sum_p V_i V_j (1 + a_i a_j)(1 + b_i b_j) / 4,     a = (-1)^N0, b = (-1)^N1
//                   = (sum_p [V V' + Va Va' + Vb Vb' + Vab Vab']) / 4,   substitutions = homologs - matches
// — four channels of values in {-1, 0, +1}, which FP4 (E2M1: +1.0 = 0b0010, sign = bit 3) holds exactly, so the
// densest matrix instruction of gfx950 applies: v_mfma_f32_32x32x64_f8f6f4, 65536 multiply-adds per wavefront in
// the cycles of the bf16 32x32x16 form.  The f32 accumulators hold integers exactly below 2^24 (a wavefront's
// window chunk contributes at most 3 x 32 x wchunk to one of them).  (seqcmp counts byte mismatches,
// libs/seqcmp.c:13-28; on the planes that is this sum.  The north star kept the matrix cores out of the merge-join
// formulation; with the pileup the work IS a contraction — measured: tools/microbench/mfma_pairs.hip)
//
// Operands are made in registers from the plane words.  A plane word holds 32 positions; its position class d
// (positions = d mod 4) becomes one dword of 8 nibbles by one shift and one mask, V landing on nibble bit 1 and the
// channel's sign plane on nibble bit 3 (a set sign bit over V = 0 is -0: harmless).  The order of K inside an
// instruction is free as long as both operands use the same one — they do, the expansion is the same code for the
// i side and the j side.  Lane l holds genome l & 31 of a group of 32 genomes; lanes 0..31 take window w, lanes
// 32..63 window w + 1: the instruction's two K blocks of 32.  A wavefront owns a tile of 64 x 64 genomes (2 x 2
// instructions per channel and step) and a chunk of windows; two wavefronts per SIMD.
typedef int pm_v8i __attribute__((ext_vector_type(8)));
typedef float pm_v16f __attribute__((ext_vector_type(16)));
static const int PM_G = 2;  // groups of 32 genomes per tile side
static const int PM_NB = 3; // register sets of plane words in flight (a set is refilled right after its step has expanded it)

static __device__ __forceinline__ void pm_expand_v(uint32_t V, uint32_t o[4])
{
	o[0] = (V << 1) & 0x22222222u;
	o[1] = V & 0x22222222u;
	o[2] = (V >> 1) & 0x22222222u;
	o[3] = (V >> 2) & 0x22222222u;
}
static __device__ __forceinline__ void pm_expand_s(uint32_t S, const uint32_t v[4], uint32_t o[4])
{
	o[0] = v[0] | ((S << 3) & 0x88888888u);
	o[1] = v[1] | ((S << 2) & 0x88888888u);
	o[2] = v[2] | ((S << 1) & 0x88888888u);
	o[3] = v[3] | (S & 0x88888888u);
}
static __device__ __forceinline__ pm_v16f pm_mfma(const uint32_t a[4], const uint32_t b[4], pm_v16f c)
{
	const pm_v8i va = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0};
	const pm_v8i vb = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], 0, 0, 0, 0};
	return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0, 0, 0); // both FP4, unscaled
}
struct PmWords {
	uint32_t v, a, b;
};
// DIAG: a tile on the diagonal of the pair grid — its sub-tile below the diagonal is left out
// The wavefront takes `cpw` window chunks of its XCD one after the other (chunk wc0, wc0 + 8, ...: the chunks are dealt
// round-robin over the XCDs) and keeps the tallies in its accumulators across them: one flush of 64-bit atomics at the end
// instead of one per chunk — with 592 chunks of 264 windows (C4, the length the L2 holds the rows of) the flushes were a
// fifth of the kernel (4.7 ms; 6.7 with chunks half as long, 4.4 with twice — where the rows no longer fit).
template <bool DIAG>
static __device__ __forceinline__ void pairs_mfma_body(const Pileup &P, uint32_t ti, uint32_t tj, uint32_t wc0, uint32_t cpw, uint32_t wchunk,
														uint32_t nwc, unsigned long long *__restrict__ subst, unsigned long long *__restrict__ homologs)
{
	constexpr int G = PM_G, NG = 2 * PM_G;
	const uint32_t lane = threadIdx.x & 63u, gl = lane & 31u, half = lane >> 5;
	const uint32_t so_i = ti * G * 128u, so_j = tj * G * 128u; // byte offset of the tile's first i / j genome in a row
	const uint32_t step = 2u * P.Npad * 4u;
	pm_v16f acc_h[G][G], acc_t[G][G];
#pragma unroll
	for (int a = 0; a < G; a++)
#pragma unroll
		for (int b = 0; b < G; b++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc_h[a][b][r] = acc_t[a][b][r] = 0.f;
#define PM_NEED(a, b) (!DIAG || (a) <= (b))
	// the current chunk's rows as buffers: an offset beyond the chunk reads 0, so the last step's odd window and the
	// loads issued ahead need no guard (and offsets stay 32-bit whatever the planes' size)
	__amdgpu_buffer_rsrc_t rv, ra, rb;
	uint32_t off = 0; // this lane's word of its window's row
	auto load = [&](PmWords (&x)[NG]) {
#pragma unroll
		for (int g = 0; g < NG; g++) {
			const uint32_t so = g < G ? so_i : so_j, im = (uint32_t)(g < G ? g : g - G) * 128u;
			x[g].v = __builtin_amdgcn_raw_buffer_load_b32(rv, off + im, so, 0);
			x[g].a = __builtin_amdgcn_raw_buffer_load_b32(ra, off + im, so, 0);
			x[g].b = __builtin_amdgcn_raw_buffer_load_b32(rb, off + im, so, 0);
		}
		off += step;
	};
	constexpr int NMF = DIAG ? (G * (G + 1)) / 2 : G * G; // matrix instructions per channel and step
	auto compute = [&](PmWords (&x)[NG]) {
		uint32_t vd[NG][4], op[NG][4];
#pragma unroll
		for (int g = 0; g < NG; g++) pm_expand_v(x[g].v, vd[g]);
#pragma unroll
		for (int a = 0; a < G; a++)
#pragma unroll
			for (int b = 0; b < G; b++)
				if (PM_NEED(a, b)) acc_h[a][b] = pm_mfma(vd[a], vd[G + b], acc_h[a][b]);
#pragma unroll
		for (int c = 0; c < 3; c++) {
#pragma unroll
			for (int g = 0; g < NG; g++) pm_expand_s(c == 0 ? x[g].a : c == 1 ? x[g].b : (x[g].a ^ x[g].b), vd[g], op[g]);
#pragma unroll
			for (int a = 0; a < G; a++)
#pragma unroll
				for (int b = 0; b < G; b++)
					if (PM_NEED(a, b)) acc_t[a][b] = pm_mfma(op[a], op[G + b], acc_t[a][b]);
		}
		load(x); // the set is free again: its words for the step PM_NB from now
		// ask the scheduler to deal the expansion's vector instructions out between the matrix instructions
		constexpr int PER = (29 * NG + 4 * NMF - 1) / (4 * NMF);
#pragma unroll
		for (int i = 0; i < 4 * NMF; i++) {
			__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
			__builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
			if (i % NMF == 0) __builtin_amdgcn_sched_group_barrier(0x020, 3 * NG / 4, 0);
		}
	};
	PmWords x[PM_NB][NG];
	for (uint32_t u = 0; u < cpw; u++) {
		const uint32_t wc = wc0 + 8u * u;
		if (wc >= nwc) break;
		const uint32_t w0 = wc * wchunk, w1 = (w0 + wchunk < P.W) ? w0 + wchunk : P.W;
		const uint32_t chunk_bytes = (w1 - w0) * P.Npad * 4u;
		const size_t row0 = (size_t)w0 * P.Npad;
		rv = __builtin_amdgcn_make_buffer_rsrc((void *)(P.plane[0] + row0), 0, chunk_bytes, 0x00020000);
		ra = __builtin_amdgcn_make_buffer_rsrc((void *)(P.plane[1] + row0), 0, chunk_bytes, 0x00020000);
		rb = __builtin_amdgcn_make_buffer_rsrc((void *)(P.plane[2] + row0), 0, chunk_bytes, 0x00020000);
		off = (half * P.Npad + gl) * 4u;
#pragma unroll
		for (int k = 0; k < PM_NB; k++) {
			load(x[k]);
			// The sets are asked for one after the other, as the loop refills them.  Left to itself the scheduler sorts these 36
			// loads by plane, the set the loop starts with has words among the last — and where the loop's first trip has to wait
			// for every load in flight (s_waitcnt vmcnt(0)), the compiler makes every trip wait so: the two sets just asked for
			// with it, a load's whole latency every three steps.  In order the waits count: vmcnt(35) ... (24), (12).
			__builtin_amdgcn_sched_barrier(0);
		}
		for (uint32_t w = w0; w < w1; w += 2 * PM_NB) {
#pragma unroll
			for (int k = 0; k < PM_NB; k++) compute(x[k]);
		}
	}
	// C/D layout of the 32x32 forms: column = lane & 31 (the B operand's row: genome j), row = (r & 3) + 8 (r >> 2)
	// + 4 (lane >> 5) (the A operand's row: genome i)
#pragma unroll
	for (int a = 0; a < G; a++)
#pragma unroll
		for (int b = 0; b < G; b++) {
			if (!PM_NEED(a, b)) continue;
			const uint32_t j = (tj * G + b) * 32u + gl;
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const uint32_t i = (ti * G + a) * 32u + (r & 3) + 8 * (r >> 2) + 4 * half;
				const int h = (int)acc_h[a][b][r], t = (int)acc_t[a][b][r];
				if (i < j && j < P.N && h) {
					atomicAdd(&homologs[(size_t)i * P.N + j], (unsigned long long)h);
					const int sb = (3 * h - t) >> 2; // matches = (h + t) / 4
					if (sb) atomicAdd(&subst[(size_t)i * P.N + j], (unsigned long long)sb);
				}
			}
		}
#undef PM_NEED
}
// One wavefront per (tile of 64 x 64 genomes, window chunk); the same XCD-aware order as pairs_kernel.
__global__ __launch_bounds__(64, 2) void pairs_mfma_kernel(Pileup P, const uint32_t *__restrict__ tiles, uint32_t ntiles,
															uint32_t wchunk, uint32_t nwc, uint32_t cpw, unsigned long long *__restrict__ subst,
															unsigned long long *__restrict__ homologs, unsigned long long *__restrict__ clk)
{
	const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
	const uint32_t tile = local % ntiles;
	const uint32_t wc0 = (local / ntiles) * cpw * 8u + xcd; // this wavefront's chunks: wc0, wc0 + 8, ... (cpw of them)
	if (wc0 >= nwc) return;
	// clk (profiling only): the wavefronts' lifetimes in shader cycles and in ticks of the constant 100 MHz counter —
	// their ratio is the clock the chip held under this kernel's mix of matrix and vector instructions
	unsigned long long c0 = 0, r0 = 0;
	if ((blockIdx.x & 127u) != 0) clk = nullptr; // (a sample of the wavefronts: two atomics each on the same two words add up otherwise)
	if (clk) {
		c0 = __builtin_amdgcn_s_memtime();
		r0 = __builtin_amdgcn_s_memrealtime();
	}
	const uint32_t ti = tiles[tile] >> 16, tj = tiles[tile] & 0xffffu;
	if (ti == tj) pairs_mfma_body<true>(P, ti, tj, wc0, cpw, wchunk, nwc, subst, homologs);
	else pairs_mfma_body<false>(P, ti, tj, wc0, cpw, wchunk, nwc, subst, homologs);
	if (clk && threadIdx.x == 0) {
		atomicAdd(&clk[0], __builtin_amdgcn_s_memtime() - c0);
		atomicAdd(&clk[1], __builtin_amdgcn_s_memrealtime() - r0);
	}
}
uint32_t pairs_mfma_tile() { return PM_G * 32u; }
uint32_t pairs_mfma_max_wchunk() { return (1u << 24) / (3u * 32u) - 8u; } // exact integers in the f32 accumulators
// cpw: window chunks a wavefront takes in a row (>= 1; cpw x wchunk must stay within pairs_mfma_max_wchunk())
void launch_pairs_mfma(const Pileup &P, const uint32_t *tiles, uint32_t ntiles, uint32_t wchunk, unsigned long long *subst,
					   unsigned long long *homologs, hipStream_t st, uint32_t cpw, unsigned long long *clk)
{
	if (!ntiles || !P.W) return;
	if (!cpw) cpw = 1;
	while (cpw > 1 && (uint64_t)cpw * wchunk > pairs_mfma_max_wchunk()) cpw--; // (never beyond what the f32 accumulators hold exactly)
	const uint32_t nwc = (P.W + wchunk - 1) / wchunk;
	const uint32_t groups = ((nwc + 7) / 8 + cpw - 1) / cpw; // per XCD
	dim3 grid(groups * 8 * ntiles);
	hipLaunchKernelGGL(pairs_mfma_kernel, grid, dim3(64), 0, st, P, tiles, ntiles, wchunk, nwc, cpw, subst, homologs, clk);
}

// The three planes carry '!' as 'A' (code 00), which is what revseqcmp's ((c ^ d) & 6) == 4 test sees
// (libs/revseqcmp.h:19-23) — but seqcmp compares bytes ('!' != 'A', libs/seqcmp.c:13-28): where two genomes are
// projected in the same direction, one holds '!' and the other 'A', the plane tallies miss one substitution.  Such
// positions are a handful (a genome's contig joins that lie inside homologies), so instead of two more planes
// through the whole pair grid they are listed by the projection and settled here: one block per listed '!', its
// threads the other genomes — covering homology by binary search in the genome's list, direction, the base at the
// query position behind it from the 2-bit codes, the genome's own '!' list.
__global__ __launch_bounds__(256) void bang_correct_kernel(Pileup P, QuerySrc Q, const DevHom *__restrict__ homs,
															const uint32_t *__restrict__ hom_rng, const uint32_t *__restrict__ list,
															const uint32_t *__restrict__ count, uint32_t cap,
															unsigned long long *__restrict__ subst)
{
	const uint32_t e = blockIdx.x;
	const uint32_t n = *count < cap ? *count : cap;
	if (e >= n) return;
	const uint32_t i = list[2 * e] & 0x7fffffffu, di = list[2 * e] >> 31, p = list[2 * e + 1];
	for (uint32_t j = threadIdx.x; j < P.N; j += blockDim.x) {
		if (j == i) continue;
		uint32_t lo = hom_rng[2 * j], hi = hom_rng[2 * j + 1];
		const uint32_t h0 = lo;
		while (lo < hi) { // first homology that starts beyond p
			const uint32_t mid = lo + ((hi - lo) >> 1);
			if (homs[mid].start <= p) lo = mid + 1;
			else hi = mid;
		}
		if (lo == h0) continue;
		const DevHom hm = homs[lo - 1];
		if (p - hm.start >= hm.len || (hm.rev ? 1u : 0u) != di) continue; // not covered, or the other strand: revseqcmp's view holds
		const uint32_t qpos = hm.rev ? hm.iq + (hm.len - 1u - (p - hm.start)) : hm.iq + (p - hm.start);
		const uint32_t b0 = Q.qbad_off[j], nb = Q.qbad_off[j + 1] - b0;
		const uint32_t k = lower_bound_u32(Q.qbad + b0, nb, qpos);
		if (k < nb && Q.qbad[b0 + k] == qpos) continue; // '!' against '!': equal bytes
		const uint64_t at = Q.goff[j] + qpos;
		const uint32_t code = (Q.q2[at >> 4] >> (30u - 2u * (uint32_t)(at & 15u))) & 3u;
		if (code != 0u) continue; // not 'A': the planes counted the substitution already
		const uint32_t a = i < j ? i : j, b = i < j ? j : i;
		atomicAdd(&subst[(size_t)a * P.N + b], 1ull);
	}
}
void launch_bang_correct(const Pileup &P, const QuerySrc &Q, const DevHom *homs, const uint32_t *hom_rng, const uint32_t *list,
						 const uint32_t *count, uint32_t cap, unsigned long long *subst, hipStream_t st)
{
	if (cap) hipLaunchKernelGGL(bang_correct_kernel, dim3(cap), dim3(256), 0, st, P, Q, homs, hom_rng, list, count, cap, subst);
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
// The pileup equals compare(list, list) of process.cxx:566-611 only for lists that are sorted by projected
// start, pairwise disjoint and inside the reference — what phase A's filter guarantees, not what a caller
// may install.  One block per genome, every entry against its predecessor; *bad is raised on a violation.
__global__ __launch_bounds__(256) void check_lists_kernel(const DevHom *__restrict__ homs, const uint32_t *__restrict__ hom_rng,
														   uint32_t L, uint32_t *__restrict__ bad)
{
	const uint32_t g = blockIdx.x, b = hom_rng[2 * g], e = hom_rng[2 * g + 1];
	for (uint32_t t = b + threadIdx.x; t < e; t += blockDim.x) {
		const DevHom h = homs[t];
		bool wrong = (uint64_t)h.start + h.len > L;
		if (t > b) {
			const DevHom p = homs[t - 1];
			wrong = wrong || h.start < p.start + p.len;
		}
		if (wrong) *bad = 1;
	}
}
void launch_check_lists(const DevHom *homs, const uint32_t *hom_rng, uint32_t N, uint32_t L, uint32_t *bad, hipStream_t st)
{
	if (N) hipLaunchKernelGGL(check_lists_kernel, dim3(N), dim3(256), 0, st, homs, hom_rng, L, bad);
}

void launch_symmetrise(uint32_t N, unsigned long long *a, unsigned long long *b, hipStream_t st)
{
	uint64_t n = (uint64_t)N * N;
	if (n) hipLaunchKernelGGL(symmetrise_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, N, a, b);
}

// genomes [g0, g1) / genome tiles [tg0, tg1) of PROJ_TG genomes: the whole pileup, or the part
// of it whose lists are ready (phase A projects eagerly, group by group)
void launch_tile_index(const Pileup &P, const QuerySrc &Q, const DevHom *homs, const uint32_t *hom_rng, uint32_t *first, uint32_t g0,
					   uint32_t g1, hipStream_t st, uint32_t *zero_flags)
{
	uint32_t ntw = (P.W + PROJ_TW - 1) / PROJ_TW;
	if (g1 > P.N) g1 = P.N;
	if (!ntw || g0 >= g1) return;
	uint64_t entries = (uint64_t)(g1 - g0) * ntw;
	hipLaunchKernelGGL(tile_index_kernel, dim3((uint32_t)((entries + 255) / 256)), dim3(256), 0, st, P, Q, homs, hom_rng, first,
					   g0, g1, zero_flags);
}
void project_resident_blocks(int out[2])
{
	int dev = 0;
	hipDeviceProp_t prop;
	if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
	for (int five = 0; five < 2; five++) {
		int per_cu = 0;
		const void *fn = five ? (const void *)project_kernel<true> : (const void *)project_kernel<false>;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
		out[five] = per_cu * prop.multiProcessorCount;
	}
}
void launch_project(const Pileup &P, bool five_planes, const QuerySrc &Q, const DevHom *homs,
					const uint32_t *hom_rng, const uint32_t *first, uint32_t *bang_flag, uint32_t tg0, uint32_t tg1,
					hipStream_t st, uint32_t *bang_list, uint32_t bang_cap, const int *resident_blocks)
{
	uint32_t ntw = (P.W + PROJ_TW - 1) / PROJ_TW, ntg = P.Npad / PROJ_TG;
	if (tg1 > ntg) tg1 = ntg;
	if (!ntw || tg0 >= tg1) return;
	const uint32_t ntiles = ntw * (tg1 - tg0);
	// as many blocks as the chip holds at once (LDS: five per CU with three planes), each taking tile after tile
	const int res = std::max(1, resident_blocks[five_planes ? 1 : 0]);
	dim3 grid(std::min<uint32_t>(ntiles, (uint32_t)res));
	if (five_planes)
		hipLaunchKernelGGL(project_kernel<true>, grid, dim3(256), 0, st, P, Q, homs, hom_rng, first, bang_flag, tg0, ntiles, (uint32_t *)nullptr, 0u);
	else
		hipLaunchKernelGGL(project_kernel<false>, grid, dim3(256), 0, st, P, Q, homs, hom_rng, first, bang_flag, tg0, ntiles, bang_list, bang_cap);
}
uint32_t project_genomes_per_tile() { return PROJ_TG; }
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

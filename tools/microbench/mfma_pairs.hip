// mfma_pairs.hip — phase B's pair tallies as a matrix-core contraction: does it pay?
//
// The pileup's tallies are a contraction over reference positions:
//     homologs(i,j) = sum_p V_i V_j
//     matches(i,j)  = sum_p V_i V_j (1 + a_i a_j)(1 + b_i b_j) / 4      a = (-1)^N0, b = (-1)^N1
//                   = (sum_p [V V' + Va Va' + Vb Vb' + Vab Vab']) / 4
// i.e. four channels of values in {-1, 0, +1} — exactly representable in FP4 (E2M1: +1 = 0b0010, sign = bit 3),
// the densest matrix-core format of gfx950 (v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 per instruction at the cycles
// of the bf16 32x32x16 form).  f32 accumulators hold integers exactly below 2^24.
//
// The operands are made from the bit planes in registers: a plane word holds 32 positions; position class d
// (positions = d mod 4) of the word becomes one dword of 8 nibbles by one rotation and one mask, V landing on
// nibble bit 1 and the sign on nibble bit 3.  The order of K inside an instruction is free as long as both operands
// use the same one — they do, the expansion is the same code.  Lane l of a wave holds genome l & 31 of a group of
// 32; lanes 0..31 take window w, lanes 32..63 window w + 1 (the instruction's two K blocks of 32).
//
// This program checks the formulation bit for bit against popcounts on random planes, and times it next to the
// product's VALU kernel (copied below) at C3's and C4's shapes.
//
// hipcc -O3 --offload-arch=gfx950 -o mfma_pairs mfma_pairs.hip ; ./mfma_pairs [N] [L]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct Planes {
	const uint32_t *V, *N0, *N1; // [W][Npad]
	uint32_t W, N, Npad;
};

static __device__ __forceinline__ uint32_t rotl(uint32_t x, uint32_t s) { return __builtin_rotateleft32(x, s); }
static __device__ __forceinline__ uint32_t rotr(uint32_t x, uint32_t s) { return __builtin_rotateright32(x, s); }

// channel 0: V on nibble bit 1 (E2M1 +1.0), one dword per position class
static __device__ __forceinline__ void expand_v(uint32_t V, uint32_t o[4])
{
	o[0] = rotl(V, 1) & 0x22222222u;
	o[1] = V & 0x22222222u;
	o[2] = rotr(V, 1) & 0x22222222u;
	o[3] = rotr(V, 2) & 0x22222222u;
}
// a signed channel: the same magnitudes, the sign plane on nibble bit 3 (-0 where V is 0: harmless)
static __device__ __forceinline__ void expand_s(uint32_t S, const uint32_t v[4], uint32_t o[4])
{
	o[0] = v[0] | (rotl(S, 3) & 0x88888888u);
	o[1] = v[1] | (rotl(S, 2) & 0x88888888u);
	o[2] = v[2] | (rotl(S, 1) & 0x88888888u);
	o[3] = v[3] | (S & 0x88888888u);
}
static __device__ __forceinline__ v16f mfma_fp4(const uint32_t a[4], const uint32_t b[4], v16f c)
{
	const v8i va = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0};
	const v8i vb = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], 0, 0, 0, 0};
#ifdef USE_SCALE_ONE
	return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#else
	return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0, 0, 0); // unscaled form of the instruction
#endif
}

// One wavefront per (tile, window chunk); a tile is GI x GJ groups of 32 genomes.
// DIAG (square tiles on the diagonal of the pair grid): the sub-tiles below the diagonal are left out.
template <int GI, int GJ, bool DIAG>
static __device__ __forceinline__ void pairs_mfma_body(const Planes &P, uint32_t ti, uint32_t tj, uint32_t wc, uint32_t wchunk,
														unsigned long long *__restrict__ subst, unsigned long long *__restrict__ homologs)
{
	const uint32_t lane = threadIdx.x & 63u, gl = lane & 31u, half = lane >> 5;
	const uint32_t w0 = wc * wchunk, w1 = (w0 + wchunk < P.W) ? w0 + wchunk : P.W;
	constexpr int NG = GI + GJ;
	uint32_t col[NG]; // genome column of this lane in each group: the tile's i groups, then its j groups
#pragma unroll
	for (int g = 0; g < GI; g++) col[g] = (ti * GI + g) * 32u + gl;
#pragma unroll
	for (int g = 0; g < GJ; g++) col[GI + g] = (tj * GJ + g) * 32u + gl;
	v16f acc_h[GI][GJ], acc_t[GI][GJ];
#pragma unroll
	for (int a = 0; a < GI; a++)
#pragma unroll
		for (int b = 0; b < GJ; b++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc_h[a][b][r] = acc_t[a][b][r] = 0.f;
#define NEED(a, b) (!DIAG || (a) <= (b))

	uint32_t nv[NG], n0[NG], n1[NG];
	auto load = [&](uint32_t w) {
		const uint32_t wl = w + half;
		const bool ok = wl < w1;
		const size_t row = (size_t)(ok ? wl : w0) * P.Npad;
#pragma unroll
		for (int g = 0; g < NG; g++) {
			nv[g] = ok ? P.V[row + col[g]] : 0u;
			n0[g] = P.N0[row + col[g]];
			n1[g] = P.N1[row + col[g]];
		}
	};
	if (w0 < w1) load(w0);
	for (uint32_t w = w0; w < w1; w += 2) {
		uint32_t cv[NG], c0[NG], c1[NG];
#pragma unroll
		for (int g = 0; g < NG; g++) {
			cv[g] = nv[g];
			c0[g] = n0[g];
			c1[g] = n1[g];
		}
		if (w + 2 < w1) load(w + 2);
		uint32_t vd[NG][4], op[NG][4];
#pragma unroll
		for (int g = 0; g < NG; g++) expand_v(cv[g], vd[g]);
#pragma unroll
		for (int a = 0; a < GI; a++)
#pragma unroll
			for (int b = 0; b < GJ; b++)
				if (NEED(a, b)) acc_h[a][b] = mfma_fp4(vd[a], vd[GI + b], acc_h[a][b]);
#pragma unroll
		for (int c = 0; c < 3; c++) {
#pragma unroll
			for (int g = 0; g < NG; g++) expand_s(c == 0 ? c0[g] : c == 1 ? c1[g] : (c0[g] ^ c1[g]), vd[g], op[g]);
#pragma unroll
			for (int a = 0; a < GI; a++)
#pragma unroll
				for (int b = 0; b < GJ; b++)
					if (NEED(a, b)) acc_t[a][b] = mfma_fp4(op[a], op[GI + b], acc_t[a][b]);
		}
	}
	// C/D layout of the 32x32 forms: column = lane & 31 (the B operand's row: genome j), row = (r & 3) + 8 (r >> 2) +
	// 4 (lane >> 5) (the A operand's row: genome i)
#pragma unroll
	for (int a = 0; a < GI; a++)
#pragma unroll
		for (int b = 0; b < GJ; b++) {
			if (!NEED(a, b)) continue;
			const uint32_t j = (tj * GJ + b) * 32u + gl;
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const uint32_t i = (ti * GI + a) * 32u + (r & 3) + 8 * (r >> 2) + 4 * half;
				const int h = (int)acc_h[a][b][r], t = (int)acc_t[a][b][r];
				if (i < j && j < P.N && h) {
					atomicAdd(&homologs[(size_t)i * P.N + j], (unsigned long long)h);
					const int s = (3 * h - t) >> 2; // matches = (h + t) / 4
					if (s) atomicAdd(&subst[(size_t)i * P.N + j], (unsigned long long)s);
				}
			}
		}
}
#undef NEED
template <int GI, int GJ>
__global__ __launch_bounds__(64) void pairs_mfma(Planes P, const uint32_t *__restrict__ tiles, uint32_t ntiles, uint32_t wchunk,
												  uint32_t nwc, unsigned long long *__restrict__ subst,
												  unsigned long long *__restrict__ homologs)
{
	const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
	const uint32_t tile = local % ntiles;
	const uint32_t wc = (local / ntiles) * 8u + xcd;
	if (wc >= nwc) return;
	const uint32_t ti = tiles[tile] >> 16, tj = tiles[tile] & 0xffffu;
	if (GI == GJ && ti == tj) pairs_mfma_body<GI, GJ, true>(P, ti, tj, wc, wchunk, subst, homologs);
	else pairs_mfma_body<GI, GJ, false>(P, ti, tj, wc, wchunk, subst, homologs);
}

// ── second version: what the product would run ──
// Two waves per SIMD (one wave's expansion issues while the other's matrix instructions run), plane words fetched by
// buffer loads (one 32-bit offset per lane, the tile's genome columns in the scalar offset and the immediate), the
// loop unrolled by two steps so that no register moves separate them.  Window chunks are even (host), so a step's
// two windows lie in the chunk or — at the end of the planes — outside the buffer, where a buffer load returns 0.
struct PlaneWords {
	uint32_t v, a, b;
};
#ifdef CLOCK_DIAG
__device__ unsigned long long g_clk[4]; // sum of shader cycles, of 100 MHz ticks, waves, steps
#endif
template <int GI, int GJ, bool DIAG>
static __device__ __forceinline__ void pairs_mfma2_body(const Planes &P, uint32_t ti, uint32_t tj, uint32_t wc, uint32_t wchunk,
														 unsigned long long *__restrict__ subst, unsigned long long *__restrict__ homologs)
{
	const uint32_t lane = threadIdx.x & 63u, gl = lane & 31u, half = lane >> 5;
	const uint32_t w0 = wc * wchunk, w1 = (w0 + wchunk < P.W) ? w0 + wchunk : P.W;
	constexpr int NG = GI + GJ;
	const uint32_t plane_bytes = P.W * P.Npad * 4u;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)P.V, 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)P.N0, 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)P.N1, 0, plane_bytes, 0x00020000);
	const uint32_t so_i = ti * GI * 128u, so_j = tj * GJ * 128u; // byte offset of the tile's first i / j genome in a row
	uint32_t off = ((w0 + half) * P.Npad + gl) * 4u;             // this lane's word of its window's row
	const uint32_t step = 2u * P.Npad * 4u;
	v16f acc_h[GI][GJ], acc_t[GI][GJ];
#pragma unroll
	for (int a = 0; a < GI; a++)
#pragma unroll
		for (int b = 0; b < GJ; b++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc_h[a][b][r] = acc_t[a][b][r] = 0.f;
#define NEED(a, b) (!DIAG || (a) <= (b))
	auto load = [&](PlaneWords (&x)[NG]) {
#pragma unroll
		for (int g = 0; g < NG; g++) {
			const uint32_t so = g < GI ? so_i : so_j, im = (uint32_t)(g < GI ? g : g - GI) * 128u;
			x[g].v = __builtin_amdgcn_raw_buffer_load_b32(rv, off + im, so, 0);
			x[g].a = __builtin_amdgcn_raw_buffer_load_b32(ra, off + im, so, 0);
			x[g].b = __builtin_amdgcn_raw_buffer_load_b32(rb, off + im, so, 0);
		}
		off += step;
	};
	auto compute = [&](const PlaneWords (&x)[NG]) {
		uint32_t vd[NG][4], op[NG][4];
#pragma unroll
		for (int g = 0; g < NG; g++) expand_v(x[g].v, vd[g]);
#pragma unroll
		for (int a = 0; a < GI; a++)
#pragma unroll
			for (int b = 0; b < GJ; b++)
				if (NEED(a, b)) acc_h[a][b] = mfma_fp4(vd[a], vd[GI + b], acc_h[a][b]);
#pragma unroll
		for (int c = 0; c < 3; c++) {
#pragma unroll
			for (int g = 0; g < NG; g++) expand_s(c == 0 ? x[g].a : c == 1 ? x[g].b : (x[g].a ^ x[g].b), vd[g], op[g]);
#pragma unroll
			for (int a = 0; a < GI; a++)
#pragma unroll
				for (int b = 0; b < GJ; b++)
					if (NEED(a, b)) acc_t[a][b] = mfma_fp4(op[a], op[GI + b], acc_t[a][b]);
		}
	};
	PlaneWords x0[NG], x1[NG];
	load(x0);
	for (uint32_t w = w0; w < w1; w += 4) {
		load(x1);
		compute(x0);
		load(x0);
		if (w + 2 < w1) compute(x1);
	}
#pragma unroll
	for (int a = 0; a < GI; a++)
#pragma unroll
		for (int b = 0; b < GJ; b++) {
			if (!NEED(a, b)) continue;
			const uint32_t j = (tj * GJ + b) * 32u + gl;
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const uint32_t i = (ti * GI + a) * 32u + (r & 3) + 8 * (r >> 2) + 4 * half;
				const int h = (int)acc_h[a][b][r], t = (int)acc_t[a][b][r];
				if (i < j && j < P.N && h) {
					atomicAdd(&homologs[(size_t)i * P.N + j], (unsigned long long)h);
					const int s = (3 * h - t) >> 2; // matches = (h + t) / 4
					if (s) atomicAdd(&subst[(size_t)i * P.N + j], (unsigned long long)s);
				}
			}
		}
#undef NEED
}
template <int GI, int GJ>
__global__ __launch_bounds__(64, 2) void pairs_mfma2(Planes P, const uint32_t *__restrict__ tiles, uint32_t ntiles, uint32_t wchunk,
													  uint32_t nwc, unsigned long long *__restrict__ subst,
													  unsigned long long *__restrict__ homologs)
{
	const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
	const uint32_t tile = local % ntiles;
	const uint32_t wc = (local / ntiles) * 8u + xcd;
	if (wc >= nwc) return;
	const uint32_t ti = tiles[tile] >> 16, tj = tiles[tile] & 0xffffu;
	if (GI == GJ && ti == tj) pairs_mfma2_body<GI, GJ, true>(P, ti, tj, wc, wchunk, subst, homologs);
	else pairs_mfma2_body<GI, GJ, false>(P, ti, tj, wc, wchunk, subst, homologs);
}

// ── third version: NB register sets of plane words in flight (a set is refilled as soon as its step has expanded it and
// is used NB steps later), no branch in the loop (chunks are multiples of 2 NB windows; beyond the planes' end the
// buffer loads return 0), and — SCHED — the request to the compiler's scheduler to deal the expansion's vector
// instructions out between the matrix instructions instead of bunching each kind.
template <int GI, int GJ, bool DIAG, int NB, int SCHED>
static __device__ __forceinline__ void pairs_mfma3_body(const Planes &P, uint32_t ti, uint32_t tj, uint32_t wc, uint32_t wchunk,
														 unsigned long long *__restrict__ subst, unsigned long long *__restrict__ homologs)
{
	const uint32_t lane = threadIdx.x & 63u, gl = lane & 31u, half = lane >> 5;
	const uint32_t w0 = wc * wchunk, w1 = (w0 + wchunk < P.W) ? w0 + wchunk : P.W;
	constexpr int NG = GI + GJ;
	const uint32_t plane_bytes = P.W * P.Npad * 4u;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)P.V, 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)P.N0, 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)P.N1, 0, plane_bytes, 0x00020000);
	const uint32_t so_i = ti * GI * 128u, so_j = tj * GJ * 128u;
	uint32_t off = ((w0 + half) * P.Npad + gl) * 4u;
	const uint32_t step = 2u * P.Npad * 4u;
	v16f acc_h[GI][GJ], acc_t[GI][GJ];
#pragma unroll
	for (int a = 0; a < GI; a++)
#pragma unroll
		for (int b = 0; b < GJ; b++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc_h[a][b][r] = acc_t[a][b][r] = 0.f;
#define NEED(a, b) (!DIAG || (a) <= (b))
	auto load = [&](PlaneWords (&x)[NG]) {
#pragma unroll
		for (int g = 0; g < NG; g++) {
			const uint32_t so = g < GI ? so_i : so_j, im = (uint32_t)(g < GI ? g : g - GI) * 128u;
			x[g].v = __builtin_amdgcn_raw_buffer_load_b32(rv, off + im, so, 0);
			x[g].a = __builtin_amdgcn_raw_buffer_load_b32(ra, off + im, so, 0);
			x[g].b = __builtin_amdgcn_raw_buffer_load_b32(rb, off + im, so, 0);
		}
		off += step;
	};
	constexpr int NMF = (DIAG ? (GI * (GI + 1)) / 2 : GI * GJ); // matrix instructions per channel
	uint32_t sink = 0; // (experiments: SCHED 2 = the expansion alone, 3 = the matrix instructions alone)
	auto compute = [&](PlaneWords (&x)[NG]) {
		if (SCHED == 4 || SCHED == 5) {
			// two phases per step — all operands first, then all matrix instructions, at raised priority — so that one
			// wavefront's expansion can issue under the other's matrix instructions
			uint32_t opa[4][NG][4];
#pragma unroll
			for (int g = 0; g < NG; g++) {
				expand_v(x[g].v, opa[0][g]);
				expand_s(x[g].a, opa[0][g], opa[1][g]);
				expand_s(x[g].b, opa[0][g], opa[2][g]);
				expand_s(x[g].a ^ x[g].b, opa[0][g], opa[3][g]);
			}
			load(x);
			if (SCHED == 5) __builtin_amdgcn_sched_barrier(0);
			__builtin_amdgcn_s_setprio(2);
#pragma unroll
			for (int c = 0; c < 4; c++)
#pragma unroll
				for (int a = 0; a < GI; a++)
#pragma unroll
					for (int b = 0; b < GJ; b++)
						if (NEED(a, b)) {
							if (c == 0) acc_h[a][b] = mfma_fp4(opa[0][a], opa[0][GI + b], acc_h[a][b]);
							else acc_t[a][b] = mfma_fp4(opa[c][a], opa[c][GI + b], acc_t[a][b]);
						}
			__builtin_amdgcn_s_setprio(0);
			if (SCHED == 5) __builtin_amdgcn_sched_barrier(0);
			return;
		}
		uint32_t vd[NG][4], op[NG][4];
		if (SCHED == 3) {
#pragma unroll
			for (int g = 0; g < NG; g++)
#pragma unroll
				for (int d = 0; d < 4; d++) vd[g][d] = op[g][d] = x[g].v & 0x2a2a2a2au;
#pragma unroll
			for (int c = 0; c < 4; c++)
#pragma unroll
				for (int a = 0; a < GI; a++)
#pragma unroll
					for (int b = 0; b < GJ; b++)
						if (NEED(a, b)) acc_t[a][b] = mfma_fp4(op[a], op[GI + b], acc_t[a][b]);
			load(x);
			return;
		}
#pragma unroll
		for (int g = 0; g < NG; g++) expand_v(x[g].v, vd[g]);
#pragma unroll
		for (int a = 0; a < GI; a++)
#pragma unroll
			for (int b = 0; b < GJ; b++)
				if (NEED(a, b)) {
					if (SCHED == 2) sink ^= vd[a][0] ^ vd[a][1] ^ vd[a][2] ^ vd[a][3] ^ vd[GI + b][0] ^ vd[GI + b][1] ^ vd[GI + b][2] ^ vd[GI + b][3];
					else acc_h[a][b] = mfma_fp4(vd[a], vd[GI + b], acc_h[a][b]);
				}
#pragma unroll
		for (int c = 0; c < 3; c++) {
#pragma unroll
			for (int g = 0; g < NG; g++) expand_s(c == 0 ? x[g].a : c == 1 ? x[g].b : (x[g].a ^ x[g].b), vd[g], op[g]);
#pragma unroll
			for (int a = 0; a < GI; a++)
#pragma unroll
				for (int b = 0; b < GJ; b++)
					if (NEED(a, b)) {
						if (SCHED == 2) sink ^= op[a][0] ^ op[a][1] ^ op[a][2] ^ op[a][3] ^ op[GI + b][0] ^ op[GI + b][1] ^ op[GI + b][2] ^ op[GI + b][3];
						else acc_t[a][b] = mfma_fp4(op[a], op[GI + b], acc_t[a][b]);
					}
		}
		load(x); // the set is free again: its words for the step NB from now
		if (SCHED == 1) {
			// per matrix instruction its share of the step's vector instructions (~29 per group and step)
			constexpr int PER = (29 * NG + 4 * NMF - 1) / (4 * NMF);
#pragma unroll
			for (int i = 0; i < 4 * NMF; i++) {
				__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
				__builtin_amdgcn_sched_group_barrier(0x002, PER, 0); // PER VALU
				if (i % (4 * NMF / 4) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 3 * NG / 4, 0); // a quarter of the loads
			}
		}
	};
	PlaneWords x[NB][NG];
#ifdef CLOCK_DIAG
	const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
	for (int k = 0; k < NB; k++) load(x[k]);
	for (uint32_t w = w0; w < w1; w += 2 * NB) {
#pragma unroll
		for (int k = 0; k < NB; k++) compute(x[k]);
	}
	if (SCHED == 2 && sink == 0x12345u) acc_h[0][0][0] = 1.f; // keeps the expansion alive
#ifdef CLOCK_DIAG
	if (lane == 0) {
		atomicAdd(&g_clk[0], __builtin_amdgcn_s_memtime() - c0);
		atomicAdd(&g_clk[1], __builtin_amdgcn_s_memrealtime() - r0);
		atomicAdd(&g_clk[2], 1ull);
		atomicAdd(&g_clk[3], (unsigned long long)((w1 - w0 + 2 * NB - 1) / (2 * NB) * NB));
	}
#endif
#pragma unroll
	for (int a = 0; a < GI; a++)
#pragma unroll
		for (int b = 0; b < GJ; b++) {
			if (!NEED(a, b)) continue;
			const uint32_t j = (tj * GJ + b) * 32u + gl;
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const uint32_t i = (ti * GI + a) * 32u + (r & 3) + 8 * (r >> 2) + 4 * half;
				const int h = (int)acc_h[a][b][r], t = (int)acc_t[a][b][r];
				if (i < j && j < P.N && h) {
					atomicAdd(&homologs[(size_t)i * P.N + j], (unsigned long long)h);
					const int s = (3 * h - t) >> 2; // matches = (h + t) / 4
					if (s) atomicAdd(&subst[(size_t)i * P.N + j], (unsigned long long)s);
				}
			}
		}
#undef NEED
}
template <int GI, int GJ, int NB, int SCHED>
__global__ __launch_bounds__(64, 2) void pairs_mfma3(Planes P, const uint32_t *__restrict__ tiles, uint32_t ntiles, uint32_t wchunk,
													  uint32_t nwc, unsigned long long *__restrict__ subst,
													  unsigned long long *__restrict__ homologs)
{
	const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
	const uint32_t tile = local % ntiles;
	const uint32_t wc = (local / ntiles) * 8u + xcd;
	if (wc >= nwc) return;
	const uint32_t ti = tiles[tile] >> 16, tj = tiles[tile] & 0xffffu;
	if (GI == GJ && ti == tj) pairs_mfma3_body<GI, GJ, true, NB, SCHED>(P, ti, tj, wc, wchunk, subst, homologs);
	else pairs_mfma3_body<GI, GJ, false, NB, SCHED>(P, ti, tj, wc, wchunk, subst, homologs);
}

// the same body with a wavefront per SIMD (512 registers): the room a 64 x 128 tile's 256 accumulators need — a quarter
// fewer vector instructions per matrix instruction than 64 x 64
template <int GI, int GJ, int NB, int SCHED>
__global__ __launch_bounds__(64, 1) void pairs_mfma3w(Planes P, const uint32_t *__restrict__ tiles, uint32_t ntiles, uint32_t wchunk,
													   uint32_t nwc, unsigned long long *__restrict__ subst,
													   unsigned long long *__restrict__ homologs)
{
	const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
	const uint32_t tile = local % ntiles;
	const uint32_t wc = (local / ntiles) * 8u + xcd;
	if (wc >= nwc) return;
	const uint32_t ti = tiles[tile] >> 16, tj = tiles[tile] & 0xffffu;
	pairs_mfma3_body<GI, GJ, false, NB, SCHED>(P, ti, tj, wc, wchunk, subst, homologs);
}

// the product's kernel (csrc/pileup_kernels.hip: pairs_kernel<false>), for the same-box comparison
static const uint32_t PAIR_IG = 16, PAIR_JT = 64;
__global__ __launch_bounds__(64) void pairs_valu(Planes P, const uint32_t *__restrict__ tiles, uint32_t ntiles, uint32_t wchunk,
												  uint32_t nwc, unsigned long long *__restrict__ subst,
												  unsigned long long *__restrict__ homologs)
{
	const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
	const uint32_t tile = local % ntiles;
	const uint32_t wc = (local / ntiles) * 8u + xcd;
	if (wc >= nwc) return;
	const uint32_t ig = tiles[tile] >> 16, jt = tiles[tile] & 0xffffu;
	const uint32_t i0 = ig * PAIR_IG;
	const uint32_t j = jt * PAIR_JT + (threadIdx.x & 63u);
	const uint32_t w0 = wc * wchunk;
	const uint32_t w1 = (w0 + wchunk < P.W) ? w0 + wchunk : P.W;
	const uint32_t *__restrict__ pV = P.V, *__restrict__ p0 = P.N0, *__restrict__ p1 = P.N1;
	uint32_t acc_h[PAIR_IG], acc_s[PAIR_IG];
#pragma unroll
	for (uint32_t t = 0; t < PAIR_IG; t++) acc_h[t] = acc_s[t] = 0;
	uint32_t vj_n = 0, aj_n = 0, bj_n = 0;
	if (w0 < w1) {
		const size_t row = (size_t)w0 * P.Npad;
		vj_n = pV[row + j];
		aj_n = p0[row + j];
		bj_n = p1[row + j];
	}
	for (uint32_t w = w0; w < w1; w++) {
		const size_t row = (size_t)w * P.Npad;
		const uint32_t vj = vj_n, aj = aj_n, bj = bj_n;
		{
			const size_t rn = (size_t)(w + 1 < w1 ? w + 1 : w) * P.Npad;
			vj_n = pV[rn + j];
			aj_n = p0[rn + j];
			bj_n = p1[rn + j];
		}
#pragma unroll
		for (uint32_t t = 0; t < PAIR_IG; t++) {
			const size_t oi = row + i0 + t;
			const uint32_t both = pV[oi] & vj;
			const uint32_t diff = (p0[oi] ^ aj) | (p1[oi] ^ bj);
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

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd()
{
	rng_state ^= rng_state << 13;
	rng_state ^= rng_state >> 7;
	rng_state ^= rng_state << 17;
	return (uint32_t)(rng_state >> 16);
}

struct Run {
	uint32_t N, Npad, W;
	uint32_t *dV, *d0, *d1;
	unsigned long long *ds, *dh;
	uint32_t *dtiles;
	int n_cu;
};

static std::vector<uint32_t> make_tiles(uint32_t N, uint32_t ti_sz, uint32_t tj_sz)
{
	std::vector<uint32_t> t;
	for (uint32_t a = 0; a < (N + ti_sz - 1) / ti_sz; a++)
		for (uint32_t b = 0; b < (N + tj_sz - 1) / tj_sz; b++)
			if ((uint64_t)a * ti_sz < (uint64_t)b * tj_sz + tj_sz - 1) t.push_back((a << 16) | b);
	return t;
}

// the launch shape of the product (phylo_abi.hip: compare_pileup), rows of three planes
static uint32_t choose_wchunk(const Run &R, uint32_t ntiles, uint32_t slots_per_cu)
{
	uint32_t row_bytes = 3u * R.Npad * 4u;
	uint32_t l2_fit = std::max<uint32_t>(64, (3u << 20) / row_bytes);
	uint32_t want = std::max<uint32_t>(1, ((uint32_t)R.n_cu * slots_per_cu) / ntiles);
	uint32_t wchunk = std::max<uint32_t>(64, (R.W + want - 1) / want);
	wchunk = std::min(wchunk, l2_fit);
	const uint32_t groups = (R.W + 8u * wchunk - 1) / (8u * wchunk);
	wchunk = std::max<uint32_t>(1, (R.W + 8u * groups - 1) / (8u * groups));
	wchunk = (wchunk + 11) / 12 * 12; // the matrix-core kernels take two windows per step, two or three steps per trip
	return wchunk;
}

template <class K> static float time_kernel(const Run &R, K kern, const std::vector<uint32_t> &tiles, uint32_t wchunk, int reps)
{
	Planes P = {R.dV, R.d0, R.d1, R.W, R.N, R.Npad};
	CK(hipMemcpy(R.dtiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
	const uint32_t nwc = (R.W + wchunk - 1) / wchunk;
	dim3 grid(((nwc + 7) / 8) * 8 * (uint32_t)tiles.size());
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipMemset(R.ds, 0, (size_t)R.N * R.N * 8));
	CK(hipMemset(R.dh, 0, (size_t)R.N * R.N * 8));
	hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, P, R.dtiles, (uint32_t)tiles.size(), wchunk, nwc, R.ds, R.dh);
	CK(hipDeviceSynchronize());
	float best = 1e30f, sum = 0;
	for (int r = 0; r < reps; r++) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, P, R.dtiles, (uint32_t)tiles.size(), wchunk, nwc, R.ds, R.dh);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		best = std::min(best, ms);
		sum += ms;
	}
	CK(hipGetLastError());
	printf("    grid %u waves, %zu tiles x %u chunks of %u windows: avg %.3f ms, best %.3f ms\n", grid.x, tiles.size(), nwc, wchunk,
		   sum / reps, best);
#ifdef CLOCK_DIAG
	{
		unsigned long long h[4];
		CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof h));
		if (h[2]) printf("    in-kernel: %.0f MHz, %.0f shader cycles per step and wave (%llu waves, %llu steps)\n", 100.0 * h[0] / h[1], (double)h[0] / h[3], h[2], h[3]);
		unsigned long long z[4] = {0, 0, 0, 0};
		CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof z));
	}
#endif
	return sum / reps;
}

template <class K> static bool check_kernel(const Run &R, K kern, const std::vector<uint32_t> &tiles, uint32_t wchunk,
											const std::vector<uint64_t> &ref_s, const std::vector<uint64_t> &ref_h, const char *name)
{
	Planes P = {R.dV, R.d0, R.d1, R.W, R.N, R.Npad};
	CK(hipMemcpy(R.dtiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
	const uint32_t nwc = (R.W + wchunk - 1) / wchunk;
	dim3 grid(((nwc + 7) / 8) * 8 * (uint32_t)tiles.size());
	CK(hipMemset(R.ds, 0, (size_t)R.N * R.N * 8));
	CK(hipMemset(R.dh, 0, (size_t)R.N * R.N * 8));
	hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, P, R.dtiles, (uint32_t)tiles.size(), wchunk, nwc, R.ds, R.dh);
	CK(hipDeviceSynchronize());
	std::vector<uint64_t> s((size_t)R.N * R.N), h((size_t)R.N * R.N);
	CK(hipMemcpy(s.data(), R.ds, s.size() * 8, hipMemcpyDeviceToHost));
	CK(hipMemcpy(h.data(), R.dh, h.size() * 8, hipMemcpyDeviceToHost));
	size_t bad = 0;
	for (uint32_t i = 0; i < R.N; i++)
		for (uint32_t j = i + 1; j < R.N; j++) {
			const size_t k = (size_t)i * R.N + j;
			if (s[k] != ref_s[k] || h[k] != ref_h[k]) {
				if (bad < 5)
					printf("    %s: pair (%u,%u) got (%llu,%llu), want (%llu,%llu)\n", name, i, j, (unsigned long long)s[k],
						   (unsigned long long)h[k], (unsigned long long)ref_s[k], (unsigned long long)ref_h[k]);
				bad++;
			}
		}
	printf("  check %-22s %s (%zu of %zu pairs differ)\n", name, bad ? "MISMATCH" : "bit-exact", bad, (size_t)R.N * (R.N - 1) / 2);
	return bad == 0;
}

int main(int argc, char **argv)
{
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	Run R;
	R.n_cu = prop.multiProcessorCount;
	// ── 1. exactness on random planes: N = 200 genomes (ragged: not a multiple of 64), 1001 windows (odd) ──
	{
		R.N = 200;
		R.Npad = 256;
		R.W = 1001;
		const size_t words = (size_t)R.W * R.Npad;
		std::vector<uint32_t> V(words), A(words), B(words);
		for (size_t k = 0; k < words; k++) {
			const uint32_t g = (uint32_t)(k % R.Npad);
			// coverage in runs, bases random; genome g < 8 identical to genome 0 in the bases
			V[k] = g >= R.N ? 0 : (rnd() & 7) ? (rnd() | rnd()) : (rnd() & 1 ? 0xffffffffu : 0u);
			A[k] = rnd() & V[k];
			B[k] = rnd() & V[k];
		}
		for (uint32_t w = 0; w < R.W; w++)
			for (uint32_t g = 1; g < 8; g++) {
				A[(size_t)w * R.Npad + g] = A[(size_t)w * R.Npad] & V[(size_t)w * R.Npad + g];
				B[(size_t)w * R.Npad + g] = B[(size_t)w * R.Npad] & V[(size_t)w * R.Npad + g];
			}
		std::vector<uint64_t> rs((size_t)R.N * R.N, 0), rh((size_t)R.N * R.N, 0);
		for (uint32_t i = 0; i < R.N; i++)
			for (uint32_t j = i + 1; j < R.N; j++) {
				uint64_t s = 0, h = 0;
				for (uint32_t w = 0; w < R.W; w++) {
					const size_t a = (size_t)w * R.Npad + i, b = (size_t)w * R.Npad + j;
					const uint32_t both = V[a] & V[b];
					h += __builtin_popcount(both);
					s += __builtin_popcount(both & ((A[a] ^ A[b]) | (B[a] ^ B[b])));
				}
				rs[(size_t)i * R.N + j] = s;
				rh[(size_t)i * R.N + j] = h;
			}
		CK(hipMalloc(&R.dV, words * 4));
		CK(hipMalloc(&R.d0, words * 4));
		CK(hipMalloc(&R.d1, words * 4));
		CK(hipMalloc(&R.ds, (size_t)R.N * R.N * 8));
		CK(hipMalloc(&R.dh, (size_t)R.N * R.N * 8));
		CK(hipMalloc(&R.dtiles, 1 << 20));
		CK(hipMemcpy(R.dV, V.data(), words * 4, hipMemcpyHostToDevice));
		CK(hipMemcpy(R.d0, A.data(), words * 4, hipMemcpyHostToDevice));
		CK(hipMemcpy(R.d1, B.data(), words * 4, hipMemcpyHostToDevice));
		bool ok = true;
		ok &= check_kernel(R, pairs_valu, make_tiles(R.N, 16, 64), 64, rs, rh, "VALU 16x64");
		ok &= check_kernel(R, pairs_mfma<2, 2>, make_tiles(R.N, 64, 64), 64, rs, rh, "MFMA fp4 64x64");
		ok &= check_kernel(R, pairs_mfma<2, 2>, make_tiles(R.N, 64, 64), 126, rs, rh, "MFMA fp4 64x64 c126");
		ok &= check_kernel(R, pairs_mfma<2, 4>, make_tiles(R.N, 64, 128), 64, rs, rh, "MFMA fp4 64x128");
		ok &= check_kernel(R, pairs_mfma<1, 2>, make_tiles(R.N, 32, 64), 64, rs, rh, "MFMA fp4 32x64");
		ok &= check_kernel(R, pairs_mfma2<2, 2>, make_tiles(R.N, 64, 64), 64, rs, rh, "MFMA v2 64x64");
		ok &= check_kernel(R, pairs_mfma2<2, 2>, make_tiles(R.N, 64, 64), 126, rs, rh, "MFMA v2 64x64 c126");
		ok &= check_kernel(R, pairs_mfma2<2, 2>, make_tiles(R.N, 64, 64), 1002, rs, rh, "MFMA v2 64x64 c1002");
		ok &= check_kernel(R, pairs_mfma2<1, 2>, make_tiles(R.N, 32, 64), 62, rs, rh, "MFMA v2 32x64 c62");
		ok &= check_kernel(R, pairs_mfma3<2, 2, 2, 0>, make_tiles(R.N, 64, 64), 60, rs, rh, "MFMA v3 nb2 c60");
		ok &= check_kernel(R, pairs_mfma3<2, 2, 3, 0>, make_tiles(R.N, 64, 64), 126, rs, rh, "MFMA v3 nb3 c126");
		ok &= check_kernel(R, pairs_mfma3<2, 2, 3, 1>, make_tiles(R.N, 64, 64), 1008, rs, rh, "MFMA v3 nb3 sched c1008");
		ok &= check_kernel(R, pairs_mfma3<2, 2, 2, 1>, make_tiles(R.N, 64, 64), 1008, rs, rh, "MFMA v3 nb2 sched c1008");
		ok &= check_kernel(R, pairs_mfma3w<2, 4, 3, 1>, make_tiles(R.N, 64, 128), 1008, rs, rh, "MFMA v3 64x128 nb3 sched");
		ok &= check_kernel(R, pairs_mfma3w<2, 4, 2, 0>, make_tiles(R.N, 64, 128), 60, rs, rh, "MFMA v3 64x128 nb2 c60");
		ok &= check_kernel(R, pairs_mfma3<2, 2, 3, 4>, make_tiles(R.N, 64, 64), 252, rs, rh, "MFMA v3 nb3 phases c252");
		ok &= check_kernel(R, pairs_mfma3<2, 2, 2, 5>, make_tiles(R.N, 64, 64), 252, rs, rh, "MFMA v3 nb2 phases+ c252");
		CK(hipFree(R.dV));
		CK(hipFree(R.d0));
		CK(hipFree(R.d1));
		CK(hipFree(R.ds));
		CK(hipFree(R.dh));
		if (!ok) printf("EXACTNESS FAILED\n");
	}
	// ── 2. timing at the product's shapes ──
	const uint32_t shapes[2][2] = {{256, 5000000}, {1024, 5000000}};
	for (int sidx = 0; sidx < 2; sidx++) {
		R.N = argc > 1 ? (uint32_t)atoi(argv[1]) : shapes[sidx][0];
		const uint32_t L = argc > 2 ? (uint32_t)atoi(argv[2]) : shapes[sidx][1];
		R.Npad = (R.N + 63) / 64 * 64;
		R.W = (L + 31) / 32;
		const size_t words = (size_t)R.W * R.Npad;
		printf("N = %u, L = %u (%u windows): planes 3 x %.2f GB\n", R.N, L, R.W, words * 4 / 1e9);
		std::vector<uint32_t> V(words);
		for (size_t k = 0; k < words; k++) V[k] = rnd() | rnd() | rnd(); // ~87 % covered, as the workloads are
		CK(hipMalloc(&R.dV, words * 4));
		CK(hipMalloc(&R.d0, words * 4));
		CK(hipMalloc(&R.d1, words * 4));
		CK(hipMalloc(&R.ds, (size_t)R.N * R.N * 8));
		CK(hipMalloc(&R.dh, (size_t)R.N * R.N * 8));
		CK(hipMemcpy(R.dV, V.data(), words * 4, hipMemcpyHostToDevice));
		for (size_t k = 0; k < words; k++) V[k] = rnd() & V[k];
		CK(hipMemcpy(R.d0, V.data(), words * 4, hipMemcpyHostToDevice));
		for (size_t k = 0; k < words; k++) V[k] = rnd();
		CK(hipMemcpy(R.d1, V.data(), words * 4, hipMemcpyHostToDevice));
		const int reps = R.N > 512 ? 5 : 20;
		const double pairpos = (double)R.N * (R.N - 1) / 2 * L;
		{
			auto t = make_tiles(R.N, 16, 64);
			printf("  VALU 16x64 (the product's kernel and launch shape)\n");
			float ms = time_kernel(R, pairs_valu, t, choose_wchunk(R, (uint32_t)t.size(), 128), reps);
			printf("    -> %.1f T pair-positions/s\n", pairpos / (ms * 1e-3) / 1e12);
		}
		for (uint32_t slots : {32u, 64u, 128u}) {
			auto t = make_tiles(R.N, 64, 64);
			printf("  MFMA fp4 64x64, %u wave slots per CU\n", slots);
			float ms = time_kernel(R, pairs_mfma<2, 2>, t, choose_wchunk(R, (uint32_t)t.size(), slots), reps);
			printf("    -> %.1f T pair-positions/s\n", pairpos / (ms * 1e-3) / 1e12);
		}
		for (uint32_t slots : {16u, 32u, 64u}) {
			auto t = make_tiles(R.N, 64, 128);
			printf("  MFMA fp4 64x128, %u wave slots per CU\n", slots);
			float ms = time_kernel(R, pairs_mfma<2, 4>, t, choose_wchunk(R, (uint32_t)t.size(), slots), reps);
			printf("    -> %.1f T pair-positions/s\n", pairpos / (ms * 1e-3) / 1e12);
		}
		for (uint32_t slots : {8u, 16u, 32u, 64u}) {
			auto t = make_tiles(R.N, 64, 64);
			printf("  MFMA v2 64x64 (2 waves per SIMD, buffer loads), %u wave slots per CU\n", slots);
			float ms = time_kernel(R, pairs_mfma2<2, 2>, t, choose_wchunk(R, (uint32_t)t.size(), slots), reps);
			printf("    -> %.1f T pair-positions/s\n", pairpos / (ms * 1e-3) / 1e12);
		}
		for (uint32_t slots : {8u, 16u, 32u}) {
			auto t = make_tiles(R.N, 64, 64);
			const uint32_t wch = choose_wchunk(R, (uint32_t)t.size(), slots);
			printf("  MFMA v3 64x64, %u wave slots per CU: nb2 / nb3 / nb2 sched / nb3 sched\n", slots);
			float m1 = time_kernel(R, pairs_mfma3<2, 2, 2, 0>, t, wch, reps);
			float m2 = time_kernel(R, pairs_mfma3<2, 2, 3, 0>, t, wch, reps);
			float m3 = time_kernel(R, pairs_mfma3<2, 2, 2, 1>, t, wch, reps);
			float m4 = time_kernel(R, pairs_mfma3<2, 2, 3, 1>, t, wch, reps);
			printf("    -> %.3f / %.3f / %.3f / %.3f ms\n", m1, m2, m3, m4);
			printf("  two phases per step with priorities: nb2 / nb3 / nb2 + sched_barrier / nb3 + sched_barrier\n");
			time_kernel(R, pairs_mfma3<2, 2, 2, 4>, t, wch, reps);
			time_kernel(R, pairs_mfma3<2, 2, 3, 4>, t, wch, reps);
			time_kernel(R, pairs_mfma3<2, 2, 2, 5>, t, wch, reps);
			time_kernel(R, pairs_mfma3<2, 2, 3, 5>, t, wch, reps);
			printf("  the same shape, expansion alone / matrix instructions alone (nb3; results meaningless)\n");
			time_kernel(R, pairs_mfma3<2, 2, 3, 2>, t, wch, reps);
			time_kernel(R, pairs_mfma3<2, 2, 3, 3>, t, wch, reps);
		}
		for (uint32_t slots : {4u, 8u, 16u}) {
			auto t = make_tiles(R.N, 64, 128);
			const uint32_t wch = choose_wchunk(R, (uint32_t)t.size(), slots);
			printf("  MFMA v3 64x128, one wavefront per SIMD, %u wave slots per CU: nb2 / nb3 / nb2 sched / nb3 sched\n", slots);
			float m1 = time_kernel(R, pairs_mfma3w<2, 4, 2, 0>, t, wch, reps);
			float m2 = time_kernel(R, pairs_mfma3w<2, 4, 3, 0>, t, wch, reps);
			float m3 = time_kernel(R, pairs_mfma3w<2, 4, 2, 1>, t, wch, reps);
			float m4 = time_kernel(R, pairs_mfma3w<2, 4, 3, 1>, t, wch, reps);
			printf("    -> %.3f / %.3f / %.3f / %.3f ms\n", m1, m2, m3, m4);
		}
		for (uint32_t slots : {16u, 32u}) {
			auto t = make_tiles(R.N, 32, 64);
			printf("  MFMA v2 32x64, %u wave slots per CU\n", slots);
			float ms = time_kernel(R, pairs_mfma2<1, 2>, t, choose_wchunk(R, (uint32_t)t.size(), slots), reps);
			printf("    -> %.1f T pair-positions/s\n", pairpos / (ms * 1e-3) / 1e12);
		}
		{
			auto t = make_tiles(R.N, 32, 64);
			printf("  MFMA fp4 32x64, 64 wave slots per CU\n");
			float ms = time_kernel(R, pairs_mfma<1, 2>, t, choose_wchunk(R, (uint32_t)t.size(), 64), reps);
			printf("    -> %.1f T pair-positions/s\n", pairpos / (ms * 1e-3) / 1e12);
		}
		CK(hipFree(R.dV));
		CK(hipFree(R.d0));
		CK(hipFree(R.d1));
		CK(hipFree(R.ds));
		CK(hipFree(R.dh));
		if (argc > 1) break;
	}
	return 0;
}

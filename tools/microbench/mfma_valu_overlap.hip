// mfma_valu_overlap.hip — do the matrix pipe and the vector ALU of a gfx950 SIMD run side by side?
//
// The pair kernel (csrc/pileup_kernels.hip: pairs_mfma_kernel) issues, per step and wavefront, 16 FP4 matrix instructions
// (v_mfma_scale_f32_32x32x64_f8f6f4: 32 cycles of the matrix pipe each) and ~116 vector instructions of operand expansion
// (4 cycles each): 512 + 464 cycles.  Its counters say the two overlap by a fifth.  This program measures what the hardware
// does with the plainest form of that mix: a loop of M matrix instructions on independent accumulators and V vector
// instructions on independent registers, alone, together in one wavefront (dealt out 1 : V/M by sched_group_barrier), and
// in separate wavefronts of one SIMD — shader cycles per iteration (s_memtime), 1 or 2 wavefronts per SIMD.
//
//   hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define HIPOK(x)                                                                                                       \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                           \
		if (e_ != hipSuccess) {                                                                                        \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                    \
			return 1;                                                                                                  \
		}                                                                                                              \
	} while (0)

// MODE 0: matrix only; 1: vector only; 2: both, interleaved; 3: even wavefronts matrix only, odd ones vector only
template <int MODE, int NV>
__global__ __launch_bounds__(512) void mix_kernel(uint32_t iters, uint32_t seed, unsigned long long *cycles, uint32_t *sink)
{
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6; // a block per CU: wavefronts w and w + 4 share a SIMD
	v16f acc[4];
#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
	uint32_t a[4][4], b[4][4];
#pragma unroll
	for (int g = 0; g < 4; g++)
#pragma unroll
		for (int k = 0; k < 4; k++) {
			a[g][k] = (seed * 2654435761u + lane * 40503u + g * 97u + k) & 0x22222222u;
			b[g][k] = (seed * 40503u + lane * 2654435761u + g * 31u + k) & 0x22222222u;
		}
	uint32_t x[16];
#pragma unroll
	for (int i = 0; i < 16; i++) x[i] = seed + lane * 17u + i;
	const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4u);
	const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4u);
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (uint32_t it = 0; it < iters; it++) {
		if (do_m) {
#pragma unroll
			for (int c = 0; c < 4; c++) // 16 matrix instructions, each accumulator touched every fourth
#pragma unroll
				for (int i = 0; i < 4; i++) {
					const v8i va = {(int)a[i][0], (int)a[i][1], (int)a[i][2], (int)a[i][3], 0, 0, 0, 0};
					const v8i vb = {(int)b[c][0], (int)b[c][1], (int)b[c][2], (int)b[c][3], 0, 0, 0, 0};
					acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[i], 4, 4, 0, 0, 0, 0);
				}
		}
		if (do_v) {
#pragma unroll
			for (int j = 0; j < NV; j++) { // NV vector instructions on 16 independent registers (shift / and-or, as the expansion)
				const int r = j & 15;
				if (j & 16) x[r] = (x[r] & 0x88888888u) | x[(r + 5) & 15]; // (one instruction each: v_and_or_b32, v_lshl_or_b32)
				else x[r] = (x[r] << 1) | x[(r + 3) & 15];
			}
		}
		if (MODE == 2) {
#pragma unroll
			for (int i = 0; i < 16; i++) {
				__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
				__builtin_amdgcn_sched_group_barrier(0x002, (NV + 15) / 16, 0);
			}
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	uint32_t s = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) s ^= x[i];
#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int r = 0; r < 16; r++) s ^= (uint32_t)acc[i][r];
	if (s == 0x12345u) sink[0] = s;
	if (lane == 0) atomicAdd(cycles, t1 - t0);
}

// The pair kernel's step without its loads: the operands of 4 groups of 32 genomes expanded from plane words held in
// registers (turned by a bit every iteration), 16 matrix instructions on them — the same code and the same dealing-out as
// pairs_mfma_body::compute.  VARIANT 1: the expansions of a channel all before its matrix instructions, in program order
// (no sched_group_barrier); 2: every group's operands expanded one step ahead of the matrix instructions that use them
static __device__ __forceinline__ void ex_v(uint32_t V, uint32_t o[4])
{
	o[0] = (V << 1) & 0x22222222u, o[1] = V & 0x22222222u, o[2] = (V >> 1) & 0x22222222u, o[3] = (V >> 2) & 0x22222222u;
}
static __device__ __forceinline__ void ex_s(uint32_t S, const uint32_t v[4], uint32_t o[4])
{
	o[0] = v[0] | ((S << 3) & 0x88888888u), o[1] = v[1] | ((S << 2) & 0x88888888u), o[2] = v[2] | ((S << 1) & 0x88888888u), o[3] = v[3] | (S & 0x88888888u);
}
static __device__ __forceinline__ v16f mm(const uint32_t a[4], const uint32_t b[4], v16f c)
{
	const v8i va = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0};
	const v8i vb = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], 0, 0, 0, 0};
	return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0, 0, 0);
}
template <int VARIANT>
__global__ __launch_bounds__(512) void step_kernel(uint32_t iters, uint32_t seed, unsigned long long *cycles, uint32_t *sink)
{
	const uint32_t lane = threadIdx.x & 63u;
	v16f acc_h[2][2], acc_t[2][2];
#pragma unroll
	for (int a = 0; a < 2; a++)
#pragma unroll
		for (int b = 0; b < 2; b++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc_h[a][b][r] = acc_t[a][b][r] = 0.f;
	uint32_t pv[4], pa[4], pb[4];
#pragma unroll
	for (int g = 0; g < 4; g++) {
		pv[g] = (seed * 2654435761u + lane * 40503u + g * 97u) | 0x11111111u;
		pa[g] = seed * 40503u + lane * 2654435761u + g * 31u;
		pb[g] = seed * 97u + lane * 31u + g * 40503u;
	}
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (uint32_t it = 0; it < iters; it++) {
		uint32_t vd[4][4], op[4][4];
#pragma unroll
		for (int g = 0; g < 4; g++) {
			pv[g] = __builtin_rotateleft32(pv[g], 1); // ("loaded" words: three an iteration and group, as the kernel's buffer loads bring)
			pa[g] = __builtin_rotateleft32(pa[g], 3);
			pb[g] = __builtin_rotateleft32(pb[g], 5);
			ex_v(pv[g], vd[g]);
		}
#pragma unroll
		for (int a = 0; a < 2; a++)
#pragma unroll
			for (int b = 0; b < 2; b++) acc_h[a][b] = mm(vd[a], vd[2 + b], acc_h[a][b]);
#pragma unroll
		for (int c = 0; c < 3; c++) {
#pragma unroll
			for (int g = 0; g < 4; g++) ex_s(c == 0 ? pa[g] : c == 1 ? pb[g] : (pa[g] ^ pb[g]), vd[g], op[g]);
#pragma unroll
			for (int a = 0; a < 2; a++)
#pragma unroll
				for (int b = 0; b < 2; b++) acc_t[a][b] = mm(op[a], op[2 + b], acc_t[a][b]);
		}
		if (VARIANT == 0) {
#pragma unroll
			for (int i = 0; i < 16; i++) {
				__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
				__builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
			}
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	uint32_t s = 0;
#pragma unroll
	for (int a = 0; a < 2; a++)
#pragma unroll
		for (int b = 0; b < 2; b++)
#pragma unroll
			for (int r = 0; r < 16; r++) s ^= (uint32_t)acc_h[a][b][r] ^ (uint32_t)acc_t[a][b][r];
	if (s == 0x12345u) sink[0] = s;
	if (lane == 0) atomicAdd(cycles, t1 - t0);
}
template <int VARIANT> static int run_step(const char *what, int waves_per_simd, int n_cu, unsigned long long *d_cyc, uint32_t *d_sink)
{
	const uint32_t iters = 4000;
	const int blocks = n_cu, waves = n_cu * 4 * waves_per_simd;
	for (int rep = 0; rep < 2; rep++) {
		HIPOK(hipMemset(d_cyc, 0, 8));
		hipLaunchKernelGGL((step_kernel<VARIANT>), dim3(blocks), dim3(256 * waves_per_simd), 0, 0, iters, 12345u + rep, d_cyc, d_sink);
		HIPOK(hipDeviceSynchronize());
	}
	unsigned long long cyc = 0;
	HIPOK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
	printf("  %-52s %d wavefront(s) per SIMD: %7.1f cycles per iteration\n", what, waves_per_simd, (double)cyc / waves / iters);
	return 0;
}

template <int MODE, int NV> static int run(const char *what, int waves_per_simd, int n_cu, unsigned long long *d_cyc, uint32_t *d_sink)
{
	const uint32_t iters = 4000;
	const int blocks = n_cu, waves = n_cu * 4 * waves_per_simd;
	for (int rep = 0; rep < 2; rep++) {
		HIPOK(hipMemset(d_cyc, 0, 8));
		hipLaunchKernelGGL((mix_kernel<MODE, NV>), dim3(blocks), dim3(256 * waves_per_simd), 0, 0, iters, 12345u + rep, d_cyc, d_sink);
		HIPOK(hipDeviceSynchronize());
	}
	unsigned long long cyc = 0;
	HIPOK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
	printf("  %-52s %d wavefront(s) per SIMD: %7.1f cycles per iteration\n", what, waves_per_simd, (double)cyc / waves / iters);
	return 0;
}

int main()
{
	hipDeviceProp_t prop;
	HIPOK(hipGetDeviceProperties(&prop, 0));
	const int n_cu = prop.multiProcessorCount;
	unsigned long long *d_cyc;
	uint32_t *d_sink;
	HIPOK(hipMalloc(&d_cyc, 8));
	HIPOK(hipMalloc(&d_sink, 4));
	printf("%s, %d CUs; an iteration = 16 FP4 32x32x64 matrix instructions and / or NV vector instructions\n", prop.name, n_cu);
	for (int w = 1; w <= 2; w++) {
		if (run<0, 112>("matrix only", w, n_cu, d_cyc, d_sink)) return 1;
		if (run<1, 112>("vector only, NV = 112", w, n_cu, d_cyc, d_sink)) return 1;
		if (run<2, 112>("both in every wavefront, NV = 112", w, n_cu, d_cyc, d_sink)) return 1;
		if (run<2, 64>("both in every wavefront, NV = 64", w, n_cu, d_cyc, d_sink)) return 1;
		if (run<2, 32>("both in every wavefront, NV = 32", w, n_cu, d_cyc, d_sink)) return 1;
		if (run<1, 64>("vector only, NV = 64", w, n_cu, d_cyc, d_sink)) return 1;
	}
	if (run<3, 112>("matrix-only and vector-only wavefronts side by side", 2, n_cu, d_cyc, d_sink)) return 1;
	for (int w = 1; w <= 2; w++) {
		if (run_step<0>("the pair kernel's step without loads, dealt out 1 : 8", w, n_cu, d_cyc, d_sink)) return 1;
		if (run_step<1>("the same, the compiler's own order", w, n_cu, d_cyc, d_sink)) return 1;
	}
	return 0;
}

// anchor_kernels.hip — phase A on gfx950: speculative chunk chains, bridges and
// the anchor→homology fold.  Replaces the OpenMP loop over queries and
// anchor_homologies (/root/reference/src/process.cxx:433-437, 198-295) and the
// ESA walk under it (src/esa.cxx:361-563).  The per-lane logic lives in
// anchor_core.h (shared with the CPU emulation tests); this file adds what only
// exists on the GPU: persistent lanes with dynamic work fetch, wave-cooperative
// resolution of long suffix comparisons, and the wave-parallel fold.
//
// Roofline: latency / random-access bound (SA, LCP, T and S are gathered), not
// an HBM streaming kernel.  Algorithmic bytes per launch:
// Σ|Q| + 26·|S|.
#include <hip/hip_runtime.h>

#include "anchor_core.h"
#include "kernels.h"

namespace phy {

static __device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

static __device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
	return (uint64_t)__shfl((unsigned long long)v, src, 64);
}

// first differing byte of two 16-byte pieces, 16 if equal
static __device__ __forceinline__ uint32_t first_diff16(const uint4 &a, const uint4 &b)
{
	uint32_t x0 = a.x ^ b.x, x1 = a.y ^ b.y, x2 = a.z ^ b.z, x3 = a.w ^ b.w;
	if (x0) return (uint32_t)(__ffs((int)x0) - 1) >> 3;
	if (x1) return 4u + ((uint32_t)(__ffs((int)x1) - 1) >> 3);
	if (x2) return 8u + ((uint32_t)(__ffs((int)x2) - 1) >> 3);
	if (x3) return 12u + ((uint32_t)(__ffs((int)x3) - 1) >> 3);
	return 16u;
}

static __device__ __forceinline__ uint32_t byte_of(const uint4 &v, uint32_t i)
{
	uint32_t w = (i < 4) ? v.x : (i < 8) ? v.y : (i < 12) ? v.z : v.w;
	return (w >> (8 * (i & 3))) & 0xffu;
}

// All currently active lanes of the wave resolve the leader's comparison
// together: lane r of the m active lanes takes the 16-byte piece r of each
// m*16-byte block.  `s_end` is the first byte past S's zero padding.
static __device__ __forceinline__ void coop_compare(const uint8_t *qp, const uint8_t *sp, uint32_t pos,
													uint32_t maxn, const uint8_t *s_end, uint32_t *out_len,
													uint32_t *out_less)
{
	const uint64_t active = __ballot(1);
	const uint32_t m = (uint32_t)__popcll(active);
	const uint32_t r = (uint32_t)__popcll(active & ((1ull << lane_id()) - 1ull));
	uint32_t base = pos;
	for (;;) {
		uint32_t off = base + r * 16u;
		bool in_q = off < maxn;
		uint32_t d = 0;
		uint32_t qb = 1, sb = 0;
		if (in_q) {
			uint4 a, b;
			__builtin_memcpy(&a, qp + off, 16);
			if (sp + off + 16 <= s_end) {
				__builtin_memcpy(&b, sp + off, 16);
			} else {
				b = make_uint4(0, 0, 0, 0); // past the end of S: the NUL the reference stops at
			}
			d = first_diff16(a, b);
			if (d < 16) {
				qb = byte_of(a, d);
				sb = byte_of(b, d);
			}
		}
		bool hit = !in_q || d < 16;
		uint64_t hm = __ballot(hit);
		if (hm) {
			int first = __ffsll((unsigned long long)hm) - 1;
			uint32_t len = in_q ? off + d : maxn;
			uint32_t less = (in_q && len < maxn) ? (sb < qb ? 1u : 0u) : 0u;
			if (len > maxn) len = maxn;
			*out_len = (uint32_t)__shfl((int)len, first, 64);
			*out_less = (uint32_t)__shfl((int)less, first, 64);
			return;
		}
		base += m * 16u;
	}
}

struct DevAlloc {
	const PhaseA *A;
	__device__ uint32_t operator()() const
	{
		uint32_t b = atomicAdd(A->pool_next, 1u);
		return b < A->pool_blocks ? b : NO_BLOCK;
	}
};

// MODE 0: speculative chunk chains. MODE 1: bridges.
template <int MODE> __global__ __launch_bounds__(256) void chain_kernel(PhaseA A, RefIndex R)
{
	typename std::conditional<MODE == 0, SpecLane, BridgeLane>::type ln;
	const uint8_t *s_end = R.S + R.n + 64;
	bool active = false, done = false, need = false;
	CmpRes res = {0, false};
	CmpReq req = {nullptr, nullptr, 0, 0};
	uint32_t pos = 0;
	DevAlloc alloc = {&A};

	for (;;) {
		if (!need && !done) {
			for (;;) {
				if (!active) {
					uint32_t it = atomicAdd(&A.fetch[MODE], 1u);
					if (it >= A.nchunks) {
						done = true;
						break;
					}
					ln.start(A, A.items[it]);
					active = true;
				}
				if (ln.ch.st == ST_STEP) {
					bool go;
					if constexpr (MODE == 0) go = ln.begin_step(A);
					else go = ln.begin_step(A, R);
					if (!go) {
						active = false;
						continue;
					}
				}
				if (ln.ch.advance(R, res, &req) == ADV_NEED_CMP) {
					need = true;
					pos = req.from;
					break;
				}
				if constexpr (MODE == 0) ln.step_done(A);
				else ln.step_done(A, alloc);
			}
		}
		if (__all(done)) break;
		// short comparisons stay in the lane …
		if (need && cmp_some(req, 64, &pos, &res)) need = false;
		// … long ones are resolved by the whole wave, one at a time
		uint64_t pending = __ballot(need);
		while (pending) {
			int leader = __ffsll((unsigned long long)pending) - 1;
			const uint8_t *qp = (const uint8_t *)shfl64((uint64_t)req.qp, leader);
			const uint8_t *sp = (const uint8_t *)shfl64((uint64_t)req.sp, leader);
			uint32_t p0 = (uint32_t)__shfl((int)pos, leader, 64);
			uint32_t mx = (uint32_t)__shfl((int)req.maxn, leader, 64);
			uint32_t len, less;
			coop_compare(qp, sp, p0, mx, s_end, &len, &less);
			if ((int)lane_id() == leader) {
				res.len = len;
				res.s_less = less != 0;
				need = false;
			}
			pending &= pending - 1;
		}
	}
}

// One wavefront per query: walk the true chain (speculative log, bridge,
// target log from the merge index, …) and fold the accepted anchors into
// homologies (process.cxx:246-292), 64 anchors per step with ballots.
struct FoldCarry {
	uint32_t lq, ls, ll; // last anchor
	uint32_t lr;         // last_was_right_anchor
	uint32_t cs, cq;     // start of the homology being grown
};

static __device__ __forceinline__ void fold_batch(const Anchor *src, uint32_t m, FoldCarry &c, uint32_t border,
												  uint32_t thr, RawHom *out, uint32_t &cnt, uint32_t cap,
												  uint32_t *error)
{
	const uint32_t lane = lane_id();
	Anchor a = {0, 0, 0};
	if (lane < m) a = src[lane];
	int up = (int)lane - 1;
	uint32_t pq = (uint32_t)__shfl_up((int)a.q, 1, 64);
	uint32_t ps = (uint32_t)__shfl_up((int)a.s, 1, 64);
	uint32_t pl = (uint32_t)__shfl_up((int)a.len, 1, 64);
	if (up < 0) {
		pq = c.lq;
		ps = c.ls;
		pl = c.ll;
	}
	Anchor prev = {pq, ps, pl};
	uint32_t right = (lane < m && is_right_anchor(prev, a, border)) ? 1u : 0u;
	uint32_t prev_right = (uint32_t)__shfl_up((int)right, 1, 64);
	if (up < 0) prev_right = c.lr;
	bool start = lane < m && !right;
	bool emit = start && (prev_right || pl / 2 >= thr);
	uint64_t sm = __ballot(start);
	uint64_t below = sm & ((1ull << lane) - 1ull);
	int js = below ? 63 - __clzll((long long)below) : 0;
	uint32_t rs = (uint32_t)__shfl((int)a.s, js, 64);
	uint32_t rq = (uint32_t)__shfl((int)a.q, js, 64);
	if (!below) {
		rs = c.cs;
		rq = c.cq;
	}
	uint64_t em = __ballot(emit);
	if (emit) {
		uint32_t slot = cnt + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
		if (slot < cap) {
			RawHom h = {rs, rq, pq + pl - rq};
			out[slot] = h;
		} else {
			*error = 3;
		}
	}
	cnt += (uint32_t)__popcll(em);
	// carry out
	int lastl = (int)m - 1;
	c.lq = (uint32_t)__shfl((int)a.q, lastl, 64);
	c.ls = (uint32_t)__shfl((int)a.s, lastl, 64);
	c.ll = (uint32_t)__shfl((int)a.len, lastl, 64);
	c.lr = (uint32_t)__shfl((int)right, lastl, 64);
	if (sm) {
		int jl = 63 - __clzll((long long)sm);
		c.cs = (uint32_t)__shfl((int)a.s, jl, 64);
		c.cq = (uint32_t)__shfl((int)a.q, jl, 64);
	}
}

__global__ __launch_bounds__(64) void fold_kernel(PhaseA A, uint32_t nq, uint32_t border, uint32_t thr,
												  RawHom *out, const uint64_t *out_base, const uint32_t *out_cap,
												  uint32_t *out_cnt)
{
	const uint32_t j = blockIdx.x;
	if (j >= nq) return;
	RawHom *dst = out + out_base[j];
	const uint32_t cap = out_cap[j];
	uint32_t cnt = 0;
	FoldCarry c = {0, 0, 0, 0, 0, 0};
	const uint32_t qlen = A.qlen[j];
	if (A.qchunk0[j] < A.qchunk0[j + 1]) {
		uint32_t gc = A.qchunk0[j], idx = 0;
		for (;;) {
			const uint32_t n_spec = A.spec_cnt[gc];
			const Anchor *log = A.spec_anchors + (size_t)gc * A.cap;
			for (uint32_t t = idx; t < n_spec; t += 64) {
				uint32_t m = n_spec - t;
				fold_batch(log + t, m < 64 ? m : 64, c, border, thr, dst, cnt, cap, A.error);
			}
			const BridgeRec *b = &A.bridge[gc];
			const uint32_t bn = b->n, target = b->target, idx_m = b->idx_m;
			if (bn) {
				fold_batch(b->a, bn < BRIDGE_INLINE ? bn : BRIDGE_INLINE, c, border, thr, dst, cnt, cap, A.error);
				uint32_t left = bn > BRIDGE_INLINE ? bn - BRIDGE_INLINE : 0;
				uint32_t blk = b->block;
				while (left && blk != NO_BLOCK) {
					uint32_t m = left < POOL_BLOCK ? left : POOL_BLOCK;
					fold_batch(A.pool[blk].a, m, c, border, thr, dst, cnt, cap, A.error);
					left -= m;
					blk = A.pool[blk].next;
				}
			}
			if (target == BRIDGE_END) break;
			gc = target;
			idx = idx_m;
		}
	}
	// fold_finish (process.cxx:285-292); every lane holds the same carry
	if (lane_id() == 0) {
		uint32_t cs = c.cs, cq = c.cq, clen = c.lq + c.ll - c.cq;
		if (c.ll >= qlen) {
			cs = c.ls;
			cq = 0;
			clen = qlen;
		}
		if (c.lr || c.ll / 2 >= thr) {
			if (cnt < cap) {
				RawHom h = {cs, cq, clen};
				dst[cnt] = h;
			} else {
				*A.error = 3;
			}
			cnt++;
		}
		out_cnt[j] = cnt;
	}
}

// ───────────────────────── launch wrappers ─────────────────────────

void launch_spec(const PhaseA &A, const RefIndex &R, int blocks, hipStream_t st)
{
	hipLaunchKernelGGL(chain_kernel<0>, dim3(blocks), dim3(256), 0, st, A, R);
}
void launch_bridge(const PhaseA &A, const RefIndex &R, int blocks, hipStream_t st)
{
	hipLaunchKernelGGL(chain_kernel<1>, dim3(blocks), dim3(256), 0, st, A, R);
}
void launch_fold(const PhaseA &A, uint32_t nq, uint32_t border, uint32_t thr, RawHom *out,
				 const uint64_t *out_base, const uint32_t *out_cap, uint32_t *out_cnt, hipStream_t st)
{
	hipLaunchKernelGGL(fold_kernel, dim3(nq), dim3(64), 0, st, A, nq, border, thr, out, out_base, out_cap, out_cnt);
}

} // namespace phy

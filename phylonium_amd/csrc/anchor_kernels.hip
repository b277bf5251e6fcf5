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

#include <cstdlib>

#include "anchor_core.h"
#include "kernels.h"

namespace phy {

static __device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

static __device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
	return (uint64_t)__shfl((unsigned long long)v, src, 64);
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
			U4 a = load16(qp + off), b = {0, 0, 0, 0};
			if (sp + off + 16 <= s_end) b = load16(sp + off); // else past the end of S: the NUL the reference stops at
			d = first_diff(a, b);
			if (d < 16) {
				qb = byte_at(a, d);
				sb = byte_at(b, d);
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
// Persistent lanes: every lane runs one chain at a time and fetches the next
// chunk when it finishes.  One loop trip runs the phases of anchor_core.h's
// Chain in their fixed order; a phase is skipped by the wavefront when none of
// its lanes is in it.
template <int MODE> __global__ __launch_bounds__(256) void chain_kernel(PhaseA A, RefIndex R)
{
	typename std::conditional<MODE == 0, SpecLane, BridgeLane>::type ln;
	Chain &ch = ln.ch;
	const uint8_t *s_end = R.S + R.n + 64;
	bool active = false, done = false;
	DevAlloc alloc = {&A};
	ch.fin = false;
	ch.st = ST_STEP;

	for (;;) {
		while (!done) {
			if (active && ch.fin) {
				if constexpr (MODE == 0) ln.step_done(A);
				else ln.step_done(A, alloc);
				ch.fin = false;
			}
			if (!active) {
				uint32_t it = atomicAdd(&A.fetch[MODE], 1u);
				if (it >= A.nchunks) {
					done = true;
					break;
				}
				ln.start(A, A.items[it]);
				active = true;
			}
			if (ch.st == ST_STEP) {
				bool go;
				if constexpr (MODE == 0) go = ln.begin_step(A);
				else go = ln.begin_step(A, R);
				if (!go) {
					active = false;
					continue;
				}
			}
			break;
		}
		if (__all(done)) break;
		const bool live = !done;

		if (live && ch.st == ST_STEP) {
			const uint8_t *a0, *a1 = nullptr;
			uint32_t n = ch.issue_step(R, &a0, &a1);
			U4 qw = load16(a0), sw = {0, 0, 0, 0};
			if (n > 1) sw = load16(a1);
			ch.consume_step(R, qw, sw);
		}
		if (live && ch.st == ST_T) {
			// one 128-byte line: bucket bounds + the records of the (<= 4) candidates
			const uint8_t *a = ch.issue_T(R);
			U4 hdr = load16(a);
			Data d;
			d.w[0] = load16(a + 16);
			d.w[1] = load16(a + 32);
			d.w[2] = load16(a + 48);
			d.w[3] = load16(a + 64);
			ch.consume_T(R, hdr, d);
		}
		if (live && ch.st == ST_CAND) {
			Data d;
			d.w[0] = load16(R.S + ch.c_pos0);
			d.w[1] = d.w[2] = d.w[3] = d.w[0];
			if (ch.c_n > 1) d.w[1] = load16(R.S + ch.c_pos1);
			if (ch.c_n > 2) d.w[2] = load16(R.S + ch.c_pos2);
			if (ch.c_n > 3) d.w[3] = load16(R.S + ch.c_pos3);
			ch.consume_cand(R, d);
		}
		if (live && ch.in_gen()) {
			if (ch.gen_advance(R)) {
				if (ch.consume_probe_sa(load16(ch.issue_probe_sa(R)))) ch.consume_probe_s(load16(ch.issue_probe_s(R)));
				if (ch.in_gen()) ch.gen_advance(R);
			}
		}
		{
			bool want = false;
			if (live && ch.st == ST_EXT) {
				const uint8_t *a[4];
				Data d;
				ch.issue_ext(R, a);
				d.w[0] = load16(a[0]);
				d.w[1] = load16(a[1]);
				d.w[2] = load16(a[2]);
				d.w[3] = load16(a[3]);
				want = ch.consume_ext(R, d);
			}
			// comparisons that ran past EXT_COOP_AT bytes are finished by the whole wave
			uint64_t pending = __ballot(want);
			while (pending) {
				int leader = __ffsll((unsigned long long)pending) - 1;
				const uint8_t *qp = (const uint8_t *)shfl64((uint64_t)(ch.Q + ch.q), leader);
				const uint8_t *sp = (const uint8_t *)shfl64((uint64_t)(R.S + ch.e_p), leader);
				uint32_t p0 = (uint32_t)__shfl((int)ch.e_pos, leader, 64);
				uint32_t mx = (uint32_t)__shfl((int)(ch.qlen - ch.q), leader, 64);
				uint32_t len, less;
				coop_compare(qp, sp, p0, mx, s_end, &len, &less);
				if ((int)lane_id() == leader) ch.deliver(R, len, less);
				pending &= pending - 1;
			}
		}
		if (live && ch.st == ST_FIN) {
			if (ch.fin_needs_lcp(R)) ch.consume_lcp(load16(ch.issue_lcp(R)));
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

static int resident_blocks(const void *fn, int n_cu)
{
	int per_cu = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
	if (const char *e = getenv("PHYLO_CHAIN_BLOCKS_PER_CU")) { // experiments only
		int v = atoi(e);
		if (v >= 1 && v <= per_cu) per_cu = v;
	}
	return per_cu * n_cu;
}

// blocks_cap: never more lanes than chunks
void launch_spec(const PhaseA &A, const RefIndex &R, int n_cu, hipStream_t st)
{
	int blocks = resident_blocks((const void *)chain_kernel<0>, n_cu);
	int need = (int)((A.nchunks + 255) / 256);
	if (need < blocks) blocks = need > 0 ? need : 1;
	hipLaunchKernelGGL(chain_kernel<0>, dim3(blocks), dim3(256), 0, st, A, R);
}
void launch_bridge(const PhaseA &A, const RefIndex &R, int n_cu, hipStream_t st)
{
	int blocks = resident_blocks((const void *)chain_kernel<1>, n_cu);
	int need = (int)((A.nchunks + 255) / 256);
	if (need < blocks) blocks = need > 0 ? need : 1;
	hipLaunchKernelGGL(chain_kernel<1>, dim3(blocks), dim3(256), 0, st, A, R);
}
void launch_fold(const PhaseA &A, uint32_t nq, uint32_t border, uint32_t thr, RawHom *out,
				 const uint64_t *out_base, const uint32_t *out_cap, uint32_t *out_cnt, hipStream_t st)
{
	hipLaunchKernelGGL(fold_kernel, dim3(nq), dim3(64), 0, st, A, nq, border, thr, out, out_base, out_cap, out_cnt);
}

} // namespace phy

// anchor_kernels.hip — phase A on gfx950, the part behind the chains: the fold of the accepted
// anchors into homologies.  Replaces the homology bookkeeping of anchor_homologies
// (/root/reference/src/process.cxx:246-292); the chains themselves (speculative chunks, bridges)
// are lean_kernels.hip.  The per-anchor logic lives in anchor_core.h (shared with the CPU
// emulation tests); this file adds the block-parallel walk over a query's chunks and bridges.
//
// Roofline: a per-query latency chain (LDS scans and shuffles), not an HBM streaming kernel.
#include <hip/hip_runtime.h>

#include "anchor_core.h"
#include "kernels.h"

namespace phy {

static __device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

static __device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
	return (uint64_t)__shfl((unsigned long long)v, src, 64);
}

// A pointer that went through LDS or a lane shuffle comes back generic, and a generic
// load is a FLAT instruction that also counts against LDS waits.  These restate that the
// address is in global memory.
typedef const uint8_t __attribute__((address_space(1))) *global_bytes;
static __device__ __forceinline__ U4 gload16(const uint8_t *p)
{
	U4 v;
	__builtin_memcpy(&v, (global_bytes)(uintptr_t)p, 16);
	return v;
}
static __device__ __forceinline__ Anchor gload_anchor(const Anchor *p)
{
	Anchor v;
	__builtin_memcpy(&v, (global_bytes)(uintptr_t)p, sizeof(Anchor));
	return v;
}

// ───────────────────────── fold: anchors → homologies ─────────────────────────
// One block of FOLD_THREADS threads per query.
//  (1) The chunk metadata of a window of FOLD_WCH chunks (bridge target, merge
//      index, bridge size, log size) is copied into LDS with coalesced loads.
//  (2) The true chain inside the window — chunk 0's log, its bridge, the target
//      chunk's log from the merge index, … — is found by pointer doubling over the
//      bridge targets (every chunk has one outgoing link), and block-wide prefix
//      sums give every live chunk its anchor segments (address, count, offset).
//      No global pointer chasing except through bridge overflow blocks (rare).
//  (3) All threads fold the window's anchors, FOLD_ITER per iteration, exactly as
//      process.cxx:246-292 does one at a time: the right-anchor test needs only
//      the previous anchor (neighbour lane / LDS across waves), a homology ends at
//      every non-right anchor, its start is the latest non-right anchor before it
//      (ballot + LDS), and the output position is a prefix count, so emission
//      stays in query order (a std::sort over lists with equal starts depends on it).

static const uint32_t FOLD_WCH = 1024;  // chunks of metadata per window
static const uint32_t FOLD_SEGS = 5120; // anchor segments per window (a chunk contributes 1-3; 5 per chunk on average would overflow, reported as error 4)
static const uint32_t FOLD_APT = 4; // anchors per thread and iteration
static const uint32_t FOLD_THREADS = 1024; // per block (= per query)
static const uint32_t FOLD_WAVES = FOLD_THREADS / 64;
static const uint32_t FOLD_PPT = FOLD_WCH / FOLD_THREADS; // chain positions per thread when segments are laid out
static const uint32_t FOLD_ITER = FOLD_THREADS * FOLD_APT;

// Workgroup barrier that orders LDS traffic only: global loads issued before it may
// still be in flight afterwards (__syncthreads would wait for them).
__device__ __forceinline__ void lds_barrier()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

static const uint32_t FOLD_STAGE = 1024;
struct FoldShared {
	uint32_t tgt[FOLD_WCH], idxm[FOLD_WCH], bn[FOLD_WCH], blk[FOLD_WCH], scnt[FOLD_WCH];
	const Anchor *seg_ptr[FOLD_SEGS];
	uint32_t seg_off[FOLD_SEGS + 1];
	uint32_t nseg, next_gc, next_idx, finished;
	// chain walk by pointer doubling: node FOLD_WCH.. = "left the window"
	uint16_t jump[11][FOLD_WCH + 1]; // jump[k][t] = node reached from t after 2^k links
	uint16_t dist[FOLD_WCH + 1];     // links from t until the window is left
	uint16_t path[FOLD_WCH];         // live chunks in chain order
	uint8_t live[FOLD_WCH + 1];
	uint32_t scan_s[FOLD_WAVES], scan_a[FOLD_WAVES];
	// per-iteration exchange between the four waves
	// (two sets, taken in turn by the iterations: the next iteration writes the other set, so no barrier has to keep it from
	// overtaking the slowest wavefront's reads of this one — three barriers an iteration instead of four)
	uint32_t wl_q[2][FOLD_WAVES], wl_s[2][FOLD_WAVES], wl_len[2][FOLD_WAVES], wl_r[2][FOLD_WAVES]; // last anchor of each wave and its right flag
	uint32_t st_has[2][FOLD_WAVES], st_s[2][FOLD_WAVES], st_q[2][FOLD_WAVES]; // latest non-right anchor of each wave
	uint32_t ecnt[2][FOLD_WAVES];
	// several blocks per query: a block's homologies wait here until FOLD_STAGE of them go out behind one atomic
	RawHom stage[FOLD_STAGE];
	uint32_t stage_base;
};

// Several blocks per query (nb of them; blockIdx = query * nb + part): every block walks the windows itself — the
// walk is a latency chain, the same for all — and folds a contiguous part of each window's iterations.  What a part
// needs from the anchors before it (the last anchor, whether that was a right anchor, the latest non-right anchor =
// the start of the run that is open) is data, not state: it is read back from the logs (carry_at).  Homologies leave
// through slots handed out by an atomic counter per query: with one block per query they come out in query order as
// they always did; with several, a query's list is in the order its parts got their slots — the sort + chain filter
// behind the fold orders by projected start anyway, and the host path re-orders by query position first (the
// reference's std::sort sees the homologies in query order, process.cxx:438).
__global__ __launch_bounds__(FOLD_THREADS) void fold_kernel(PhaseA A, uint32_t j0, uint32_t nq, uint32_t nb, uint32_t border, uint32_t thr,
												   RawHom *out, const uint64_t *out_base, const uint32_t *out_cap,
												   uint32_t *out_cnt)
{
	__shared__ FoldShared sh;
	const uint32_t j = j0 + blockIdx.x / nb, part = blockIdx.x % nb;
	if (j >= nq) return;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	RawHom *dst = out + out_base[j];
	const uint32_t cap = out_cap[j];
	const uint32_t qlen = A.qlen[j];
	const uint32_t c_begin = A.qchunk0[j], c_end = A.qchunk0[j + 1];
	// carry (identical in every thread): the last anchor, whether it was a right anchor, the latest non-right anchor
	uint32_t lq = 0, ls = 0, ll = 0, lr = 0, cs = 0, cq = 0, cnt = 0; // (cnt: homologies so far, one block per query only)
	uint32_t gc = c_begin, idx = 0;
	bool more = c_begin < c_end;
	uint32_t stg_n = 0; // homologies in the block's stage (the same in every thread)
	auto flush_stage = [&]() {
		__syncthreads();
		if (stg_n) {
			if (tid == 0) sh.stage_base = atomicAdd(&out_cnt[j], stg_n);
			__syncthreads();
			const uint32_t b = sh.stage_base;
			for (uint32_t t = tid; t < stg_n; t += FOLD_THREADS) {
				if (b + t < cap) dst[b + t] = sh.stage[t];
				else *A.error = 3;
			}
			__syncthreads();
		}
		stg_n = 0;
	};
#ifdef PHY_FOLD_TIMING
	unsigned long long ft[6] = {0, 0, 0, 0, 0, 0}, ft_prev = __builtin_amdgcn_s_memrealtime();
#define FOLD_TICK(i)                                                    \
	{                                                                   \
		const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
		ft[i] += now_ - ft_prev;                                        \
		ft_prev = now_;                                                 \
	}
#else
#define FOLD_TICK(i)
#endif

	while (more) {
		// (1) metadata window [gc, gc + FOLD_WCH)
		const uint32_t wbeg = gc;
		const uint32_t wn = (c_end - wbeg < FOLD_WCH) ? c_end - wbeg : FOLD_WCH;
		for (uint32_t t = tid; t < wn; t += FOLD_THREADS) {
			const BridgeRec *b = &A.bridge[wbeg + t];
			sh.tgt[t] = b->target;
			sh.idxm[t] = b->idx_m;
			sh.bn[t] = b->n;
			sh.blk[t] = b->block;
			sh.scnt[t] = A.spec_cnt[wbeg + t];
		}
		__syncthreads();
		FOLD_TICK(0)
		// (2) walk.  Every chunk has one outgoing link (its bridge's merge target), so the
		// true chain inside the window is found without following it link by link:
		//   jump[k][t]  by doubling;  dist[t] = number of links until the window is left;
		//   a chunk is live iff it is reached from the entry chunk — marked level by
		//   level (live ∪= jump[k](live), k = 10..0);  its position in the chain is
		//   dist[entry] - dist[t].
		// Then block-wide prefix sums over the chain give every live chunk its segment
		// slots and anchor offsets.
		const uint32_t SINK = wn; // any node >= wn means "outside the window"
		bool plain_mine = true; // every chunk's link goes to the next chunk (or out of the window from the last one)
		for (uint32_t t = tid; t <= wn; t += FOLD_THREADS) {
			uint32_t nx = SINK;
			if (t < wn) {
				uint32_t tg = sh.tgt[t];
				if (tg != BRIDGE_END && tg - wbeg < wn) nx = tg - wbeg;
				plain_mine = plain_mine && nx == t + 1u;
			}
			sh.jump[0][t] = (uint16_t)nx;
			sh.dist[t] = t < wn ? 1 : 0;
			sh.live[t] = 0;
		}
		// Nearly every window is like that — a bridge that ends beyond the next chunk has walked a whole chunk without
		// meeting the speculative chain — and then the chain is the window itself: no doubling, no marking (30 barriers).
		const bool plain = __syncthreads_and(plain_mine ? 1 : 0) != 0;
		if (plain) {
			for (uint32_t t = tid; t < wn; t += FOLD_THREADS) sh.path[t] = (uint16_t)t;
			if (tid == 0) sh.dist[0] = (uint16_t)wn;
			__syncthreads();
		} else {
		// (neither loop over the levels unrolled: eleven levels' LDS addresses held in registers took the kernel past its 128)
#pragma unroll 1
		for (uint32_t k = 0; k < 10; k++) {
			uint32_t nd[FOLD_WCH / FOLD_THREADS + 1], nj[FOLD_WCH / FOLD_THREADS + 1], c = 0;
			for (uint32_t t = tid; t <= wn; t += FOLD_THREADS, c++) {
				uint32_t jt = sh.jump[k][t];
				nd[c] = (uint32_t)sh.dist[t] + (uint32_t)sh.dist[jt];
				nj[c] = sh.jump[k][jt];
			}
			__syncthreads();
			c = 0;
			for (uint32_t t = tid; t <= wn; t += FOLD_THREADS, c++) {
				sh.dist[t] = (uint16_t)nd[c];
				sh.jump[k + 1][t] = (uint16_t)nj[c];
			}
			__syncthreads();
		}
		const uint32_t entry = gc - wbeg;
		if (tid == 0) sh.live[entry] = 1;
		__syncthreads();
#pragma unroll 1
		for (int k = 10; k >= 0; k--) {
			for (uint32_t t = tid; t < wn; t += FOLD_THREADS)
				if (sh.live[t]) {
					uint32_t jt = sh.jump[k][t];
					if (jt < wn) sh.live[jt] = 1;
				}
			__syncthreads();
		}
		const uint32_t npath_walked = sh.dist[entry]; // live chunks in this window
		for (uint32_t t = tid; t < wn; t += FOLD_THREADS)
			if (sh.live[t]) sh.path[npath_walked - sh.dist[t]] = (uint16_t)t;
		__syncthreads();
		}
		const uint32_t npath = sh.dist[gc - wbeg];
		FOLD_TICK(1)
		// segments: every thread takes FOLD_PPT consecutive chain positions
		{
			uint32_t my_seg[FOLD_PPT], my_an[FOLD_PPT], my_spec[FOLD_PPT], my_idx[FOLD_PPT];
			uint32_t ssum = 0, asum = 0;
#pragma unroll
			for (uint32_t e = 0; e < FOLD_PPT; e++) {
				const uint32_t pz = tid * FOLD_PPT + e;
				my_seg[e] = my_an[e] = my_spec[e] = my_idx[e] = 0;
				if (pz < npath) {
					const uint32_t t = sh.path[pz];
					const uint32_t idx_in = pz == 0 ? idx : sh.idxm[sh.path[pz - 1]];
					const uint32_t sc = sh.scnt[t], b_n = sh.bn[t];
					const uint32_t spec_n = sc > idx_in ? sc - idx_in : 0u;
					const uint32_t nblk = b_n > BRIDGE_INLINE ? (b_n - BRIDGE_INLINE + POOL_BLOCK - 1) / POOL_BLOCK : 0u;
					my_idx[e] = idx_in;
					my_spec[e] = spec_n;
					my_seg[e] = (spec_n ? 1u : 0u) + (b_n ? 1u : 0u) + nblk;
					my_an[e] = spec_n + b_n;
				}
				ssum += my_seg[e];
				asum += my_an[e];
			}
			// exclusive block scan of (ssum, asum)
			uint32_t ps = ssum, pa = asum;
#pragma unroll
			for (int dd = 1; dd < 64; dd <<= 1) {
				uint32_t a1 = (uint32_t)__shfl_up((int)ps, dd, 64), a2 = (uint32_t)__shfl_up((int)pa, dd, 64);
				if ((int)lane >= dd) {
					ps += a1;
					pa += a2;
				}
			}
			if (lane == 63) {
				sh.scan_s[wave] = ps;
				sh.scan_a[wave] = pa;
			}
			__syncthreads();
			uint32_t bs = 0, ba = 0;
			for (uint32_t w2 = 0; w2 < wave; w2++) {
				bs += sh.scan_s[w2];
				ba += sh.scan_a[w2];
			}
			uint32_t tot_s = 0, tot_a = 0;
			for (uint32_t w2 = 0; w2 < FOLD_WAVES; w2++) {
				tot_s += sh.scan_s[w2];
				tot_a += sh.scan_a[w2];
			}
			uint32_t slot = bs + ps - ssum, off = ba + pa - asum;
			if (tot_s > FOLD_SEGS) {
				if (tid == 0) *A.error = 4; // more anchor segments than a window's list holds
			} else {
#pragma unroll
				for (uint32_t e = 0; e < FOLD_PPT; e++) {
					const uint32_t pz = tid * FOLD_PPT + e;
					if (pz >= npath) break;
					const uint32_t t = sh.path[pz], g = wbeg + t;
					if (my_spec[e]) {
						sh.seg_ptr[slot] = A.spec_anchors + (size_t)chunk_geom(A, j, g - c_begin).log0 + my_idx[e];
						sh.seg_off[slot++] = off;
						off += my_spec[e];
					}
					uint32_t b_n = sh.bn[t];
					if (b_n) {
						uint32_t m = b_n < BRIDGE_INLINE ? b_n : BRIDGE_INLINE;
						sh.seg_ptr[slot] = A.bridge[g].a;
						sh.seg_off[slot++] = off;
						off += m;
						uint32_t left = b_n - m, bk = sh.blk[t];
						while (left && bk != NO_BLOCK) {
							uint32_t mm = left < POOL_BLOCK ? left : POOL_BLOCK;
							sh.seg_ptr[slot] = A.pool[bk].a;
							sh.seg_off[slot++] = off;
							off += mm;
							left -= mm;
							bk = A.pool[bk].next;
						}
					}
				}
			}
			if (tid == 0) {
				const uint32_t last = sh.path[npath - 1];
				const uint32_t tg = sh.tgt[last];
				sh.nseg = tot_s > FOLD_SEGS ? 0 : tot_s;
				sh.seg_off[sh.nseg] = tot_s > FOLD_SEGS ? 0 : tot_a;
				sh.finished = (tg == BRIDGE_END || tot_s > FOLD_SEGS) ? 1u : 0u;
				sh.next_gc = tg;
				sh.next_idx = sh.idxm[last];
			}
		}
		__syncthreads();
		const uint32_t nseg = sh.nseg, total = sh.seg_off[nseg];
		FOLD_TICK(2)
		// (3) block-parallel fold over the window's `total` anchors, FOLD_ITER per iteration
		// (FOLD_APT consecutive anchors per thread: the shuffles, scans and barriers of an
		// iteration are a fixed latency chain, so it pays to put many anchors behind each).
		// The next iteration's anchors are requested before this one's are folded: the
		// barriers below wait for LDS only (lds_barrier), so the loads stay in flight.
		// anchor k of the window (k < total), wherever its segment is
		auto anchor_at = [&](uint32_t k) {
			uint32_t lo = 0, hi = nseg;
			while (hi - lo > 1) {
				const uint32_t mid = (lo + hi) >> 1;
				if (sh.seg_off[mid] <= k) lo = mid;
				else hi = mid;
			}
			return gload_anchor(sh.seg_ptr[lo] + (k - sh.seg_off[lo]));
		};
		// The carry in front of anchor a0 of this window, from the data: (lq, ls, ll) = anchor a0 - 1, lr = was it a right
		// anchor, (cs, cq) = the latest non-right anchor before a0 — looked for backwards, FOLD_THREADS anchors per round;
		// what lies before the window is the carry the window started with (wq .. wcq).  All threads take part and get
		// the same values.
		const uint32_t wq = lq, ws = ls, wl = ll, wr = lr, wcs = cs, wcq = cq;
		auto carry_at = [&](uint32_t a0) {
			if (a0 == 0) {
				lq = wq, ls = ws, ll = wl, lr = wr, cs = wcs, cq = wcq;
				return;
			}
			const Anchor window_last = {wq, ws, wl};
			const Anchor a1 = anchor_at(a0 - 1), p1 = a0 >= 2 ? anchor_at(a0 - 2) : window_last;
			lq = a1.q, ls = a1.s, ll = a1.len;
			lr = is_right_anchor(p1, a1, border) ? 1u : 0u;
			cs = wcs, cq = wcq;
			for (uint32_t top = a0; top > 0;) { // anchors [top - FOLD_THREADS, top), thread t looks at top - 1 - t
				const bool in = tid < top;
				const uint32_t k = in ? top - 1u - tid : 0u;
				bool nonright = false;
				Anchor ak = {0, 0, 0};
				if (in) {
					ak = anchor_at(k);
					const Anchor pk = k ? anchor_at(k - 1) : window_last;
					nonright = !is_right_anchor(pk, ak, border);
				}
				const uint64_t m = __ballot(nonright);
				if (lane == 0) sh.scan_s[wave] = m ? (uint32_t)(__ffsll((unsigned long long)m) - 1) : 64u;
				__syncthreads();
				uint32_t fw = FOLD_WAVES, fl = 0; // the first wave (highest anchors) that found one, and its lane
				for (uint32_t w2 = 0; w2 < FOLD_WAVES; w2++)
					if (fw == FOLD_WAVES && sh.scan_s[w2] < 64u) {
						fw = w2;
						fl = sh.scan_s[w2];
					}
				if (fw < FOLD_WAVES && wave == fw && lane == fl) {
					sh.st_s[0][0] = ak.s;
					sh.st_q[0][0] = ak.q;
				}
				__syncthreads();
				if (fw < FOLD_WAVES) {
					cs = sh.st_s[0][0];
					cq = sh.st_q[0][0];
					__syncthreads();
					break;
				}
				top = top > FOLD_THREADS ? top - FOLD_THREADS : 0u;
			}
		};
		// this block's part of the window's iterations
		const uint32_t n_it = (total + FOLD_ITER - 1) / FOLD_ITER;
		const uint32_t it0 = (uint32_t)((uint64_t)n_it * part / nb), it1 = (uint32_t)((uint64_t)n_it * (part + 1) / nb);
		const uint32_t base0 = it0 * FOLD_ITER, base1 = it1 * FOLD_ITER < total ? it1 * FOLD_ITER : total;
		if (nb > 1 && it0 < it1) carry_at(base0);
		FOLD_TICK(3)
		uint32_t cur = 0; // segment cursor of this thread (anchor indices only grow)
		Anchor an[FOLD_APT];
		uint32_t evn = 0;
		auto fetch = [&](uint32_t base) {
			const uint32_t k0 = base + tid * FOLD_APT;
			evn = 0;
#pragma unroll
			for (uint32_t e = 0; e < FOLD_APT; e++) an[e].q = an[e].s = an[e].len = 0;
			if (k0 < total) {
				// the segment holding anchor k0: last s in [cur, nseg) with seg_off[s] <= k0
				uint32_t lo = cur, hi = nseg;
				while (hi - lo > 1) {
					const uint32_t mid = (lo + hi) >> 1;
					if (sh.seg_off[mid] <= k0)
						lo = mid;
					else
						hi = mid;
				}
				cur = lo;
				uint32_t s_beg = sh.seg_off[cur], s_end = sh.seg_off[cur + 1];
				const Anchor *s_ptr = sh.seg_ptr[cur];
#pragma unroll
				for (uint32_t e = 0; e < FOLD_APT; e++)
					if (k0 + e < total) {
						if (k0 + e >= s_end) {
							do s_end = sh.seg_off[++cur + 1];
							while (k0 + e >= s_end);
							s_beg = sh.seg_off[cur];
							s_ptr = sh.seg_ptr[cur];
						}
						an[e] = gload_anchor(s_ptr + (k0 + e - s_beg));
						evn = e + 1;
					}
			}
		};
		if (base0 < base1) fetch(base0);
		for (uint32_t base = base0; base < base1; base += FOLD_ITER) {
			const uint32_t k0 = base + tid * FOLD_APT;
			const uint32_t pb = ((base - base0) / FOLD_ITER) & 1u; // this iteration's set of the exchange arrays
			Anchor a[FOLD_APT];
#pragma unroll
			for (uint32_t e = 0; e < FOLD_APT; e++) a[e] = an[e];
			const uint32_t ev = evn; // valid anchors of this thread (the invalid ones are at the very end)
			// This iteration's anchors were asked for an iteration ago: they are waited for HERE, before the next ones are
			// asked for.  The compiler cannot count the loads fetch() issues under its conditions, and where the iteration
			// first looks at its own anchors it waited for everything in flight (s_waitcnt vmcnt(0)) — the anchors just asked
			// for with them, their whole latency in every iteration: the fetch ahead hid nothing.
			__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
			if (base + FOLD_ITER < base1) fetch(base + FOLD_ITER);
			const uint32_t m = total - base < FOLD_ITER ? total - base : FOLD_ITER; // valid anchors this iteration
			Anchor tl = a[0]; // this thread's last valid anchor
#pragma unroll
			for (uint32_t e = 1; e < FOLD_APT; e++)
				if (e < ev) tl = a[e];
			const bool wave_last = ev > 0 && (lane == 63 || k0 + ev == total);
			if (wave_last) {
				sh.wl_q[pb][wave] = tl.q;
				sh.wl_s[pb][wave] = tl.s;
				sh.wl_len[pb][wave] = tl.len;
			}
			lds_barrier();
			// (Measured and left: every thread fetching its first anchor's predecessor itself — one more 16-byte load beside
			// its four — makes this barrier and the shuffle below unnecessary; the iteration then has two barriers and is
			// slower all the same: C3 0.159 -> 0.168 ms, C4 0.475 -> 0.500, C5 0.875 -> 0.905.)
			Anchor prev0;
			prev0.q = (uint32_t)__shfl_up((int)tl.q, 1, 64);
			prev0.s = (uint32_t)__shfl_up((int)tl.s, 1, 64);
			prev0.len = (uint32_t)__shfl_up((int)tl.len, 1, 64);
			if (lane == 0) {
				if (wave == 0) {
					prev0.q = lq;
					prev0.s = ls;
					prev0.len = ll;
				} else {
					prev0.q = sh.wl_q[pb][wave - 1];
					prev0.s = sh.wl_s[pb][wave - 1];
					prev0.len = sh.wl_len[pb][wave - 1];
				}
			}
			// r: bit e = anchor e is a right anchor of its predecessor
			uint32_t r = (ev > 0 && is_right_anchor(prev0, a[0], border)) ? 1u : 0u;
#pragma unroll
			for (uint32_t e = 1; e < FOLD_APT; e++)
				if (e < ev && is_right_anchor(a[e - 1], a[e], border)) r |= 1u << e;
			const uint32_t rl = ev ? (r >> (ev - 1)) & 1u : 0u;
			// this thread's latest non-right anchor
			bool has_t = false;
			uint32_t ts = 0, tq = 0;
#pragma unroll
			for (uint32_t e = 0; e < FOLD_APT; e++)
				if (e < ev && !((r >> e) & 1u)) {
					has_t = true;
					ts = a[e].s;
					tq = a[e].q;
				}
			const uint64_t sm = __ballot(has_t);
			if (wave_last) sh.wl_r[pb][wave] = rl;
			if (lane == 0) sh.st_has[pb][wave] = sm ? 1u : 0u;
			if (sm && (int)lane == 63 - __clzll((long long)sm)) {
				sh.st_s[pb][wave] = ts;
				sh.st_q[pb][wave] = tq;
			}
			lds_barrier();
			uint32_t prev_right0 = (uint32_t)__shfl_up((int)rl, 1, 64);
			if (lane == 0) prev_right0 = wave == 0 ? lr : sh.wl_r[pb][wave - 1];
			// a homology ends at every non-right anchor; it is emitted iff the anchor before it
			// was a right anchor or long enough (process.cxx:261)
			uint32_t em = (ev > 0 && !(r & 1u) && (prev_right0 || prev0.len / 2 >= thr)) ? 1u : 0u;
#pragma unroll
			for (uint32_t e = 1; e < FOLD_APT; e++)
				if (e < ev && !((r >> e) & 1u) && (((r >> (e - 1)) & 1u) || a[e - 1].len / 2 >= thr)) em |= 1u << e;
			const uint32_t ne = (uint32_t)__popc(em);
			uint32_t pe = ne; // inclusive wave scan of the emit counts
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				uint32_t t1 = (uint32_t)__shfl_up((int)pe, d, 64);
				if ((int)lane >= d) pe += t1;
			}
			if (lane == 63) sh.ecnt[pb][wave] = pe;
			// the run that is open when this thread's first anchor arrives started at the latest
			// non-right anchor of an earlier thread / wave / iteration
			const uint64_t below = sm & ((1ull << lane) - 1ull);
			const int js = below ? 63 - __clzll((long long)below) : 0;
			uint32_t rs = (uint32_t)__shfl((int)ts, js, 64);
			uint32_t rq = (uint32_t)__shfl((int)tq, js, 64);
			if (!below) {
				rs = cs;
				rq = cq;
				for (int w2 = (int)wave - 1; w2 >= 0; w2--)
					if (sh.st_has[pb][w2]) {
						rs = sh.st_s[pb][w2];
						rq = sh.st_q[pb][w2];
						break;
					}
			}
			lds_barrier();
			// slots of the query's list: one block per query counts them itself (query order); several blocks take them
			// from the query's counter, a wavefront at a time
			// With several blocks the slots come from the query's counter.  An atomic per wavefront and iteration (round 3's
			// first version) put its round trip — and, the counters being in-order, the wait for the next iteration's
			// anchors already requested — into every iteration: 21 us instead of 10.  The block's homologies are staged in
			// LDS instead and leave FOLD_STAGE at a time behind one atomic.
			uint32_t slot = pe - ne, tot = 0;
			for (uint32_t w2 = 0; w2 < wave; w2++) slot += sh.ecnt[pb][w2];
			for (uint32_t w2 = 0; w2 < FOLD_WAVES; w2++) tot += sh.ecnt[pb][w2];
			bool staged = false;
			if (nb == 1) {
				slot += cnt;
				cnt += tot;
			} else {
				if (stg_n + tot > FOLD_STAGE) flush_stage();
				if (tot > FOLD_STAGE) { // (more than the stage holds in one iteration: straight out)
					if (tid == 0) sh.stage_base = atomicAdd(&out_cnt[j], tot);
					__syncthreads();
					slot += sh.stage_base;
				} else {
					slot += stg_n;
					stg_n += tot;
					staged = true;
				}
			}
#pragma unroll
			for (uint32_t e = 0; e < FOLD_APT; e++) {
				if ((em >> e) & 1u) {
					const Anchor &pv = e == 0 ? prev0 : a[e ? e - 1 : 0];
					const RawHom h = {rs, rq, pv.q + pv.len - rq};
					if (staged) {
						sh.stage[slot] = h;
					} else if (slot < cap) {
						dst[slot] = h;
					} else {
						*A.error = 3;
					}
					slot++;
				}
				if (e < ev && !((r >> e) & 1u)) {
					rs = a[e].s;
					rq = a[e].q;
				}
			}
			// carry out (same values in every thread)
			const uint32_t lw = ((m - 1) / FOLD_APT) >> 6;
			lq = sh.wl_q[pb][lw];
			ls = sh.wl_s[pb][lw];
			ll = sh.wl_len[pb][lw];
			lr = sh.wl_r[pb][lw];
			for (int w2 = (int)FOLD_WAVES - 1; w2 >= 0; w2--)
				if (sh.st_has[pb][w2]) {
					cs = sh.st_s[pb][w2];
					cq = sh.st_q[pb][w2];
					break;
				}
			// (no barrier here: the next iteration writes the other set of the exchange arrays, and the set after that is
			// three barriers away)
		}
		// the carry behind the window's last anchor: every block needs it for the next window (and block 0 for the
		// query's last homology); the block that folded the window's end has it already
		FOLD_TICK(4)
		if (nb > 1 && !(it0 < it1 && base1 == total)) carry_at(total);
		FOLD_TICK(5)
		more = !sh.finished;
		gc = sh.next_gc;
		idx = sh.next_idx;
		__syncthreads();
	}
	if (nb > 1) flush_stage();
#ifdef PHY_FOLD_TIMING
	if (tid == 0 && blockIdx.x / nb == 1)
		printf("fold timing, query %u part %u of %u (us): metadata %.1f  walk %.1f  segments %.1f  carry in %.1f  fold %.1f  carry out %.1f\n", j, part,
			   nb, ft[0] / 100.0, ft[1] / 100.0, ft[2] / 100.0, ft[3] / 100.0, ft[4] / 100.0, ft[5] / 100.0);
#endif
	// fold_finish (process.cxx:285-292)
	if (tid == 0 && part == 0) {
		uint32_t fs = cs, fq = cq, flen = lq + ll - cq;
		if (ll >= qlen) {
			fs = ls;
			fq = 0;
			flen = qlen;
		}
		if (nb == 1) out_cnt[j] = cnt;
		if (lr || ll / 2 >= thr) {
			const uint32_t slot = atomicAdd(&out_cnt[j], 1u);
			if (slot < cap) {
				RawHom h = {fs, fq, flen};
				dst[slot] = h;
			} else {
				*A.error = 3;
			}
		}
	}
}

// ───────────────────────── launch wrappers ─────────────────────────

// queries [j0, j1)
void launch_fold(const PhaseA &A, uint32_t j0, uint32_t j1, uint32_t border, uint32_t thr, RawHom *out,
				 const uint64_t *out_base, const uint32_t *out_cap, uint32_t *out_cnt, hipStream_t st, uint32_t blocks_per_query,
				 bool zero_cnt)
{
	if (j1 <= j0) return;
	const uint32_t nb = blocks_per_query ? blocks_per_query : 1u;
	if (zero_cnt) (void)hipMemsetAsync(out_cnt + j0, 0, (size_t)(j1 - j0) * 4, st); // the list lengths are counted up by the blocks
	hipLaunchKernelGGL(fold_kernel, dim3((j1 - j0) * nb), dim3(FOLD_THREADS), 0, st, A, j0, j1, nb, border, thr, out, out_base, out_cap, out_cnt);
}

} // namespace phy

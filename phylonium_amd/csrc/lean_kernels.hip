// lean_kernels.hip — phase A's chain kernels on 2-bit packed operands (lean_core.h),
// plus the kernels that make the packed tables.
//
// The speculative chunk chains, their overruns and the bridges; their logs, exits and bridge
// records are what fold_kernel (anchor_kernels.hip) reads.  Reference: anchor_homologies + the ESA match under it, /root/reference/src/process.cxx:198-295, src/esa.cxx:361-563.
//
// One loop trip = one batch of loads for all 64 lanes whatever phase each is in
// (32 B at pA, 32 B at pB, 8 B at pY), then the per-phase digest.  Steps the packed
// path cannot answer are resolved by the whole wavefront from the raw bytes
// (coop_resolve): 64 suffixes compared per round, so a bucket of any size takes
// log64 rounds, and every special case ('!' in the window, the query's end, repeats)
// is just byte comparison there.
//
// Roofline: one random 16-byte slot per step out of the k-mer table (a memory request each: the
// chip serves ~55 G random rows a second whatever their size, profiles/r04_gather_bench.jsonl) +
// instruction issue; algorithmic bytes per launch: Σ|Q| + 26·|S|.
#include <hip/hip_runtime.h>

#include <atomic>

#include "kernels.h"
#include "lean_core.h"

namespace phy {

typedef const uint8_t __attribute__((address_space(1))) *lean_global_bytes;
static __device__ __forceinline__ U4 lg16(const uint8_t *p)
{
	U4 v;
	__builtin_memcpy(&v, (lean_global_bytes)(uintptr_t)p, 16);
	return v;
}
struct U2 {
	uint32_t x, y;
};
static __device__ __forceinline__ U2 lg8(const uint8_t *p)
{
	U2 v;
	__builtin_memcpy(&v, (lean_global_bytes)(uintptr_t)p, 8);
	return v;
}
static __device__ __forceinline__ uint32_t lane64() { return threadIdx.x & 63u; }
static __device__ __forceinline__ uint32_t bcast(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }

// ───────────────────────── wave-cooperative comparisons (raw bytes) ─────────────────────────

// All 64 lanes compare qp[pos ..) with sp[pos ..): lane r takes the 16-byte piece r of every
// 1 KiB block.  maxn = bytes of the query from qp; s_end = first byte past S's zero padding.
static __device__ void wave_compare(const uint8_t *qp, const uint8_t *sp, uint32_t pos, uint32_t maxn,
									const uint8_t *s_end, uint32_t *out_len, uint32_t *out_less)
{
	const uint32_t r = lane64();
	uint32_t base = pos;
	for (;;) {
		const uint32_t off = base + r * 16u;
		const bool in_q = off < maxn;
		uint32_t d = 0, qb = 1, sb = 0;
		if (in_q) {
			const U4 a = lg16(qp + off);
			U4 b = {0, 0, 0, 0};
			if (sp + off + 16 <= s_end) b = lg16(sp + off); // past the end of S: the NUL the reference stops at
			d = first_diff(a, b);
			if (d < 16) {
				qb = byte_at(a, d);
				sb = byte_at(b, d);
			}
		}
		const bool hit = !in_q || d < 16;
		const uint64_t hm = __ballot(hit);
		if (hm) {
			const int first = __ffsll((unsigned long long)hm) - 1;
			uint32_t len = in_q ? off + d : maxn;
			const uint32_t less = (in_q && len < maxn) ? (sb < qb ? 1u : 0u) : 0u;
			if (len > maxn) len = maxn;
			*out_len = bcast(len, first);
			*out_less = bcast(less, first);
			return;
		}
		base += 64u * 16u;
	}
}

// Every lane with `on` compares the query (qp, n bytes) with its own suffix S + sa: alone for the
// first LONE bytes, then — one lane after the other — with the whole wavefront's help.
static __device__ void lanes_compare(const uint8_t *qp, uint32_t n, const uint8_t *S, uint32_t sa, bool on,
									 const uint8_t *s_end, uint32_t *out_len, uint32_t *out_less)
{
	const uint32_t LONE = 256;
	uint32_t len = 0, less = 0, i = 0;
	bool open = on;
	while (open) { // 32 bytes per trip: the two loads of each side are in flight together
		if (i >= n) {
			len = n;
			less = 0;
			open = false;
			break;
		}
		if (i >= LONE) break;
		U4 a[2], b[2];
#pragma unroll
		for (int k = 0; k < 2; k++) {
			a[k] = lg16(qp + i + 16 * k);
			b[k] = U4{0, 0, 0, 0};
			if (S + sa + i + 16 * k + 16 <= s_end) b[k] = lg16(S + sa + i + 16 * k);
		}
		uint32_t d = 32, qb = 0, sb = 0;
#pragma unroll
		for (int k = 1; k >= 0; k--) {
			const uint32_t dk = first_diff(a[k], b[k]);
			if (dk < 16) {
				d = 16u * (uint32_t)k + dk;
				qb = byte_at(a[k], dk);
				sb = byte_at(b[k], dk);
			}
		}
		if (d < 32) {
			len = i + d;
			if (len >= n) {
				len = n;
				less = 0;
			} else {
				less = sb < qb ? 1u : 0u;
			}
			open = false;
			break;
		}
		i += 32;
	}
	uint64_t longm = __ballot(open);
	while (longm) {
		const int who = __ffsll((unsigned long long)longm) - 1;
		const uint32_t wsa = bcast(sa, who);
		uint32_t l2, s2;
		wave_compare(qp, S + wsa, LONE, n, s_end, &l2, &s2);
		if ((int)lane64() == who) {
			len = l2;
			less = s2;
		}
		longm &= longm - 1;
	}
	*out_len = len;
	*out_less = less;
}

// One whole step of lane `leader`, from the raw bytes, by the definition (lean_core.h:
// lean_resolve_scalar is the same in plain loops): lucky_anchor; else the insertion point of the
// query suffix among the suffixes of S — 64 probes per round — the longer of its two neighbours'
// matches, unique iff the LCP array says the next suffix outward does not share it.
static __device__ void coop_resolve(LeanLane &ln, int leader, const PhaseA &A, const RefIndex &R, const LeanIndex &X,
									const uint8_t *s_end)
{
	const uint32_t lane = lane64();
	const uint32_t qw0 = bcast(ln.qw0, leader), q = bcast(ln.q, leader), qlen = bcast(ln.qlen, leader);
	const uint32_t lq = bcast(ln.lq, leader), ls = bcast(ln.ls, leader), ll = bcast(ln.ll, leader);
	const uint32_t q_cap = bcast(ln.q_cap, leader);
	const uint8_t *Q = A.qbase + ((uint64_t)qw0 << 4) + q;
	const uint32_t n = qlen - q;
	// A speculative chain's comparisons stop at its cap (lean_core.h: q_cap) when the step's outcome no
	// longer depends on the length: first with the cap; if that leaves anything open, once more without.
	uint32_t cap_rel = (q_cap != NO_BAD && q_cap > q && q_cap - q < n) ? q_cap - q : NO_BAD;
	// A window under an over-deep entry of the reference's 6-mer cache (lean_core.h: LeanIndex::quirk): the search
	// goes on below the entry's interval [in_lo, in_hi) from depth `skip` — get_match_from(query, qlen, ij.l, ij),
	// src/esa.cxx:556-562 — whatever the query holds before that.  (Tiny subjects only; no cap there.)
	uint32_t skip = 0, in_lo = 0, in_hi = R.n;
	const uint32_t qe = X.nquirk ? lean_quirk_lookup(X, Q, n) : LEAN_NO_QUIRK;
	if (qe != LEAN_NO_QUIRK) {
		const U4 e = X.quirk[qe];
		skip = e.y >> 8;
		in_lo = e.z;
		in_hi = e.w;
		cap_rel = NO_BAD;
	}
	uint32_t r_pos = 0, r_len = 0;
	bool r_acc = false, r_cut = false;
	for (int pass = 0; pass < 2; pass++) {
		const uint32_t neff = (pass == 0 && cap_rel != NO_BAD) ? cap_rel : n;
		const bool capped = neff < n;
		bool redo = false;
		const uint32_t advance = q - lq;
		if ((ls + advance < R.n) && (advance - ll <= R.threshold)) { // lucky_anchor, process.cxx:227-242
			uint32_t len, less;
			wave_compare(Q, R.S + (ls + advance), 0, neff, s_end, &len, &less);
			if (len >= R.threshold) {
				r_pos = ls + advance;
				r_len = len;
				r_acc = true;
				r_cut = capped && len >= neff;
				break;
			}
		}
		uint32_t lo = in_lo, hi = in_hi;
		if (qe == LEAN_NO_QUIRK) { // the k-mer's bucket bounds the search when the window starts with k nucleotides
			uint32_t valid;
			const uint32_t code = window_code(lg16(Q), &valid);
			const uint32_t qv = valid < n ? valid : n;
			if (qv >= R.k) {
				const U2 hdr = lg8((const uint8_t *)(R.T + (size_t)(code >> (2u * (16u - R.k)))));
				lo = hdr.x;
				hi = hdr.y;
			}
		}
		const uint8_t *const Qs = Q + skip, *const Ss = R.S + skip; // comparisons start behind the bytes taken as matched
		const uint32_t ns = neff - skip;
		while (hi - lo > 62u) {
			const uint32_t m = hi - lo;
			const uint32_t r = lo + (uint32_t)(((uint64_t)(lane + 1u) * m) / 65u); // lo <= r < hi, nondecreasing in lane
			const uint32_t sa = lg16((const uint8_t *)(R.SAX + r)).x;
			uint32_t len, less;
			lanes_compare(Qs, ns, Ss, sa, true, s_end, &len, &less);
			if (capped && __ballot(len >= neff)) { // a probe ran into the cap: its order is unknown
				redo = true;
				break;
			}
			const uint32_t cnt = (uint32_t)__popcll(__ballot(less != 0));
			const uint32_t nlo = cnt ? bcast(r, (int)cnt - 1) + 1u : lo;
			const uint32_t nhi = cnt < 64u ? bcast(r, (int)cnt) : hi;
			lo = nlo;
			hi = nhi;
		}
		if (redo) continue;
		const uint32_t m = hi - lo; // lanes 0 .. m+1 take ranks lo-1 .. hi
		// (a rank outside an over-deep interval shares less than its depth with the window as the reference reads it:
		// never the better neighbour, its length stays 0)
		const bool on = lane < m + 2u && !(lo == 0 && lane == 0) && lo + lane - 1u < R.n && lo + lane - 1u >= in_lo && lo + lane - 1u < in_hi;
		const uint32_t rank = on ? lo + lane - 1u : 0u;
		const uint32_t sa = lg16((const uint8_t *)(R.SAX + rank)).x;
		uint32_t len, less;
		lanes_compare(Qs, ns, Ss, sa, on, s_end, &len, &less);
		if (on) len += skip;
		if (capped) {
			const uint64_t hit = __ballot(on && len >= neff);
			if (hit) {
				// one suffix alone reaches the cap: it is the best neighbour whichever side it is on; unique
				// iff neither of its neighbours in the suffix array shares that much with it
				if ((hit & (hit - 1)) == 0) {
					const int who = __ffsll((unsigned long long)hit) - 1;
					const uint32_t wr = bcast(rank, who);
					if (R.LCP[wr] < neff && R.LCP[wr + 1] < neff) {
						r_pos = bcast(sa, who);
						r_len = neff;
						r_acc = true;
						r_cut = true;
						break;
					}
				}
				continue; // open: once more without the cap
			}
		}
		const uint32_t cnt = (uint32_t)__popcll(__ballot(on && lane >= 1u && lane <= m && less != 0));
		const uint32_t ins = lo + cnt;
		const uint32_t lp = ins > 0 ? bcast(len, (int)cnt) : 0u, pp = bcast(sa, (int)cnt);
		const uint32_t lsu = ins < R.n ? bcast(len, (int)cnt + 1) : 0u, ps = bcast(sa, (int)cnt + 1);
		const bool pbest = lp > lsu;
		const uint32_t lmax = pbest ? lp : lsu;
		const bool cand = lp != lsu && lmax >= R.threshold;
		uint32_t l = 0;
		if (cand) l = pbest ? R.LCP[ins - 1] : R.LCP[ins + 1];
		r_pos = pbest ? pp : ps;
		r_len = lmax;
		r_acc = cand && l < lmax;
		break;
	}
	if ((int)lane == leader) {
		ln.finish(r_pos, r_len, r_acc);
		ln.ovr = r_cut;
	}
}

// the long tail of one comparison (lean_ext handed it over at e_pos bases); a speculative chain's
// comparison stops at its cap when it may (lean_core.h: may_cut)
static __device__ void coop_ext(LeanLane &ln, int leader, const PhaseA &A, const RefIndex &R, const uint8_t *s_end)
{
	const uint32_t qw0 = bcast(ln.qw0, leader), q = bcast(ln.q, leader), qlen = bcast(ln.qlen, leader);
	const uint32_t e_p = bcast(ln.e_p, leader), e_pos = bcast(ln.e_pos, leader), q_cap = bcast(ln.q_cap, leader);
	const bool cut_ok = bcast(ln.q_cap != NO_BAD && ln.may_cut(ln.q_cap - ln.q) ? 1u : 0u, leader) != 0;
	const uint8_t *Q = A.qbase + ((uint64_t)qw0 << 4) + q;
	const uint32_t n = qlen - q;
	const uint32_t lim = cut_ok && q_cap - q < n ? q_cap - q : n;
	uint32_t len, less;
	wave_compare(Q, R.S + e_p, e_pos & ~15u, lim, s_end, &len, &less);
	if ((int)lane64() == leader) {
		if (len >= lim && lim < n) ln.finish_cut(lim);
		else lean_deliver(ln, R, len, less);
	}
}

// The open ends of the comparisons that speculative chains cut (lean_core.h: overruns), in two passes.
// Pass 1, one lane per chunk: does the next chunk continue this chunk's cut match (then its end is that
// chunk's end: LINK bit), or is this the last chunk of such a run — then the match is compared on, by
// the whole wavefront, at most two chunk lengths, and the chunk is closed except for its flag bits
// (neighbouring lanes still read them).
__global__ __launch_bounds__(64) void lean_overrun_direct_kernel(PhaseA A, RefIndex R, uint32_t c_lo, uint32_t c_hi)
{
	if (*A.overrun == 0) return;
	const uint32_t lane = lane64();
	const uint32_t gc0 = c_lo + blockIdx.x * 64u + lane; // chunks [c_lo, c_hi): whole queries
	const bool on = gc0 < c_hi;
	const uint32_t gc = on ? gc0 : 0u;
	const uint8_t *s_end = R.S + R.n + 64;
	const uint32_t j = A.chunk_query[gc];
	LeanOverrun o = lean_overrun_of(A, gc);
	if (!on) o.flagged = 0;
	bool link = false;
	if (o.flagged && gc + 1u < A.qchunk0[j + 1]) {
		const LeanOverrun nx = lean_overrun_of(A, gc + 1u);
		link = lean_overrun_links(o, nx, chunk_geom(A, j, gc + 1u - A.qchunk0[j]).q0);
	}
	if (link) A.spec_cnt[gc] |= LEAN_LINK_BIT;
	uint64_t direct = __ballot(o.flagged && !link);
	while (direct) {
		const int who = __ffsll((unsigned long long)direct) - 1;
		const uint32_t wj = bcast(j, who), wq = bcast(o.q_s, who), wp = bcast(o.pos, who), wv = bcast(o.verified, who);
		uint32_t len, less;
		wave_compare(A.qbase + A.qoff[wj] + wq, R.S + wp, wv & ~15u, A.qlen[wj] - wq, s_end, &len, &less);
		if ((int)lane == who) {
			lean_overrun_close(A, j, gc, o, wq + len, false);
			atomicAdd(A.overrun + 1, 1u); // statistics: runs closed, bytes compared for them
			atomicAdd(A.overrun + 2, len - (wv & ~15u));
		}
		direct &= direct - 1;
	}
}

// Pass 2, one wavefront per query, its chunks from the last to the first, 64 at a time: a linked chunk
// takes the end of the nearest closed chunk above it (pointer doubling inside the group, a carry between
// groups); all flag bits are cleared.
__global__ __launch_bounds__(64) void lean_overrun_chain_kernel(PhaseA A, uint32_t j0, uint32_t j1)
{
	if (*A.overrun == 0) return;
	const uint32_t j = j0 + blockIdx.x; // queries [j0, j1)
	if (j >= j1) return;
	const uint32_t c0 = A.qchunk0[j], c1 = A.qchunk0[j + 1];
	const uint32_t lane = lane64();
	uint32_t car_end = 0;
	for (uint32_t top = c1; top > c0;) {
		const uint32_t base = top - c0 >= 64u ? top - 64u : c0; // this group: chunks [base, top)
		const uint32_t m = top - base;
		const bool on = lane < m;
		const uint32_t gc = base + (on ? lane : 0u);
		const uint32_t word = on ? A.spec_cnt[gc] : 0u;
		const SpecExit x = A.spec_exit[gc];
		const bool flagged = (word & LEAN_OVERRUN_BIT) != 0, link = (word & LEAN_LINK_BIT) != 0;
		uint32_t end = x.lq + x.ll; // a closed chunk's end (others: overwritten below or unused)
		uint32_t known = link ? 0u : 1u, hop = 1;
		if (link && lane == m - 1u) {
			end = car_end;
			known = 1;
		}
#pragma unroll
		for (int it = 0; it < 6; it++) { // an unknown lane's end is the end of lane + hop
			const int t = (int)(lane + hop < 64u ? lane + hop : 63u);
			const uint32_t tk = bcast(known, t), te = bcast(end, t), th = bcast(hop, t);
			if (!known) {
				if (tk) {
					end = te;
					known = 1;
				} else {
					hop += th;
				}
			}
		}
		if (flagged) {
			if (link) {
				const LeanOverrun o = {1u, x.lq, x.ls, x.ll};
				lean_overrun_close(A, j, gc, o, end, true);
			} else {
				A.spec_cnt[gc] = word & LEAN_COUNT_MASK;
			}
		}
		car_end = bcast(end, 0);
		top = base;
	}
}

// A lane's finished words of the visited bitmap, collected in LDS until the 64-byte group they belong to is
// complete (or the chunk ends): one full 64-byte store per 512 positions instead of sixteen lone 4-byte ones.
// Words the chain jumped over are zero, as the cleared bitmap has them.
struct VisLds {
	uint32_t (*buf)[256]; // [16][256], dword-major like the ring
	uint32_t *visited;
	uint32_t tid;
	uint32_t group; // word index / 16 of the group being collected; NO_BAD: none
	uint32_t lo, hi; // the chunk's own words [lo, hi): a group that reaches beyond them is shared with a neighbouring chunk's lane
	__device__ void flush()
	{
		if (group == NO_BAD) return;
		const uint32_t w0 = group * 16u;
		if (w0 >= lo && w0 + 16u <= hi) {
			U4 *dst = (U4 *)(visited + (size_t)w0);
#pragma unroll
			for (int i = 0; i < 4; i++) {
				dst[i] = U4{buf[4 * i][tid], buf[4 * i + 1][tid], buf[4 * i + 2][tid], buf[4 * i + 3][tid]};
				buf[4 * i][tid] = buf[4 * i + 1][tid] = buf[4 * i + 2][tid] = buf[4 * i + 3][tid] = 0;
			}
		} else { // the chunk's first or last group: only its own words
			for (uint32_t i = 0; i < 16u; i++) {
				if (w0 + i >= lo && w0 + i < hi) visited[w0 + i] = buf[i][tid];
				buf[i][tid] = 0;
			}
		}
		group = NO_BAD;
	}
	__device__ void put(uint32_t idx, uint32_t bits)
	{
		if ((idx >> 4) != group) {
			flush();
			group = idx >> 4;
		}
		buf[idx & 15u][tid] = bits;
	}
	__device__ void close() { flush(); }
	// a word of the group being collected (or the first word at all): no flush
	__device__ bool cheap(uint32_t idx) const { return group == NO_BAD || (idx >> 4) == group; }
	__device__ void put_cheap(uint32_t idx, uint32_t bits)
	{
		group = idx >> 4;
		buf[idx & 15u][tid] = bits;
	}
};

struct LeanAlloc {
	const PhaseA *A;
	__device__ uint32_t operator()() const
	{
		const uint32_t b = atomicAdd(A->pool_next, 1u);
		return b < A->pool_blocks ? b : NO_BLOCK;
	}
};

// The speculative kernel's work queue in the form a lane starts from (anchor_core.h: WorkItem, QDesc): made once per plan.
__global__ __launch_bounds__(256) void lean_work_kernel(PhaseA A, LeanIndex X, WorkItem *__restrict__ work, QDesc *__restrict__ qdesc, uint32_t nq)
{
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t < A.nchunks) work[t] = LeanSpec::make_item(A, X, A.items[t]);
	if (t < nq) qdesc[t] = LeanSpec::make_qdesc(A, X, t);
}
void launch_lean_work(const PhaseA &A, const LeanIndex &X, WorkItem *work, QDesc *qdesc, uint32_t nq, hipStream_t st)
{
	const uint32_t n = A.nchunks > nq ? A.nchunks : nq;
	if (!n) return;
	hipLaunchKernelGGL(lean_work_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A, X, work, qdesc, nq);
}

// STEP-only trips between two full ones (0: none), taken while at least NUM / DEN of the wavefront's live lanes are in
// STEP.  Same-box A/B on C3, anchor_spec / anchor_bridge: none 3.14 / 0.45 ms; 1 trip at 3/4 2.79 / 0.36; 2 at 1/2
// 2.78 / 0.36; 4 at 1/2 2.76 / 0.345; 4 at 1/4 2.75 / 0.344; 8 at 1/4 2.91 / 0.37 (C5: 19.1 -> 16.4 / 0.74 -> 0.50;
// c2like 0.568 -> 0.503; C4 11.7 -> 10.2).
static const int FAST_TRIPS = 4, FAST_NUM = 1, FAST_DEN = 4;
static const uint32_t PRIO_ROT = 16u; // trips between two turns of the wavefronts' issue priorities
// Every chunk's exit state is continued into the chunk behind it until the continuation stands on a position that
// chunk's own chain visited in an equivalent state (LeanBridge::begin_step).  Where that is the case at once — the
// chunk's last match ran on into the next chunk and that chunk's chain found the same match — there is nothing to
// walk: one thread per chunk finds out, writes those bridges' records, and leaves the others started and packed
// (LeanBridge::pack + the first words of the query) for the lanes of lean_chain_kernel<1>.
__global__ __launch_bounds__(256) void lean_bridge_prepare_kernel(PhaseA A, RefIndex R, LeanIndex X, BridgeZero Z)
{
	const uint32_t gc = blockIdx.x * 256u + threadIdx.x;
	// (counters of the kernels that follow — the fold's list lengths, the filter's total, the projection's flags —, zeroed
	// here instead of by a fill each)
	for (uint32_t i = gc; i < Z.n[0]; i += gridDim.x * 256u) Z.p[0][i] = 0;
	if (gc < Z.n[1]) Z.p[1][gc] = 0;
	if (gc < Z.n[2]) Z.p[2][gc] = 0;
	if (gc >= A.nchunks) return;
	LeanBridge L;
	L.start(A, X, gc);
	if (!L.begin_step(A, X, R)) return;
	const uint32_t slot = atomicAdd(A.bridge_todo, 1u);
	uint32_t w[LeanBridge::PACKED_WORDS];
	L.pack(w);
	const uint32_t *q2 = X.Q2 + L.ln.qw0 + (L.ln.q >> 4);
#pragma unroll
	for (uint32_t i = 0; i < LeanBridge::PACKED_RING; i++) w[24 + i] = q2[i];
	U4 *dst = (U4 *)(A.bridge_start + (size_t)slot * LeanBridge::PACKED_WORDS);
#pragma unroll
	for (int i = 0; i < 8; i++) dst[i] = U4{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
}

// MODE 0: speculative chunk chains, the work items are A.items[0 .. nchunks) (the plan's work order).  MODE 1: the
// bridges lean_bridge_prepare_kernel left to walk.  Persistent lanes with dynamic work fetch through A.fetch[MODE].
template <int MODE>
__global__ __launch_bounds__(256) void lean_chain_kernel(PhaseA A, RefIndex R, LeanIndex X)
{
	typename std::conditional<MODE == 0, LeanSpec, LeanBridge>::type L;
	LeanLane &ln = L.ln;
	// a lane's ring, dword-major so that lane l always hits LDS bank l
	__shared__ uint32_t ring[LEAN_RING_WORDS][256];
	__shared__ uint32_t visbuf[MODE == 0 ? 16 : 1][256];
	const uint32_t tid = threadIdx.x;
	const uint8_t *s_end = R.S + R.n + 64;
	const uint8_t *const slot_b = (const uint8_t *)R.SLOT, *const sax_b = (const uint8_t *)R.SAX,
						 *const q2_b = (const uint8_t *)X.Q2, *const s2_b = (const uint8_t *)X.S2;
	VisLds vis = {visbuf, A.visited, tid, NO_BAD, 0u, 0u};
	if constexpr (MODE == 0) {
#pragma unroll
		for (int i = 0; i < 16; i++) visbuf[i][tid] = 0;
	}
	bool active = false, done = false;
	uint32_t trip = 0;
	// The next work item, fetched ahead (MODE 0).  A lane that has run out of work and asks the queue then — counter, item,
	// the query's descriptor: three dependent loads — keeps the other 63 lanes of its wavefront waiting, sixty-four times
	// a round.  Instead a lane claims its next item when it is an eighth of a chunk from the end of this one, and the
	// trips that follow bring the item and the descriptor in beside their own loads; at the chunk's end it starts from
	// registers.  (pf: 0 nothing claimed, 1 the counter's answer is on its way, 2 the item, 3 the descriptor — ready,
	// 4 the queue is empty.)  queue_empty: a lane of this wavefront has seen the end of the queue.
	uint32_t pf = 0, nx_it = 0;
	WorkItem nx_w = {0, 0, 0, 0};
	QDesc nx_d = {0, 0, 0, 0, 0, 0, 0, 0};
	bool queue_empty = false;
	const uint32_t prio_pass = (uint32_t)(((uint64_t)blockIdx.x * 4u) / gridDim.x);
	LeanAlloc alloc = {&A};
	ln.fin = false;
	ln.ph = LP_STEP;
	const uint32_t n_items = MODE == 0 ? A.nchunks : *A.bridge_todo;
#ifdef PHY_LEAN_TIMING
	const unsigned long long t_wave0 = __builtin_amdgcn_s_memrealtime();
	unsigned long long first_query = ~0ull;
	uint32_t bsteps = 0;
	unsigned long long ext_cnt[6] = {0, 0, 0, 0, 0, 0};
	unsigned long long phase_trips[8] = {0, 0, 0, 0, 0, 0, 0, 0}, phase_lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long tm[6] = {0, 0, 0, 0, 0, 0}, t_prev = clock64();
#define LEAN_TICK(i)                                                                                                    \
	{                                                                                                                   \
		__builtin_amdgcn_s_waitcnt(0);                                                                                  \
		const unsigned long long t_now = clock64();                                                                     \
		tm[i] += t_now - t_prev;                                                                                        \
		t_prev = t_now;                                                                                                 \
	}
#else
#define LEAN_TICK(i)
#endif

	for (;;) {
		LEAN_TICK(4)
		// finish the previous step, start the next one or the next chunk
		if (active && ln.fin) {
			if constexpr (MODE == 0) L.step_done(A);
			else L.step_done(A, alloc);
			ln.fin = false;
		}
		if (active && ln.ph == LP_STEP) {
			if constexpr (MODE == 0) active = L.begin_step(A, X, vis);
			else active = L.begin_step(A, X, R);
#ifdef PHY_LEAN_TIMING
			if constexpr (MODE == 1) { // how many steps the bridges that end here took
				if (active) {
					bsteps++;
				} else if (X.dbg) {
					const uint32_t b = bsteps <= 2u ? bsteps : bsteps <= 4u ? 3u : bsteps <= 8u ? 4u : bsteps <= 16u ? 5u : bsteps <= 32u ? 6u : bsteps <= 64u ? 7u : 8u;
					atomicAdd(&X.dbg[16 + 4 * 8192 + 64 + b], 1ull);
					atomicMax(&X.dbg[16 + 4 * 8192 + 64 + 9], (unsigned long long)bsteps);
				}
			}
#endif
		}
		if constexpr (MODE == 0) {
			// the next item on its way: one stage a trip
			if (__any(pf == 1u || pf == 2u)) {
				if (pf == 2u) {
					const U4 *dp = (const U4 *)(A.qdesc + nx_w.j);
					const U4 d0 = dp[0], d1 = dp[1];
					nx_d = QDesc{d0.x, d0.y, d0.z, d0.w, d1.x, 0u, 0u, 0u};
					pf = 3u;
				} else if (pf == 1u) {
					if (nx_it >= n_items) {
						pf = 4u;
					} else {
						const U4 v = *(const U4 *)(A.work + nx_it);
						nx_w = WorkItem{v.x, v.y, v.z, v.w};
						pf = 2u;
					}
				}
			}
			if (active && pf == 0u && !queue_empty && ln.ph == LP_STEP && ln.q + (A.C >> 3) >= L.q_end) {
				nx_it = atomicAdd(&A.fetch[0], 1u);
				pf = 1u;
			}
		}
		if (!active && !done) {
			uint32_t it = n_items;
			if constexpr (MODE == 0) {
				if (pf == 3u) it = nx_it;                                      // ready: no load below
				else if (pf == 1u || pf == 2u) it = nx_it;                      // (claimed late: the rest of the chain now)
				else if (pf == 0u && !queue_empty) it = atomicAdd(&A.fetch[0], 1u); // (the first item, or a chunk shorter than the lead)
			} else {
				if (!queue_empty) it = atomicAdd(&A.fetch[MODE], 1u);
			}
			done = it >= n_items;
			if (!done) {
				if constexpr (MODE == 0) {
					if (pf != 3u) {
						if (pf != 2u) {
							const U4 v = *(const U4 *)(A.work + it);
							nx_w = WorkItem{v.x, v.y, v.z, v.w};
						}
						const U4 *dp = (const U4 *)(A.qdesc + nx_w.j);
						const U4 d0 = dp[0], d1 = dp[1];
						nx_d = QDesc{d0.x, d0.y, d0.z, d0.w, d1.x, 0u, 0u, 0u};
					}
					pf = 0u;
					L.start_desc(A, nx_w, nx_d);
#ifdef PHY_LEAN_TIMING
					if (first_query == ~0ull) first_query = nx_w.j;
#endif
					vis.lo = L.vis_idx; // the chunk's words: from its first position to its end (chunks are multiples of 64 positions)
					vis.hi = lean_visited_word(ln, L.q_end_full);
					active = L.begin_step(A, X, vis);
				} else {
					// a bridge as lean_bridge_prepare_kernel left it: started, past its first begin_step, its ring filled
					uint32_t w[LeanBridge::PACKED_WORDS];
					const uint8_t *rec = (const uint8_t *)(A.bridge_start + (size_t)it * LeanBridge::PACKED_WORDS);
#pragma unroll
					for (int i = 0; i < 8; i++) {
						const U4 v = lg16(rec + 16 * i);
						w[4 * i] = v.x, w[4 * i + 1] = v.y, w[4 * i + 2] = v.z, w[4 * i + 3] = v.w;
					}
					L.unpack(A, w);
#ifdef PHY_LEAN_TIMING
					bsteps = 1;
#endif
#pragma unroll
					for (uint32_t i = 0; i < LeanBridge::PACKED_RING; i++) ring[(ln.wb + i) & 15u][tid] = w[24 + i];
					active = true;
				}
			}
		}
		queue_empty = queue_empty || __any(done || pf == 4u);
		if (__all(done && !active)) break;
		LEAN_TICK(0)

		// Fast trips.  A trip of the loop below carries the code of every phase one of the wavefront's lanes is in, five
		// loads and their unpacking; most lanes, most of the time, are in STEP.  While nearly all of the wavefront's
		// lanes are, up to FAST_TRIPS trips are taken here that know nothing else: window, the slot's 16 bytes (and
		// the lucky window), lean_step, and a finished step's bookkeeping.  The few lanes in another phase (an extension,
		// a bucket walk, a refill, the slow resolver) sit these out and have their turn in the full trip that follows.
		bool need_bs = false; // a lane whose step has finished and whose begin_step is still owed
		for (int f = 0; f < FAST_TRIPS; f++) {
			bool go = active && !need_bs && ln.ph == LP_STEP && !X.force_slow;
			if (go) go = lean_step_phase(ln, X) == LP_STEP;
			const uint32_t n_go = (uint32_t)__popcll(__ballot(go)), n_act = (uint32_t)__popcll(__ballot(active));
			if (!n_go || n_go * FAST_DEN < n_act * FAST_NUM) break;
			const uint8_t *fA = s2_b, *fY = s2_b, *fV = s2_b;
			if (go) {
				const uint32_t w = ln.q >> 4;
				ln.qcode = code_window(ring[w & 15u][tid], ring[(w + 1u) & 15u][tid], ln.q & 15u);
				fA = slot_b + (uint64_t)(ln.qcode >> (2u * (16u - R.k))) * 16u;
				if (ln.lucky_ok(R)) fY = s2_b + (uint64_t)((ln.ls + (ln.q - ln.lq)) >> 4) * 4u;
				if constexpr (MODE == 1) fV = (const uint8_t *)(A.visited + L.vw_idx + 1u); // (LeanBridge: pv_base)
			}
			const U4 fx = lg16(fA);
			const U2 fy = lg8(fY);
			if constexpr (MODE == 1) {
				const U2 fv = lg8(fV);
				if (go) L.pv_base = L.vw_idx + 1u, L.pv0 = fv.x, L.pv1 = fv.y;
			}
			if (go) {
				const uint32_t fd[4] = {fx.x, fx.y, fx.z, fx.w};
				lean_step(ln, R, X, fd, fy.x, fy.y);
				if (ln.fin) {
					if constexpr (MODE == 0) L.step_done(A);
					else L.step_done(A, alloc);
					ln.fin = false;
					// (the speculative chains' rarer bookkeeping — a group of visited words to flush, the chunk's end —
					// waits for the end of the fast trips: once per lane that needs it instead of its code in every trip)
					if constexpr (MODE == 0) need_bs = !L.begin_step_fast(X, vis);
					else active = L.begin_step(A, X, R);
				}
			}
			trip++;
		}
		if constexpr (MODE == 0) {
			if (__any(need_bs)) {
				if (need_bs) active = L.begin_step(A, X, vis);
				need_bs = false;
			}
		}
		if (__all(done && !active)) break;
		// which phase is the lane in this trip; a STEP needs its window from the ring
		uint32_t ph = active ? ln.ph : (uint32_t)LP_SLOW + 8u;
		if (active && ph == LP_STEP) {
			ph = lean_step_phase(ln, X);
			ln.ph = ph;
			if (ph == LP_STEP) {
				const uint32_t w = ln.q >> 4;
				ln.qcode = code_window(ring[w & 15u][tid], ring[(w + 1u) & 15u][tid], ln.q & 15u);
			}
		}
		// Look-ahead for the bridge kernel's tail.  Once the queue is empty a wavefront is left with a few long walks —
		// two chains on sequence that does not align meet by chance, one step in thirteen — and runs whole trips for
		// them.  While the walker's last anchor is out of lucky_anchor's reach (process.cxx:227-242) and it accepts
		// none, its step at a position is a function of that position alone (anchor(), process.cxx:219-225): the lanes
		// without a bridge work out the steps at the positions q + 1, q + 2, ... in the same trip — same ring, a slot
		// each — and the walker then follows its own steps through them, testing for the merge at every landing, as
		// many steps as stay inside what was looked at (~5) instead of one.
		int look_leader = -1;
		uint32_t look_k = 0, look_q0 = 0, look_vis0 = 0;
		if constexpr (MODE == 1) {
			const uint64_t live = __ballot(active), idle = __ballot(!active);
			if (__ballot(!active && !done) == 0 && live && (uint32_t)__popcll(live) <= LEAN_LOOK_MAX_LIVE && !X.force_slow) {
				const uint64_t can = __ballot(active && ph == LP_STEP && !ln.lucky_ok(R) && L.cur_gc != BRIDGE_END);
				if (can) {
					const uint32_t rot = trip & 63u; // the walkers take turns
					const uint64_t hi = can & ~((1ull << rot) - 1ull);
					look_leader = __ffsll((unsigned long long)(hi ? hi : can)) - 1;
					look_q0 = bcast(ln.q, look_leader);
					const uint32_t l_wb = bcast(ln.wb, look_leader), l_we = bcast(ln.we, look_leader), l_qlen = bcast(ln.qlen, look_leader);
					const uint32_t l_qb = bcast(ln.qb_next, look_leader);
					const uint32_t l_c0 = bcast(L.cur_q0, look_leader), l_cl = bcast(L.cur_len, look_leader);
					look_vis0 = bcast(lean_visited_word(ln, ln.q), look_leader);
					if (!active) {
						const uint32_t k = (uint32_t)__popcll(idle & ((1ull << lane64()) - 1ull)) + 1u; // this helper's position: q0 + k
						const uint32_t hq = look_q0 + k, w = hq >> 4;
						// a position whose step the walker could take on the packed path from where it stands: the window in
						// its ring, pure nucleotides with a base to spare, the same chunk (the same visited words' owner)
						const bool fits = w >= l_wb && w + 1u < l_we && hq + 17u <= l_qlen && (l_qb == NO_BAD || l_qb >= hq + 16u) &&
										  hq - l_c0 < l_cl && hq - look_q0 < 96u;
						if (fits) {
							const uint32_t ltid = (tid & ~63u) + (uint32_t)look_leader;
							ln.q = hq;
							ln.qcode = code_window(ring[w & 15u][ltid], ring[(w + 1u) & 15u][ltid], hq & 15u);
							ln.fin = false;
							ln.ph = LP_STEP;
							look_k = k;
							ph = LP_LOOK;
						}
					}
				}
			}
		}
		trip++;
		// The SIMD's arbiter breaks ties between ready wavefronts by age: of the three or four chain wavefronts a SIMD holds,
		// the one dispatched first issues first, trip after trip, and the last one is left the gaps — same trips, 20 % longer
		// (measured per wavefront).  Rotating priorities even that out.
		if ((trip & (PRIO_ROT - 1u)) == 0u) {
			const uint32_t pr = ((trip / PRIO_ROT) + prio_pass) & 3u;
			if (pr == 0u) __builtin_amdgcn_s_setprio(0);
			else if (pr == 1u) __builtin_amdgcn_s_setprio(1);
			else if (pr == 2u) __builtin_amdgcn_s_setprio(2);
			else __builtin_amdgcn_s_setprio(3);
		}
		// one batch of loads for every phase
		// (a STEP / SEARCH lane needs the 16 bytes of its slot only: the batch's other loads go to one address that
		// every such lane shares)
		const uint8_t *pA = s2_b, *pB = s2_b, *pY = s2_b;
		uint32_t a1 = 0; // where the second 16 bytes of pA's 32 lie
		if (ph == LP_STEP || ph == LP_SEARCH || ph == LP_LOOK) {
			pA = slot_b + (uint64_t)(ln.qcode >> (2u * (16u - R.k))) * 16u;
			if (ph == LP_STEP && ln.lucky_ok(R)) pY = s2_b + (uint64_t)((ln.ls + (ln.q - ln.lq)) >> 4) * 4u;
			if constexpr (MODE == 1) { // the visited bits of the positions ahead: for the walker's next begin_step (LeanBridge: pv_base) ...
				if (ph == LP_STEP) pB = (const uint8_t *)(A.visited + L.vw_idx);
				if ((int)lane64() == look_leader) pB = (const uint8_t *)(A.visited + look_vis0); // ... and for the one that is looked ahead for
			}
		} else if (ph == LP_EXT) {
			const uint32_t e0 = ln.e_pos - ((ln.q + ln.e_pos) & 15u);
			pA = q2_b + ((uint64_t)ln.qw0 + ((ln.q + e0) >> 4)) * 4u;
			pB = s2_b + (uint64_t)((ln.e_p + e0) >> 4) * 4u;
			pY = pB + 32;
			a1 = 16;
		} else if (ph == LP_SCAN) {
			pA = sax_b + (uint64_t)ln.s_rank * 16u;
			pB = pA + 32;
			a1 = 16;
		} else if (ph == LP_REFILL) {
			pA = q2_b + ((uint64_t)ln.qw0 + (ln.q >> 4)) * 4u;
			pB = pA + 32;
			a1 = 16;
		}
		LEAN_TICK(1)
#ifdef PHY_LEAN_TIMING
		for (uint32_t p = 0; p < 7; p++) {
			const unsigned long long m = __ballot(ph == p);
			if (m) phase_trips[p]++;
			phase_lanes[p] += (unsigned long long)__popcll(m);
		}
		if (__ballot(active && ph == LP_STEP && ln.lucky_ok(R))) phase_trips[7]++;
#endif
		uint32_t d[16], y[2];
		{
			const U4 x0 = lg16(pA), x1 = lg16(pA + a1), x2 = lg16(pB), x3 = lg16(pB + 16);
			const U2 yy = lg8(pY);
			y[0] = yy.x, y[1] = yy.y;
			d[0] = x0.x, d[1] = x0.y, d[2] = x0.z, d[3] = x0.w;
			d[4] = x1.x, d[5] = x1.y, d[6] = x1.z, d[7] = x1.w;
			d[8] = x2.x, d[9] = x2.y, d[10] = x2.z, d[11] = x2.w;
			d[12] = x3.x, d[13] = x3.y, d[14] = x3.z, d[15] = x3.w;
		}
		LEAN_TICK(2)
		// digest
		if (ph == LP_STEP || ph == LP_SEARCH || ph == LP_LOOK) { // (one body for the three: lean_step_any)
			if constexpr (MODE == 1) {
				if (ph == LP_STEP) L.pv_base = L.vw_idx + 1u, L.pv0 = d[9], L.pv1 = d[10];
			}
			lean_step_any(ln, R, X, d, y[0], y[1], ph == LP_STEP);
		} else if (ph == LP_EXT) {
			uint32_t sw[9];
			ln.wb = (ln.q + (ln.e_pos - ((ln.q + ln.e_pos) & 15u))) >> 4;
			ln.we = ln.wb + 8;
#pragma unroll
			for (int i = 0; i < 8; i++) {
				sw[i] = d[8 + i];
				ring[(ln.wb + (uint32_t)i) & 15u][tid] = d[i]; // the query words double as the ring's new content
			}
			sw[8] = y[0];
#ifdef PHY_LEAN_TIMING
			const uint32_t ext_first = ln.e_pos == 16u ? 1u : 0u, ext_kind = ln.e_kind == EXT_LUCKY ? 1u : 0u;
#endif
			lean_ext(ln, R, X, d, sw);
#ifdef PHY_LEAN_TIMING
			{
				const bool over = ln.ph != LP_EXT; // the comparison ended in this trip
				const uint32_t len = ln.fin ? ln.r_len : 0xffffu;
				ext_cnt[0] += ext_first && ext_kind;          // first EXT trip of a lucky check
				ext_cnt[1] += ext_first && !ext_kind;         // ... of a candidate
				ext_cnt[2] += !ext_first;                     // a later one
				ext_cnt[3] += ext_first && over && ln.fin && len < 32u;
				ext_cnt[4] += ext_first && over && ln.fin && len < 48u;
				ext_cnt[5] += ext_first && over && !ln.fin;   // went on to SEARCH / SLOW
			}
#endif
		} else if (ph == LP_SCAN) {
			lean_scan(ln, R, U4{d[0], d[1], d[2], d[3]}, U4{d[4], d[5], d[6], d[7]}, U4{d[8], d[9], d[10], d[11]},
					  U4{d[12], d[13], d[14], d[15]});
		} else if (ph == LP_REFILL) {
			ln.wb = ln.q >> 4;
			ln.we = ln.wb + 16;
#pragma unroll
			for (int i = 0; i < 16; i++) ring[(ln.wb + (uint32_t)i) & 15u][tid] = d[i];
			ln.ph = LP_STEP;
		}
		if constexpr (MODE == 1) {
			if (look_leader >= 0) {
				// the helpers' answers: a plain step (finished on the packed path, no anchor) and where it leads
				const bool plain = ph == LP_LOOK && ln.fin && !ln.r_accepted && ln.ph == LP_STEP;
				const uint32_t nxt = ln.q;
				if (ph == LP_LOOK) { // a lane without a bridge again
					ln.fin = false;
					ln.ph = LP_STEP;
				}
				const bool lead_plain = bcast((active && ln.fin && !ln.r_accepted && ln.ph == LP_STEP) ? 1u : 0u, look_leader) != 0;
				if (lead_plain) {
					uint32_t p = bcast(ln.q, look_leader); // where the walker's own step has taken it
#ifdef PHY_LEAN_TIMING
					uint32_t look_hops = 0;
#endif
					const uint32_t v0 = bcast(d[8], look_leader), v1 = bcast(d[9], look_leader), v2 = bcast(d[10], look_leader), v3 = bcast(d[11], look_leader);
					const uint32_t qw0 = bcast(ln.qw0, look_leader);
					for (;;) {
						const uint32_t k = p - look_q0;
						const uint64_t at = __ballot(ph == LP_LOOK && look_k == k);
						if (!at) break; // beyond what was looked at: the walker's next trip
						// a landing the chunk's own chain visited too may be where the two merge: LeanBridge::begin_step decides
						// that (with that chain's log) — the walker stops here
						const uint32_t wr = ((qw0 >> 1) + (p >> 5)) - look_vis0;
						const uint32_t vw = wr == 0u ? v0 : wr == 1u ? v1 : wr == 2u ? v2 : v3;
						if (wr > 3u || ((vw >> (p & 31u)) & 1u)) break;
						const int who = __ffsll((unsigned long long)at) - 1;
						if (!bcast(plain ? 1u : 0u, who)) break;
						p = bcast(nxt, who);
#ifdef PHY_LEAN_TIMING
						look_hops++;
#endif
					}
					if ((int)lane64() == look_leader) ln.q = p;
#ifdef PHY_LEAN_TIMING
					if (X.dbg && lane64() == 0) {
						atomicAdd(&X.dbg[16 + 4 * 8192 + 64 + 10], 1ull);
						atomicAdd(&X.dbg[16 + 4 * 8192 + 64 + 11], (unsigned long long)look_hops);
					}
#endif
				}
#ifdef PHY_LEAN_TIMING
				if (X.dbg && lane64() == 0) {
					atomicAdd(&X.dbg[16 + 4 * 8192 + 64 + 12], 1ull);
					atomicAdd(&X.dbg[16 + 4 * 8192 + 64 + 13], (unsigned long long)__popcll(__ballot(ph == LP_LOOK)));
					atomicAdd(&X.dbg[16 + 4 * 8192 + 64 + 14], (unsigned long long)__popcll(__ballot(plain)));
				}
#endif
			}
		}
		LEAN_TICK(3)
		// what the packed path could not answer: the wavefront resolves it, one lane at a time
		uint64_t slow = __ballot(active && (ln.ph == LP_SLOW || ln.ph == LP_SLOWEXT));
		while (slow) {
			const int leader = __ffsll((unsigned long long)slow) - 1;
			const uint32_t lph = bcast(ln.ph, leader);
#ifdef PHY_LEAN_TIMING
			tm[5] += lph == LP_SLOW ? 1ull : (1ull << 32);
#endif
			if (lph == LP_SLOW) coop_resolve(ln, leader, A, R, X, s_end);
			else coop_ext(ln, leader, A, R, s_end);
			slow &= slow - 1;
		}
	}
#ifdef PHY_LEAN_TIMING
	if (X.dbg && lane64() == 0) {
		for (int i = 0; i < 6; i++) atomicAdd(&X.dbg[MODE * 8 + i], tm[i]);
		atomicAdd(&X.dbg[MODE * 8 + 6], 1ull);
		for (int i = 0; i < 8; i++) {
			atomicAdd(&X.dbg[16 + 4 * 8192 + MODE * 16 + i], phase_trips[i]);
			atomicAdd(&X.dbg[16 + 4 * 8192 + MODE * 16 + 8 + i], phase_lanes[i]);
		}
		atomicAdd(&X.dbg[16 + 4 * 8192 + 32 + MODE], (unsigned long long)trip);
		}
		for (int i = 0; i < 6; i++) { // (per lane)
			unsigned long long v = ext_cnt[i];
			for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
			if (lane64() == 0) atomicAdd(&X.dbg[16 + 4 * 8192 + 34 + MODE * 6 + i], v);
		}
		if (X.dbg && lane64() == 0) {
		{ // per wavefront: start, end (100 MHz counter), trips, the first chunk's query (MODE 1: behind MODE 0's 4096 records)
			unsigned long long *w = X.dbg + 16 + 4 * (size_t)(MODE * 4096 + (blockIdx.x * 4 + (tid >> 6)) % 4096);
			w[0] = t_wave0;
			w[1] = __builtin_amdgcn_s_memrealtime();
			w[2] = trip;
			w[3] = first_query;
		}
	}
#endif
}

// ───────────────────────── packed tables ─────────────────────────

// 16 bytes -> one dword of 2-bit codes (first byte in bits 31..30); non-ACGT -> 0
__global__ __launch_bounds__(256) void pack2_kernel(const uint8_t *__restrict__ src, uint64_t words, uint32_t *__restrict__ dst)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= words) return;
	const U4 v = load16(src + 16 * w);
	uint32_t c = 0;
#pragma unroll
	for (uint32_t i = 0; i < 16; i++) {
		const uint32_t code = nuc_code((uint8_t)byte_at(v, i));
		c |= (code < 4u ? code : 0u) << (30u - 2u * i);
	}
	dst[w] = c;
}

// The other direction, for genomes that arrive packed (phylo_set_genomes_packed): the byte
// arena from Q2.  One thread per 16-byte piece of the arena; bytes outside every genome are
// zero (the padding the chain kernels rely on).  off[] ascends; a block looks up the genome
// of its first piece once (block-uniform → scalar loads) and a thread walks on from there.
__global__ __launch_bounds__(256) void unpack2_kernel(uint32_t *__restrict__ q2, const uint64_t *__restrict__ off,
													   const uint32_t *__restrict__ len, uint32_t n, uint64_t words,
													   uint8_t *__restrict__ dst)
{
	const uint64_t wb = (uint64_t)blockIdx.x * blockDim.x;
	uint32_t lo = 0, hi = n; // last genome starting at or before the block's first byte (n: none → genome 0 is ahead)
	while (lo < hi) {
		const uint32_t mid = lo + ((hi - lo) >> 1);
		if (off[mid] <= 16 * wb) lo = mid + 1;
		else hi = mid;
	}
	uint32_t j = lo ? lo - 1 : 0;
	const uint64_t w = wb + threadIdx.x;
	if (w >= words) return;
	const uint64_t p = 16 * w;
	while (j + 1 < n && off[j + 1] <= p) j++;
	uint32_t o[4] = {0, 0, 0, 0};
	const uint32_t c = q2[w];
	uint32_t keep = 0; // the code bits that belong to a genome: everything else in Q2 is to be 0 whatever the caller sent
	if (n && p >= off[j] && p < off[j] + len[j]) {
		const uint64_t left = off[j] + len[j] - p;
		keep = left >= 16 ? 0xffffffffu : ~(0xffffffffu >> (2u * (uint32_t)left));
#pragma unroll
		for (uint32_t i = 0; i < 16; i++) {
			const uint32_t code = (c >> (30u - 2u * i)) & 3u;
			const uint32_t b = i < left ? (0x54474341u >> (8u * code)) & 0xffu : 0u;
			o[i >> 2] |= b << (8u * (i & 3u));
		}
	}
	if (c & ~keep) q2[w] = c & keep;
	*(uint4 *)(dst + p) = uint4{o[0], o[1], o[2], o[3]};
}

// '!' at the listed positions: entry t of genome j (bad_off[j] <= t < bad_off[j+1])
__global__ __launch_bounds__(256) void patch_bad_kernel(const uint32_t *__restrict__ bad, const uint32_t *__restrict__ bad_off,
														 uint32_t n, const uint64_t *__restrict__ off, uint8_t *__restrict__ dst,
														 uint32_t *__restrict__ q2)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= bad_off[n]) return;
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		const uint32_t mid = lo + ((hi - lo) >> 1);
		if (bad_off[mid + 1] <= t) lo = mid + 1;
		else hi = mid;
	}
	const uint64_t p = off[lo] + bad[t];
	dst[p] = (uint8_t)'!';
	atomicAnd(&q2[p >> 4], ~(3u << (30u - 2u * (uint32_t)(p & 15u)))); // a separator's code is 0 (the packed kernels rely on it)
}

// Non-ACGT positions of sequences (genomes, or S as one sequence), sorted per sequence.
// A block takes one segment of BAD_SEG bytes of one sequence.  Pass 1 (out == nullptr) counts
// per segment; pass 2 writes the segment's positions, in order, from seg_off[segment].
static const uint32_t BAD_SEG = 1u << 20;
__global__ __launch_bounds__(256) void bad_positions_kernel(const uint8_t *__restrict__ base, const uint64_t *__restrict__ off,
															 const uint32_t *__restrict__ len, const uint32_t *__restrict__ seg_seq,
															 const uint32_t *__restrict__ seg_first, uint32_t *__restrict__ seg_cnt,
															 const uint32_t *__restrict__ seg_off, uint32_t *__restrict__ out)
{
	__shared__ uint32_t wsum[4];
	__shared__ uint32_t carry;
	const uint32_t sg = blockIdx.x, j = seg_seq[sg];
	const uint32_t p0 = (sg - seg_first[j]) * BAD_SEG;
	const uint32_t n = len[j], p1 = p0 + BAD_SEG < n ? p0 + BAD_SEG : n;
	const uint8_t *s = base + off[j];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) carry = out ? seg_off[sg] : 0u;
	__syncthreads();
	for (uint32_t t0 = p0; t0 < p1; t0 += 4096u) {
		const uint32_t p = t0 + threadIdx.x * 16u;
		uint32_t mask = 0;
		if (p < p1) {
			const U4 v = load16(s + p);
			uint32_t b0, b1, b2, b3;
			(void)code4(v.x, &b0);
			(void)code4(v.y, &b1);
			(void)code4(v.z, &b2);
			(void)code4(v.w, &b3);
			// 0x80 per bad byte -> one bit per byte
			mask = ((b0 >> 7) * 0x00204081u >> 21 & 0xfu) | (((b1 >> 7) * 0x00204081u >> 21 & 0xfu) << 4) |
				   (((b2 >> 7) * 0x00204081u >> 21 & 0xfu) << 8) | (((b3 >> 7) * 0x00204081u >> 21 & 0xfu) << 12);
			if (p + 16 > p1) mask &= (1u << (p1 - p)) - 1u;
		}
		if (!__syncthreads_or(mask != 0)) continue;
		const uint32_t c = (uint32_t)__popc(mask);
		uint32_t incl = c;
#pragma unroll
		for (int dd = 1; dd < 64; dd <<= 1) {
			const uint32_t t = (uint32_t)__shfl_up((int)incl, dd, 64);
			if ((int)lane >= dd) incl += t;
		}
		if (lane == 63) wsum[wave] = incl;
		__syncthreads();
		uint32_t o = carry + incl - c;
		for (uint32_t w2 = 0; w2 < wave; w2++) o += wsum[w2];
		if (out) {
			uint32_t mm = mask;
			while (mm) {
				const uint32_t b = (uint32_t)__ffs((int)mm) - 1u;
				out[o++] = p + b;
				mm &= mm - 1u;
			}
		}
		__syncthreads();
		if (threadIdx.x == 255) carry = o + (out ? 0u : c); // pass 2: o already advanced past this thread's positions
		__syncthreads();
	}
	if (!out && threadIdx.x == 0) seg_cnt[sg] = carry;
}

// ───────────────────────── launch wrappers ─────────────────────────

void launch_pack2(const uint8_t *src, uint64_t bytes, uint32_t *dst, hipStream_t st)
{
	const uint64_t words = bytes / 16;
	if (!words) return;
	hipLaunchKernelGGL(pack2_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, st, src, words, dst);
}
void launch_unpack2(uint32_t *q2, const uint64_t *off, const uint32_t *len, uint32_t n, uint64_t bytes, uint8_t *dst,
					const uint32_t *bad, const uint32_t *bad_off, uint32_t nbad, hipStream_t st)
{
	const uint64_t words = bytes / 16;
	if (words) hipLaunchKernelGGL(unpack2_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, st, q2, off, len, n, words, dst);
	if (nbad) hipLaunchKernelGGL(patch_bad_kernel, dim3((nbad + 255) / 256), dim3(256), 0, st, bad, bad_off, n, off, dst, q2);
}
uint32_t bad_segment_bytes() { return BAD_SEG; }
void launch_bad_positions(const uint8_t *base, const uint64_t *off, const uint32_t *len, const uint32_t *seg_seq,
						  const uint32_t *seg_first, uint32_t nseg, uint32_t *seg_cnt, const uint32_t *seg_off, uint32_t *out,
						  hipStream_t st)
{
	if (!nseg) return;
	hipLaunchKernelGGL(bad_positions_kernel, dim3(nseg), dim3(256), 0, st, base, off, len, seg_seq, seg_first, seg_cnt, seg_off, out);
}

void launch_lean_overruns(const PhaseA &A, const RefIndex &R, uint32_t nq, hipStream_t st)
{
	if (!nq || !A.nchunks) return;
	hipLaunchKernelGGL(lean_overrun_direct_kernel, dim3((A.nchunks + 63u) / 64u), dim3(64), 0, st, A, R, 0u, A.nchunks);
	hipLaunchKernelGGL(lean_overrun_chain_kernel, dim3(nq), dim3(64), 0, st, A, 0u, nq);
}

static int lean_resident(const void *fn, int n_cu)
{
	int per_cu = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
	return per_cu * n_cu;
}
int lean_spec_resident_blocks(int n_cu)
{
	// (n_cu, blocks) in one atomic word: contexts driven from different host threads may ask at the same time
	static std::atomic<uint64_t> cache{0};
	const uint64_t seen = cache.load(std::memory_order_acquire);
	if ((int)(seen >> 32) == n_cu && (uint32_t)seen) return (int)(uint32_t)seen;
	const int blocks = lean_resident((const void *)lean_chain_kernel<0>, n_cu);
	cache.store(((uint64_t)(uint32_t)n_cu << 32) | (uint32_t)blocks, std::memory_order_release);
	return blocks;
}
void launch_lean_spec(const PhaseA &A, const RefIndex &R, const LeanIndex &X, int n_cu, hipStream_t st, int max_blocks)
{
	int blocks = lean_spec_resident_blocks(n_cu);
	if (max_blocks > 0 && max_blocks < blocks) blocks = max_blocks;
	const int need = (int)((A.nchunks + 255) / 256);
	if (need < blocks) blocks = need > 0 ? need : 1;
	hipLaunchKernelGGL(lean_chain_kernel<0>, dim3(blocks), dim3(256), 0, st, A, R, X);
}
// Blocks for `count` chunks' bridges (about half of them need walking: lean_bridge_prepare_kernel settles the rest).  Most
// walks end within a few steps and a few run for dozens, so with a lane per bridge nearly every wavefront is soon left
// with a handful of walkers and runs whole trips for them: fewer lanes, each taking bridge after bridge from the counter,
// keep the wavefronts filled until the queue is empty and leave only the last walkers' tail.  How few is a matter of what
// a trip costs — with the STEP-only trips (FAST_TRIPS) a block per CU, a third of a lane per chunk, at most three
// blocks on two CUs (C3, 742 blocks' worth of chunks: 0.41 ms at 96 blocks, 0.36 at 128, 0.325 at 192, 0.319 at 256,
// 0.343 at 384; C4, 2230: 0.81 at 128, 0.66 at 192, 0.59 at 256, 0.56 at 384) — and twice the CUs with k = 14's
// slot table, whose fetches want more of them in flight.
static int lean_bridge_blocks(uint32_t count, int n_cu, uint32_t k)
{
	int blocks = lean_resident((const void *)lean_chain_kernel<1>, n_cu);
	const int need = (int)((count + 255) / 256);
	if (need < blocks) blocks = need > 0 ? need : 1;
	const int few = std::min(std::max(n_cu, need / 3), k >= 14u ? 2 * n_cu : 3 * n_cu / 2);
	if (few < blocks) blocks = few;
#ifdef PHY_DEV_HOOKS
	if (const char *e = getenv("PHY_BRIDGE_BLOCKS")) blocks = std::max(1, std::min(std::max(need, 1), atoi(e))); // experiments
#endif
	return blocks;
}
void launch_lean_bridge(const PhaseA &A, const RefIndex &R, const LeanIndex &X, int n_cu, hipStream_t st, const BridgeZero &Z)
{
	hipLaunchKernelGGL(lean_bridge_prepare_kernel, dim3((A.nchunks + 255u) / 256u), dim3(256), 0, st, A, R, X, Z);
	hipLaunchKernelGGL(lean_chain_kernel<1>, dim3(lean_bridge_blocks(A.nchunks, n_cu, R.k)), dim3(256), 0, st, A, R, X);
}

} // namespace phy

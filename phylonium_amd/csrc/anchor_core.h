// anchor_core.h — the anchor chain (phase A) as a per-lane state machine.
//
// Replaces, with identical results:
//   esa::get_match_cached / get_match / get_match_from / get_interval
//       (/root/reference/src/esa.cxx:361-563)
//   lcp                         (src/process.cxx:171-184)
//   the lucky_anchor / anchor lambdas and the position chain of
//   anchor_homologies           (src/process.cxx:198-282)
//
// The reference walks a child-table ESA (SA+LCP+CLD+FVC, 26 B/entry, ~25
// dependent reads per match).  What the chain needs from a match is only
// (length of the longest prefix of the query suffix that occurs in S, whether it
// occurs exactly once, where) — SURVEY §3.3.  Here that is answered by a k-mer
// bucket table T over the suffix array plus a binary search inside the bucket:
// the longest match is max(lcp(query, pred), lcp(query, succ)) at the query's
// insertion point, and it is unique iff exactly one neighbour attains it and the
// LCP array says the next suffix outward does not share it.  Typically 1-2
// suffix comparisons and 4-5 dependent reads per match.
//
// Everything here is plain C++ over raw pointers so that the same code is
// compiled by hipcc for gfx950 (kernels in anchor_kernels.hip) and by g++ for
// the CPU emulation harness under tests/emul (test infrastructure; the product
// never runs it).
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define PHY_HD __host__ __device__ __forceinline__
#else
#define PHY_HD inline
#endif

namespace phy {

struct RefIndex {
	const uint8_t *S;    // n bytes of subject + '#' + revcomp, then >= 64 zero bytes
	const uint32_t *SA;  // n entries
	const uint32_t *LCP; // n+1 entries; LCP[r] = lcp(suffix SA[r-1], suffix SA[r]); LCP[0]=LCP[n]=0
	const uint32_t *T;   // 4^k+1 entries; T[c] = #suffixes lexicographically < k-mer c
	uint32_t n;          // |S| = 2L+1
	uint32_t k;          // bucket k-mer length (1..14)
	uint32_t threshold;  // minimum anchor length
};

struct Anchor {
	uint32_t q, s, len; // this_pos_Q, this_pos_S, this_length of an accepted anchor
};

PHY_HD uint64_t load8(const uint8_t *p)
{
	uint64_t v;
	__builtin_memcpy(&v, p, 8);
	return v;
}

PHY_HD uint32_t ctz64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__ffsll((unsigned long long)x) - 1u;
#else
	return (uint32_t)__builtin_ctzll(x);
#endif
}

// A<C<G<T → 0..3; returns 4 for anything else.
PHY_HD uint32_t nuc_code(uint8_t b)
{
	uint32_t v = ((b >> 1) & 3u) ^ ((b >> 2) & 1u);
	return (((0x54474341u >> (8 * v)) & 0xffu) == b) ? v : 4u;
}

PHY_HD bool kmer_code(const uint8_t *q, uint32_t k, uint32_t *code)
{
	uint32_t c = 0;
	for (uint32_t i = 0; i < k; i++) {
		uint32_t v = nuc_code(q[i]);
		if (v > 3) return false;
		c = (c << 2) | v;
	}
	*code = c;
	return true;
}

struct CmpReq {
	const uint8_t *qp; // query suffix
	const uint8_t *sp; // subject suffix
	uint32_t from;     // bytes already known equal
	uint32_t maxn;     // remaining query length (compare stops there)
};

struct CmpRes {
	uint32_t len; // common prefix length, <= maxn
	bool s_less;  // subject byte < query byte at the first difference (false when len==maxn)
};

// Compare up to `budget` bytes past `from`. Returns true when decided.
// The query is followed by zero padding and S by zero padding, and query bytes
// inside [0,maxn) are never zero, so running past the end of S stops the scan
// by itself (that is the NUL the reference's loops stop at, esa.cxx:461,
// process.cxx:179).
PHY_HD bool cmp_some(const CmpReq &r, uint32_t budget, uint32_t *pos, CmpRes *out)
{
	uint32_t i = *pos;
	uint32_t stop = (r.maxn - i > budget) ? i + budget : r.maxn;
	while (i < stop) {
		uint64_t a = load8(r.qp + i), b = load8(r.sp + i);
		uint64_t x = a ^ b;
		if (x) {
			uint32_t byte = ctz64(x) >> 3;
			i += byte;
			if (i >= r.maxn) {
				out->len = r.maxn;
				out->s_less = false;
			} else {
				out->len = i;
				out->s_less = (uint8_t)(b >> (8 * byte)) < (uint8_t)(a >> (8 * byte));
			}
			return true;
		}
		i += 8;
	}
	if (i >= r.maxn) {
		out->len = r.maxn;
		out->s_less = false;
		return true;
	}
	*pos = i;
	return false;
}

enum ChainState : uint32_t {
	ST_STEP = 0, // at a fresh query position
	ST_LUCKY_R,
	ST_BS,
	ST_BS_R,
	ST_NBP,
	ST_NBP_R,
	ST_NBS,
	ST_NBS_R,
	ST_FIN
};

enum AdvanceResult : uint32_t { ADV_NEED_CMP = 0, ADV_STEP_DONE = 1 };

// One chain = the loop of anchor_homologies (process.cxx:245-282) without the
// homology bookkeeping (that is a fold over the accepted anchors, done later).
struct Chain {
	const uint8_t *Q; // query bytes
	uint32_t qlen;
	uint32_t q;          // this_pos_Q
	uint32_t lq, ls, ll; // last_pos_Q, last_pos_S, last_length
	uint32_t st;

	// search state
	uint32_t lo, hi, mid;
	uint32_t l_lo, l_hi; // lcp(query, suffix lo-1) / lcp(query, suffix hi) when known
	uint32_t p_lo, p_hi; // their positions in S
	uint32_t flags;      // bit0 lo_known, bit1 hi_known, bit2 k-mer bucket valid
	uint32_t p;          // subject position of the pending comparison
	uint32_t lp, pp, lsu, psu;

	// result of the step that just finished (valid after ADV_STEP_DONE)
	uint32_t r_q, r_s, r_len;
	bool r_accepted;

	PHY_HD void reset(const uint8_t *query, uint32_t query_len, uint32_t q0, uint32_t a_q, uint32_t a_s,
					  uint32_t a_len)
	{
		Q = query;
		qlen = query_len;
		q = q0;
		lq = a_q;
		ls = a_s;
		ll = a_len;
		st = ST_STEP;
	}

	PHY_HD void finish_step(uint32_t pos, uint32_t len, bool accepted)
	{
		r_q = q;
		r_s = pos;
		r_len = len;
		r_accepted = accepted;
		if (accepted) { // process.cxx:275-277
			lq = q;
			ls = pos;
			ll = len;
		}
		q += len + 1; // process.cxx:281
		st = ST_STEP;
	}

	// Runs the machine until it needs a suffix comparison (fills req) or the
	// current step is complete. `res` is the answer to the previous request.
	PHY_HD AdvanceResult advance(const RefIndex &R, const CmpRes &res, CmpReq *req)
	{
		for (;;) {
			switch (st) {
				case ST_STEP: {
					// lucky_anchor, process.cxx:227-242
					uint32_t advance = q - lq;
					uint32_t gap = advance - ll;
					uint32_t try_s = ls + advance;
					if (try_s < R.n && gap <= R.threshold) {
						p = try_s;
						req->qp = Q + q;
						req->sp = R.S + try_s;
						req->from = 0;
						req->maxn = qlen - q;
						st = ST_LUCKY_R;
						return ADV_NEED_CMP;
					}
					begin_search(R);
					break;
				}
				case ST_LUCKY_R: {
					if (res.len >= R.threshold) {
						finish_step(p, res.len, true);
						return ADV_STEP_DONE;
					}
					begin_search(R);
					break;
				}
				case ST_BS: {
					if (lo < hi) {
						mid = lo + ((hi - lo) >> 1);
						p = R.SA[mid];
						uint32_t a = (flags & 1u) ? l_lo : 0u, b = (flags & 2u) ? l_hi : 0u;
						req->qp = Q + q;
						req->sp = R.S + p;
						req->from = a < b ? a : b;
						req->maxn = qlen - q;
						st = ST_BS_R;
						return ADV_NEED_CMP;
					}
					st = ST_NBP;
					break;
				}
				case ST_BS_R: {
					if (res.s_less) {
						lo = mid + 1;
						l_lo = res.len;
						p_lo = p;
						flags |= 1u;
					} else {
						hi = mid;
						l_hi = res.len;
						p_hi = p;
						flags |= 2u;
					}
					st = ST_BS;
					break;
				}
				case ST_NBP: {
					// predecessor of the insertion point
					if (flags & 1u) {
						lp = l_lo;
						pp = p_lo;
					} else if (lo == 0 || ((flags & 6u) == 6u && l_hi >= R.k)) {
						lp = 0; // none, or outside a bucket whose member already shares >= k
						pp = 0;
					} else {
						p = R.SA[lo - 1];
						req->qp = Q + q;
						req->sp = R.S + p;
						req->from = 0;
						req->maxn = qlen - q;
						st = ST_NBP_R;
						return ADV_NEED_CMP;
					}
					st = ST_NBS;
					break;
				}
				case ST_NBP_R: {
					lp = res.len;
					pp = p;
					st = ST_NBS;
					break;
				}
				case ST_NBS: {
					if (flags & 2u) {
						lsu = l_hi;
						psu = p_hi;
					} else if (hi >= R.n || ((flags & 5u) == 5u && l_lo >= R.k)) {
						lsu = 0;
						psu = 0;
					} else {
						p = R.SA[hi];
						req->qp = Q + q;
						req->sp = R.S + p;
						req->from = 0;
						req->maxn = qlen - q;
						st = ST_NBS_R;
						return ADV_NEED_CMP;
					}
					st = ST_FIN;
					break;
				}
				case ST_NBS_R: {
					lsu = res.len;
					psu = p;
					st = ST_FIN;
					break;
				}
				default: { // ST_FIN — anchor(), process.cxx:219-225
					uint32_t lmax, pos;
					bool uniq;
					if (lp > lsu) {
						lmax = lp;
						pos = pp;
						// rank lo-1 is the best; unique iff rank lo-2 does not share lmax
						uniq = lmax >= R.threshold && R.LCP[lo - 1] < lmax;
					} else if (lsu > lp) {
						lmax = lsu;
						pos = psu;
						uniq = lmax >= R.threshold && R.LCP[hi + 1] < lmax;
					} else {
						lmax = lp;
						pos = 0;
						uniq = false;
					}
					finish_step(pos, lmax, uniq && lmax >= R.threshold);
					return ADV_STEP_DONE;
				}
			}
		}
	}

  private:
	PHY_HD void begin_search(const RefIndex &R)
	{
		uint32_t code;
		flags = 0;
		l_lo = l_hi = 0;
		if (qlen - q >= R.k && kmer_code(Q + q, R.k, &code)) {
			lo = R.T[code];
			hi = R.T[code + 1];
			flags = 4u;
		} else {
			lo = 0;
			hi = R.n;
		}
		st = ST_BS;
	}
};

// ───────────────────────── phase-A work layout ─────────────────────────
//
// Every query is cut into chunks of C positions (C a power of two).  Chunk c of
// query j has global id qchunk0[j] + c.  A *speculative* chain starts at every
// chunk boundary in the state the reference has at q = 0 (last_* = 0, which is
// lucky-ineligible for q > threshold) and runs to the end of its chunk,
// logging its accepted anchors, a visited bit per position, and its exit
// state.  A *bridge* then continues each chunk's exit state into the following
// chunk(s) until it stands at a position the speculative chain there also
// visited in an equivalent state; from there on the two are the same chain.
// The true chain of a query is chunk 0's speculative log, its bridge, the
// target chunk's log from the merge index, its bridge, …

struct SpecExit {
	uint32_t q, lq, ls, ll;
};

static const uint32_t BRIDGE_INLINE = 4;    // anchors stored inside the bridge record
static const uint32_t POOL_BLOCK = 14;      // anchors per overflow block
static const uint32_t BRIDGE_END = 0xffffffffu;
static const uint32_t NO_BLOCK = 0xffffffffu;

struct BridgeRec {
	uint32_t target;  // global chunk id merged into, or BRIDGE_END
	uint32_t idx_m;   // first anchor of target's log that belongs to the true chain
	uint32_t n;       // anchors accepted by the bridge
	uint32_t block;   // first overflow block or NO_BLOCK
	Anchor a[BRIDGE_INLINE];
};

struct PoolBlock {
	Anchor a[POOL_BLOCK];
	uint32_t next;
	uint32_t pad;
};

struct PhaseA {
	// inputs
	const uint8_t *qbase;      // all genomes, each followed by >= 64 zero bytes
	const uint64_t *qoff;      // [nq] byte offset of genome j in qbase
	const uint32_t *qlen;      // [nq]
	const uint32_t *qchunk0;   // [nq+1] first global chunk id of each query
	const uint32_t *items;     // [nchunks] work order: global chunk ids, round-robin over queries
	const uint32_t *chunk_query; // [nchunks] query id of each global chunk
	uint32_t nchunks;
	uint32_t C;                // chunk length (power of two)
	uint32_t logC;
	uint32_t cap;              // anchor slots per chunk
	// speculative logs
	Anchor *spec_anchors;      // [nchunks*cap]
	uint32_t *spec_cnt;        // [nchunks]
	SpecExit *spec_exit;       // [nchunks]
	uint32_t *visited;         // [nchunks*C/32] bitmap
	// bridges
	BridgeRec *bridge;         // [nchunks]
	PoolBlock *pool;
	uint32_t pool_blocks;
	uint32_t *pool_next;       // bump allocator
	uint32_t *error;           // set nonzero on pool exhaustion / capacity overflow
	// counters for dynamic work fetch
	uint32_t *fetch;           // [2]
};

PHY_HD bool lucky_eligible(uint32_t q, uint32_t aq, uint32_t as, uint32_t al, const RefIndex &R)
{
	uint32_t advance = q - aq;
	return (as + advance < R.n) && (advance - al <= R.threshold);
}

// ── speculative chain driver (one lane) ──
struct SpecLane {
	Chain ch;
	uint32_t gc;        // global chunk id; BRIDGE_END when out of work
	uint32_t q_end;     // chunk end (clipped to query length)
	uint32_t cnt;       // anchors logged
	uint32_t vis_word;  // visited bits being accumulated
	uint32_t vis_idx;   // word index (global) of vis_word, or NO_BLOCK

	PHY_HD void start(const PhaseA &A, uint32_t chunk)
	{
		gc = chunk;
		uint32_t j = A.chunk_query[chunk];
		uint32_t c = chunk - A.qchunk0[j];
		uint32_t q0 = c << A.logC;
		uint32_t ql = A.qlen[j];
		uint32_t e = q0 + A.C;
		q_end = e < ql ? e : ql;
		ch.reset(A.qbase + A.qoff[j], ql, q0, 0, 0, 0);
		cnt = 0;
		vis_word = 0;
		vis_idx = NO_BLOCK;
	}

	// Called when ch.st == ST_STEP. Returns false when the chunk is finished.
	PHY_HD bool begin_step(const PhaseA &A)
	{
		if (ch.q >= q_end) {
			if (vis_idx != NO_BLOCK) A.visited[vis_idx] = vis_word;
			A.spec_cnt[gc] = cnt;
			SpecExit x = {ch.q, ch.lq, ch.ls, ch.ll};
			A.spec_exit[gc] = x;
			return false;
		}
		uint32_t local = ch.q & (A.C - 1);
		uint32_t w = gc * (A.C >> 5) + (local >> 5);
		if (w != vis_idx) {
			if (vis_idx != NO_BLOCK) A.visited[vis_idx] = vis_word;
			vis_idx = w;
			vis_word = 0;
		}
		vis_word |= 1u << (local & 31);
		return true;
	}

	PHY_HD void step_done(const PhaseA &A)
	{
		if (ch.r_accepted) {
			if (cnt < A.cap) {
				Anchor a = {ch.r_q, ch.r_s, ch.r_len};
				A.spec_anchors[(size_t)gc * A.cap + cnt] = a;
			} else {
				*A.error = 1;
			}
			cnt++;
		}
	}
};

// ── bridge driver (one lane) ──
struct BridgeLane {
	Chain ch;
	uint32_t src;      // chunk whose exit state is being continued; BRIDGE_END when idle
	uint32_t qc0;      // first global chunk of the query
	uint32_t cur_gc;   // chunk of the speculative log being compared against
	uint32_t sp_cnt, sp_idx;
	Anchor Ls;         // last anchor the speculative chain had accepted before ch.q
	uint32_t n;        // anchors accepted by this bridge
	uint32_t first_block, cur_block;

	PHY_HD void start(const PhaseA &A, uint32_t chunk)
	{
		src = chunk;
		uint32_t j = A.chunk_query[chunk];
		qc0 = A.qchunk0[j];
		SpecExit x = A.spec_exit[chunk];
		ch.reset(A.qbase + A.qoff[j], A.qlen[j], x.q, x.lq, x.ls, x.ll);
		cur_gc = BRIDGE_END;
		sp_cnt = sp_idx = 0;
		Ls.q = Ls.s = Ls.len = 0;
		n = 0;
		first_block = cur_block = NO_BLOCK;
	}

	PHY_HD void finish(const PhaseA &A, uint32_t target, uint32_t idx_m)
	{
		BridgeRec *b = &A.bridge[src];
		b->target = target;
		b->idx_m = idx_m;
		b->n = n;
		b->block = first_block;
	}

	// Returns false when the bridge has merged or reached the end of the query.
	PHY_HD bool begin_step(const PhaseA &A, const RefIndex &R)
	{
		if (ch.q >= ch.qlen) {
			finish(A, BRIDGE_END, 0);
			return false;
		}
		uint32_t gc = qc0 + (ch.q >> A.logC);
		if (gc != cur_gc) {
			cur_gc = gc;
			sp_cnt = A.spec_cnt[gc];
			sp_idx = 0;
			Ls.q = Ls.s = Ls.len = 0;
		}
		const Anchor *log = A.spec_anchors + (size_t)gc * A.cap;
		while (sp_idx < sp_cnt && log[sp_idx].q < ch.q) {
			Ls = log[sp_idx];
			sp_idx++;
		}
		uint32_t local = ch.q & (A.C - 1);
		uint32_t w = A.visited[gc * (A.C >> 5) + (local >> 5)];
		if ((w >> (local & 31)) & 1u) {
			bool eb = lucky_eligible(ch.q, ch.lq, ch.ls, ch.ll, R);
			bool es = lucky_eligible(ch.q, Ls.q, Ls.s, Ls.len, R);
			bool merged = false;
			if (!eb && !es) merged = true;
			else if (eb && es && (ch.ls - ch.lq == Ls.s - Ls.q) && (ch.lq + ch.ll == Ls.q + Ls.len))
				merged = true;
			if (merged) {
				finish(A, gc, sp_idx);
				return false;
			}
		}
		return true;
	}

	// alloc: bump allocator for overflow blocks; returns NO_BLOCK on exhaustion
	template <class Alloc> PHY_HD void step_done(const PhaseA &A, Alloc alloc)
	{
		if (!ch.r_accepted) return;
		Anchor a = {ch.r_q, ch.r_s, ch.r_len};
		if (n < BRIDGE_INLINE) {
			A.bridge[src].a[n] = a;
		} else {
			uint32_t k = (n - BRIDGE_INLINE) % POOL_BLOCK;
			if (k == 0) {
				uint32_t nb = alloc();
				if (nb == NO_BLOCK) {
					*A.error = 2;
					n++;
					return;
				}
				A.pool[nb].next = NO_BLOCK;
				if (cur_block == NO_BLOCK) first_block = nb;
				else A.pool[cur_block].next = nb;
				cur_block = nb;
			}
			if (cur_block != NO_BLOCK) A.pool[cur_block].a[k] = a;
		}
		n++;
	}
};

// ───────────────────────── fold: anchors → homologies ─────────────────────────
//
// process.cxx:246-292 as a fold over the accepted anchors.  `cur` is the
// homology being grown; it starts as the reference's `homology(0,0)`.

struct RawHom {
	uint32_t iref, iq, len; // index_reference, index_query, length (before reverseEh)
};

struct FoldState {
	Anchor last;      // last_pos_Q/S, last_length
	bool last_right;  // last_was_right_anchor
	uint32_t cur_s, cur_q, cur_len;
};

PHY_HD void fold_init(FoldState *f)
{
	f->last.q = f->last.s = f->last.len = 0;
	f->last_right = false;
	f->cur_s = f->cur_q = f->cur_len = 0;
}

PHY_HD bool is_right_anchor(const Anchor &last, const Anchor &a, uint32_t border)
{
	uint32_t end_s = last.s + last.len, end_q = last.q + last.len;
	return a.s > end_s && (a.q - end_q == a.s - end_s) && ((a.s < border) == (last.s < border));
}

// Returns true when a homology was completed (written to *out).
PHY_HD bool fold_anchor(FoldState *f, const Anchor &a, uint32_t border, uint32_t threshold, RawHom *out)
{
	bool emitted = false;
	if (is_right_anchor(f->last, a, border)) {
		uint32_t end_q = f->last.q + f->last.len;
		f->cur_len += a.q - end_q + a.len;
		f->last_right = true;
	} else {
		if (f->last_right || f->last.len / 2 >= threshold) {
			out->iref = f->cur_s;
			out->iq = f->cur_q;
			out->len = f->cur_len;
			emitted = true;
		}
		f->cur_s = a.s;
		f->cur_q = a.q;
		f->cur_len = a.len;
		f->last_right = false;
	}
	f->last = a;
	return emitted;
}

PHY_HD bool fold_finish(FoldState *f, uint32_t qlen, uint32_t threshold, RawHom *out)
{
	if (f->last.len >= qlen) { // process.cxx:285-287
		f->cur_s = f->last.s;
		f->cur_q = 0;
		f->cur_len = qlen;
	}
	if (f->last_right || f->last.len / 2 >= threshold) {
		out->iref = f->cur_s;
		out->iq = f->cur_q;
		out->len = f->cur_len;
		return true;
	}
	return false;
}

} // namespace phy

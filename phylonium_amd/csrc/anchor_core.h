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

struct U4 { // 16 bytes of sequence or four table entries
	uint32_t x, y, z, w;
};

struct RefIndex {
	const uint8_t *S;    // n bytes of subject + '#' + revcomp, then >= 64 zero bytes
	const U4 *SAX;       // n records (+4 pad), one per rank: see sax_record()
	const uint32_t *LCP; // n+1 entries (+4 pad); LCP[r] = lcp(suffix SA[r-1], suffix SA[r]); LCP[0]=LCP[n]=0
	const U4 *SLOT;      // 4^k slots of 64 bytes (SLOT_RECS U4 each): see slot_pack
	uint32_t n;          // |S| = 2L+1
	uint32_t k;          // bucket k-mer length (1..14)
	uint32_t threshold;  // minimum anchor length
};

struct Anchor {
	uint32_t q, s, len; // this_pos_Q, this_pos_S, this_length of an accepted anchor
};

PHY_HD U4 load16(const uint8_t *p)
{
	U4 v;
	__builtin_memcpy(&v, p, 16);
	return v;
}

PHY_HD uint32_t ctz32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__ffs((int)x) - 1u;
#else
	return (uint32_t)__builtin_ctz(x);
#endif
}

// Index (0..15) of the first nonzero byte of four dwords, 16 if all are zero.  Four
// independent bit scans and a minimum: a chain of ?: here compiles to nested branches,
// and this sits in every step of every chain.
PHY_HD uint32_t first_set_byte(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
	const uint32_t t0 = x0 ? ctz32(x0) : 128u;
	const uint32_t t1 = x1 ? 32u + ctz32(x1) : 128u;
	const uint32_t t2 = x2 ? 64u + ctz32(x2) : 128u;
	const uint32_t t3 = x3 ? 96u + ctz32(x3) : 128u;
	const uint32_t m01 = t0 < t1 ? t0 : t1, m23 = t2 < t3 ? t2 : t3;
	return (m01 < m23 ? m01 : m23) >> 3;
}

// A<C<G<T → 0..3; returns 4 for anything else.
PHY_HD uint32_t nuc_code(uint8_t b)
{
	uint32_t v = ((b >> 1) & 3u) ^ ((b >> 2) & 1u);
	return (((0x54474341u >> (8 * v)) & 0xffu) == b) ? v : 4u;
}

PHY_HD uint32_t byte_at(const U4 &v, uint32_t i)
{
	uint32_t w = (i < 4) ? v.x : (i < 8) ? v.y : (i < 12) ? v.z : v.w;
	return (w >> (8 * (i & 3))) & 0xffu;
}

// Four bytes → four 2-bit codes (A0 C1 G2 T3) packed first-byte-first into 8
// bits; `bad` gets 0x80 in every byte that is not one of A,C,G,T.
PHY_HD uint32_t code4(uint32_t x, uint32_t *bad)
{
	uint32_t c = ((x >> 1) & 0x03030303u) ^ ((x >> 2) & 0x01010101u);
	uint32_t b0 = c & 0x01010101u, b1 = (c >> 1) & 0x01010101u;
	uint32_t expect = 0x41414141u + 2u * b0 + 6u * b1 + 11u * (b0 & b1); // 'A','C','G','T' per byte
	uint32_t diff = expect ^ x;
	*bad = (((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) & 0x80808080u;
	return (c * 0x40100401u) >> 24;
}

// 16 bytes → 32-bit code (first byte in bits 31-30) and the number of leading
// bytes that are A,C,G,T
PHY_HD uint32_t window_code(const U4 &q, uint32_t *valid)
{
	uint32_t b0, b1, b2, b3;
	uint32_t c = (code4(q.x, &b0) << 24) | (code4(q.y, &b1) << 16) | (code4(q.z, &b2) << 8) | code4(q.w, &b3);
	*valid = first_set_byte(b0, b1, b2, b3);
	return c;
}

PHY_HD uint32_t clz32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__clz((int)x);
#else
	return (uint32_t)__builtin_clz(x);
#endif
}

// SAX record of rank r: everything a comparison with suffix SA[r] usually needs,
// in 16 bytes — x: SA[r]; y: 2-bit codes of the suffix's first 16 bytes (bits
// beyond the valid length are 0); z: number of leading A,C,G,T bytes (<= 16);
// w: min(LCP[r],LCP_CLIP) | min(LCP[r+1],LCP_CLIP) << 16.  The clip is 13 bits so
// that a record also fits the 12 bytes a slot has for it (slot_pack).
static const uint32_t LCP_CLIP = 0x1fffu;
PHY_HD U4 sax_record(const uint8_t *S, uint32_t sa, uint32_t lcp_r, uint32_t lcp_r1)
{
	U4 w = load16(S + sa);
	uint32_t valid;
	uint32_t code = window_code(w, &valid);
	if (valid < 16) code &= ~(0xffffffffu >> (2 * valid)); // valid == 0 → shift by 0 → mask all
	if (valid == 0) code = 0;
	U4 r;
	r.x = sa;
	r.y = code;
	r.z = valid;
	r.w = (lcp_r < LCP_CLIP ? lcp_r : LCP_CLIP) | ((lcp_r1 < LCP_CLIP ? lcp_r1 : LCP_CLIP) << 16);
	return r;
}

// Slot of k-mer c: 64 bytes = 16 dwords: lo = T[c], hi = T[c+1], then the SAX records of
// ranks base .. base+3 (base = lo ? lo-1 : 0: the bucket's predecessor, up to two
// members and its successor for buckets of <= 2 suffixes) at 12 bytes each — SA, prefix
// code, and valid length (5 bits) | LCP[r] (13) | LCP[r+1] (13) in one dword.  A whole
// small-bucket search is therefore ONE fetch of half a cache line.  64 rather than 128
// bytes because the table is gathered from at random and what that costs on this chip is
// set by how many pages are in play (TLB reach, ~3 GB: csrc/gather_bench.hip), not by the
// bytes: at 128 B the table alone is 2.1 GB for a 5 Mbp reference.
static const uint32_t SLOT_RECS = 4; // U4 units per slot

PHY_HD void slot_pack(uint32_t lo, uint32_t hi, const U4 rec[4], U4 out[4])
{
	uint32_t d[16];
	d[0] = lo;
	d[1] = hi;
	for (int i = 0; i < 4; i++) {
		d[2 + 3 * i] = rec[i].x;
		d[3 + 3 * i] = rec[i].y;
		d[4 + 3 * i] = (rec[i].z & 31u) | ((rec[i].w & LCP_CLIP) << 5) | (((rec[i].w >> 16) & LCP_CLIP) << 18);
	}
	d[14] = d[15] = 0;
	for (int i = 0; i < 4; i++) out[i] = U4{d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]};
}

// Compare the query window (code qcode, qv valid bytes, n bytes left in the query)
// with a suffix prefix (code pre, sv valid bytes).  0: decided (*len,*less);
// 1: all 16 bytes equal and the query goes on — extend from byte 16;
// 2: a non-ACGT byte or the end of S is involved — compare the raw bytes.
PHY_HD uint32_t packed_cmp(uint32_t qcode, uint32_t qv, uint32_t n, uint32_t pre, uint32_t sv, uint32_t *len,
						   uint32_t *less)
{
	// written with selects, not branches: on the GPU every divergent `if` costs
	// scalar exec-mask instructions, and the CU has one scalar unit for all its waves
	const uint32_t x = qcode ^ pre;
	const uint32_t d = x ? (clz32(x) >> 1) : 16u;
	const uint32_t m = qv < sv ? qv : sv;
	const uint32_t sh = 30u - 2u * (d < 15u ? d : 15u);
	const bool mism = d < m;
	const bool full = !mism && m == 16u;                        // 16 bytes equal
	const bool qend = !mism && n == m && (m == 16u || qv == m); // the query ends inside the window
	*len = mism ? d : n;
	*less = (mism && ((pre >> sh) & 3u) < ((qcode >> sh) & 3u)) ? 1u : 0u;
	return (mism || qend) ? 0u : (full ? 1u : 2u);
}

PHY_HD uint32_t sel4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t i)
{
	return i == 0 ? a0 : i == 1 ? a1 : i == 2 ? a2 : a3;
}

// index of the first differing byte of two 16-byte windows, 16 if equal
PHY_HD uint32_t first_diff(const U4 &a, const U4 &b)
{
	return first_set_byte(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
}

// The query is followed by zero padding and S by zero padding, and query bytes
// inside [0,n) are never zero, so running past the end of S ends a match by
// itself (that is the NUL the reference's loops stop at, esa.cxx:461,
// process.cxx:179); running past the end of the query is cut by `n`.

enum ChainState : uint32_t {
	ST_STEP = 0, // load the query window (and the lucky window)
	ST_T,        // load the k-mer's slot: bucket bounds + the records around it
	ST_CAND,     // load the windows of up to four candidate suffixes
	ST_EXT,      // extend one comparison by 32 bytes
	ST_FIN,      // decide; maybe load LCP[best], LCP[best+1]
	ST_BS,       // generic binary search (large buckets, non-ACGT k-mers) …
	ST_BS_R,
	ST_PR_SA,    // … probe: load SA[rank]
	ST_PR_S,     // … probe: load the suffix window
	ST_NBP,
	ST_NBP_R,
	ST_NBS,
	ST_NBS_R
};

enum ExtKind : uint32_t { EXT_LUCKY = 0, EXT_CAND = 1, EXT_PROBE = 2 };

static const uint32_t EXT_COOP_AT = 16 + 8 * 32; // lanes extend this far alone, then ask the wave

struct Data {
	U4 w[4];
};

// the 64 bytes of a slot → its header {lo, hi} and the four SAX records (slot_pack)
PHY_HD void slot_unpack(const U4 &r0, const U4 &r1, const U4 &r2, const U4 &r3, U4 *hdr, Data *d)
{
	const uint32_t w[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
	hdr->x = w[0];
	hdr->y = w[1];
	hdr->z = hdr->w = 0;
	for (int i = 0; i < 4; i++) {
		const uint32_t p = w[4 + 3 * i];
		d->w[i].x = w[2 + 3 * i];
		d->w[i].y = w[3 + 3 * i];
		d->w[i].z = p & 31u;
		d->w[i].w = ((p >> 5) & LCP_CLIP) | (((p >> 18) & LCP_CLIP) << 16);
	}
}

// One chain = the loop of anchor_homologies (process.cxx:245-282) without the
// homology bookkeeping (that is a fold over the accepted anchors, done later).
//
// A step walks through phases in a fixed order — STEP, T, SA, CAND, GEN, EXT,
// FIN — and each phase is "issue a batch of independent 16-byte loads, then
// digest them".  The kernel runs the phases in that order inside one loop trip,
// every phase executed once for all lanes currently in it: lanes stay roughly in
// step with each other, each phase's code runs with most lanes active, and a
// typical step (k-mer bucket of <= 2 suffixes) costs one trip.
struct Chain {
	const uint8_t *Q; // query bytes
	uint32_t qlen;
	uint32_t q;          // this_pos_Q
	uint32_t lq, ls, ll; // last_pos_Q, last_pos_S, last_length
	uint32_t st;
	bool fin;            // a step has just finished: r_* are valid

	U4 qc;               // Q[q .. q+16)
	uint32_t qcode, qv;  // its 2-bit code and how many of its bytes are A,C,G,T and inside the query
	uint32_t flags;      // bit0 lo_known, bit1 hi_known, bit2 k-mer bucket valid
	uint32_t lo, hi, mid;
	uint32_t l_lo, l_hi, p_lo, p_hi; // generic search bookkeeping
	// small-bucket candidates: ranks c_rank0 .. c_rank0+c_n-1
	uint32_t c_rank0, c_n, c_pending;
	uint32_t c_pos0, c_pos1, c_pos2, c_pos3, c_len0, c_len1, c_len2, c_len3;
	uint32_t c_lcp0, c_lcp1, c_lcp2, c_lcp3; // clipped LCP pairs of the candidates' ranks
	uint32_t c_less; // bit i = candidate i < query
	uint32_t c_raw;  // candidates that need their raw bytes compared
	// extension / probe
	uint32_t e_kind, e_idx, e_pos, e_p; // which comparison, bytes known equal, subject position
	uint32_t pr_rank, pr_ret;           // probe rank, state to return to
	uint32_t pr_pos, pr_len, pr_less, pr_lcp;
	// neighbours of the insertion point
	uint32_t lp, pp, lsu, psu;
	uint32_t lcp_p, lcp_s; // clipped LCP pairs of the predecessor's / successor's rank

	// result of the step that just finished (valid while fin)
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
		fin = false;
	}

	PHY_HD bool lucky_ok(const RefIndex &R) const
	{
		uint32_t advance = q - lq;
		return (ls + advance < R.n) && (advance - ll <= R.threshold); // process.cxx:228-235
	}

	PHY_HD void finish_step(uint32_t pos, uint32_t len, bool accepted)
	{
		r_q = q;
		r_s = pos;
		r_len = len;
		r_accepted = accepted;
		lq = accepted ? q : lq; // process.cxx:275-277
		ls = accepted ? pos : ls;
		ll = accepted ? len : ll;
		q += len + 1; // process.cxx:281
		st = ST_STEP;
		fin = true;
	}

	// compare a 16-byte subject window with the query window; true when decided
	PHY_HD bool window_cmp(const U4 &sw, uint32_t *len, uint32_t *less) const
	{
		const uint32_t n = qlen - q;
		const uint32_t m = n < 16 ? n : 16;
		const uint32_t d = first_diff(qc, sw);
		const uint32_t di = d < 15u ? d : 15u;
		const bool mism = d < m;
		*len = mism ? d : n;
		*less = (mism && byte_at(sw, di) < byte_at(qc, di)) ? 1u : 0u;
		return mism || n <= 16;
	}

	PHY_HD void start_ext(uint32_t kind, uint32_t idx, uint32_t p)
	{
		e_kind = kind;
		e_idx = idx;
		e_pos = 16;
		e_p = p;
		st = ST_EXT;
	}

	PHY_HD void set_len(uint32_t i, uint32_t v)
	{
		if (i == 0) c_len0 = v;
		else if (i == 1) c_len1 = v;
		else if (i == 2) c_len2 = v;
		else c_len3 = v;
	}
	PHY_HD uint32_t cpos(uint32_t i) const { return sel4(c_pos0, c_pos1, c_pos2, c_pos3, i); }
	PHY_HD uint32_t clen(uint32_t i) const { return sel4(c_len0, c_len1, c_len2, c_len3, i); }

	PHY_HD void lucky_done(const RefIndex &R, uint32_t len)
	{
		if (len >= R.threshold) finish_step(e_p, len, true); // process.cxx:241
		else begin_search(R);
	}

	PHY_HD void begin_search(const RefIndex &R)
	{
		const bool kv = qv >= R.k;
		flags = kv ? 4u : 0u;
		l_lo = l_hi = 0;
		lo = kv ? (qcode >> (2u * (16u - R.k))) : 0u; // phase T reads the slot of this code
		hi = R.n;
		st = kv ? ST_T : ST_BS;
	}

	// all candidates compared (or one more needs extending)
	PHY_HD void cand_next(const RefIndex &R)
	{
		if (c_pending) {
			uint32_t i = ctz32(c_pending);
			start_ext(EXT_CAND, i, cpos(i));
			return;
		}
		// insertion point: bucket members (at most two on this path) smaller than the
		// query come first
		const uint32_t o = lo - c_rank0;
		const uint32_t s0 = (lo < hi) ? ((c_less >> o) & 1u) : 0u;
		const uint32_t s1 = (lo + 1 < hi) ? (s0 & (c_less >> (o + 1)) & 1u) : 0u;
		const uint32_t ins = lo + s0 + s1;
		const bool hasp = ins > c_rank0, hass = ins < R.n;
		const uint32_t ip = hasp ? ins - 1 - c_rank0 : 0u, is = hass ? ins - c_rank0 : 0u;
		lp = hasp ? clen(ip) : 0u;
		pp = hasp ? cpos(ip) : 0u;
		lcp_p = hasp ? sel4(c_lcp0, c_lcp1, c_lcp2, c_lcp3, ip) : 0u;
		lsu = hass ? clen(is) : 0u;
		psu = hass ? cpos(is) : 0u;
		lcp_s = hass ? sel4(c_lcp0, c_lcp1, c_lcp2, c_lcp3, is) : 0u;
		lo = hi = ins;
		st = ST_FIN;
	}

	// a finished comparison goes back to whoever asked for it
	PHY_HD void deliver(const RefIndex &R, uint32_t len, uint32_t less)
	{
		if (e_kind == EXT_LUCKY) {
			lucky_done(R, len);
		} else if (e_kind == EXT_CAND) {
			set_len(e_idx, len);
			if (less) c_less |= 1u << e_idx;
			c_pending &= ~(1u << e_idx);
			cand_next(R);
		} else {
			pr_len = len;
			pr_less = less;
			st = pr_ret;
		}
	}

	// ── phase STEP: lucky_anchor (process.cxx:227-242), then start the search ──
	PHY_HD uint32_t issue_step(const RefIndex &R, const uint8_t **a0, const uint8_t **a1) const
	{
		*a0 = Q + q;
		if (lucky_ok(R)) {
			*a1 = R.S + (ls + (q - lq));
			return 2;
		}
		return 1;
	}
	// pre_step: take the query window; post_step: lucky_anchor, then start the search.
	// (Split so the kernel can fetch the k-mer's slot together with the lucky window.)
	PHY_HD void pre_step(const U4 &qw)
	{
		qc = qw;
		uint32_t valid, n = qlen - q;
		qcode = window_code(qw, &valid);
		qv = valid < n ? valid : n;
	}
	PHY_HD void post_step(const RefIndex &R, const U4 &sw)
	{
		const bool lucky = lucky_ok(R);
		uint32_t len, less;
		const bool decided = window_cmp(sw, &len, &less);
		const uint32_t try_s = ls + (q - lq);
		begin_search(R); // the default outcome; overridden below
		if (lucky && !decided) start_ext(EXT_LUCKY, 0, try_s);
		e_p = lucky ? try_s : e_p;
		if (lucky && decided && len >= R.threshold) finish_step(try_s, len, true); // process.cxx:241
	}
	PHY_HD void consume_step(const RefIndex &R, const U4 &qw, const U4 &sw)
	{
		pre_step(qw);
		post_step(R, sw);
	}
	// slot of the current window's k-mer, or nullptr when the k-mer is not pure ACGT
	PHY_HD const uint8_t *slot_of_window(const RefIndex &R) const
	{
		if (qv < R.k) return nullptr;
		return (const uint8_t *)(R.SLOT + (size_t)(qcode >> (2u * (16u - R.k))) * SLOT_RECS);
	}

	// ── phase T: the slot of the query's k-mer ──
	PHY_HD const uint8_t *issue_T(const RefIndex &R) const
	{
		return (const uint8_t *)(R.SLOT + (size_t)lo * SLOT_RECS);
	}
	// hdr = record 0; d = records 1..4
	PHY_HD void consume_T(const RefIndex &R, const U4 &hdr, const Data &d)
	{
		lo = hdr.x;
		hi = hdr.y;
		if (hi - lo <= 2) {
			c_rank0 = lo > 0 ? lo - 1 : 0;
			consume_SA(R, d);
		} else {
			st = ST_BS;
		}
	}

	// ── phase SA: the records of the bucket's predecessor, members, successor ──
	// one candidate record against the query window; `on` = the candidate exists
	PHY_HD void take_record(uint32_t i, const U4 &r, uint32_t n, bool on, uint32_t *clen_i)
	{
		uint32_t len = 0, less = 0;
		const uint32_t k = packed_cmp(qcode, qv, n, r.y, r.z, &len, &less);
		*clen_i = len;
		c_less |= ((on && k == 0u) ? less : 0u) << i;
		c_pending |= ((on && k == 1u) ? 1u : 0u) << i;
		c_raw |= ((on && k == 2u) ? 1u : 0u) << i;
	}
	PHY_HD void consume_SA(const RefIndex &R, const Data &d)
	{
		const uint32_t last = hi < R.n ? hi : R.n - 1;
		const uint32_t n = qlen - q;
		c_n = last - c_rank0 + 1;
		c_pos0 = d.w[0].x;
		c_pos1 = d.w[1].x;
		c_pos2 = d.w[2].x;
		c_pos3 = d.w[3].x;
		c_lcp0 = d.w[0].w;
		c_lcp1 = d.w[1].w;
		c_lcp2 = d.w[2].w;
		c_lcp3 = d.w[3].w;
		c_pending = c_less = c_raw = 0;
		take_record(0, d.w[0], n, true, &c_len0);
		take_record(1, d.w[1], n, c_n > 1, &c_len1);
		take_record(2, d.w[2], n, c_n > 2, &c_len2);
		take_record(3, d.w[3], n, c_n > 3, &c_len3);
		if (c_raw) st = ST_CAND; // rare: a '!' / '#' / end of S inside a window
		else cand_next(R);
	}

	// ── phase CAND: raw 16-byte windows for the candidates flagged in c_raw ──
	PHY_HD void raw_one(uint32_t i, const U4 &sw)
	{
		if (!((c_raw >> i) & 1u)) return;
		uint32_t len, less;
		if (window_cmp(sw, &len, &less)) {
			set_len(i, len);
			c_less |= less << i;
		} else {
			c_pending |= 1u << i;
		}
	}
	PHY_HD void consume_cand(const RefIndex &R, const Data &d)
	{
		raw_one(0, d.w[0]);
		raw_one(1, d.w[1]);
		raw_one(2, d.w[2]);
		raw_one(3, d.w[3]);
		c_raw = 0;
		cand_next(R);
	}

	// ── phase GEN: generic binary search, one probe per trip ──
	// part a: run the bookkeeping until a probe needs SA[rank] (returns true)
	PHY_HD bool gen_advance(const RefIndex &R)
	{
		for (;;) {
			switch (st) {
				case ST_BS:
					if (lo < hi) {
						mid = lo + ((hi - lo) >> 1);
						pr_rank = mid;
						pr_ret = ST_BS_R;
						st = ST_PR_SA;
						return true;
					}
					st = ST_NBP;
					break;
				case ST_BS_R:
					if (pr_less) {
						lo = mid + 1;
						l_lo = pr_len;
						p_lo = pr_pos;
						lcp_p = pr_lcp;
						flags |= 1u;
					} else {
						hi = mid;
						l_hi = pr_len;
						p_hi = pr_pos;
						lcp_s = pr_lcp;
						flags |= 2u;
					}
					st = ST_BS;
					break;
				case ST_NBP: // predecessor of the insertion point
					if (flags & 1u) {
						lp = l_lo;
						pp = p_lo;
					} else if (lo == 0 || ((flags & 6u) == 6u && l_hi >= R.k)) {
						lp = 0; // none, or outside a bucket whose member already shares >= k
						pp = 0;
					} else {
						pr_rank = lo - 1;
						pr_ret = ST_NBP_R;
						st = ST_PR_SA;
						return true;
					}
					st = ST_NBS;
					break;
				case ST_NBP_R:
					lp = pr_len;
					pp = pr_pos;
					lcp_p = pr_lcp;
					st = ST_NBS;
					break;
				case ST_NBS:
					if (flags & 2u) {
						lsu = l_hi;
						psu = p_hi;
					} else if (hi >= R.n || ((flags & 5u) == 5u && l_lo >= R.k)) {
						lsu = 0;
						psu = 0;
					} else {
						pr_rank = hi;
						pr_ret = ST_NBS_R;
						st = ST_PR_SA;
						return true;
					}
					st = ST_FIN;
					return false;
				case ST_NBS_R:
					lsu = pr_len;
					psu = pr_pos;
					lcp_s = pr_lcp;
					st = ST_FIN;
					return false;
				default: return st == ST_PR_SA;
			}
		}
	}
	PHY_HD bool in_gen() const { return st >= ST_BS; }
	PHY_HD const uint8_t *issue_probe_sa(const RefIndex &R) const { return (const uint8_t *)(R.SAX + pr_rank); }
	// returns true when the raw suffix window is needed as well
	PHY_HD bool consume_probe_sa(const U4 &r)
	{
		uint32_t len = 0, less = 0;
		pr_pos = r.x;
		pr_lcp = r.w;
		uint32_t k = packed_cmp(qcode, qv, qlen - q, r.y, r.z, &len, &less);
		if (k == 0) {
			pr_len = len;
			pr_less = less;
			st = pr_ret;
			return false;
		}
		if (k == 1) {
			start_ext(EXT_PROBE, 0, pr_pos);
			return false;
		}
		st = ST_PR_S;
		return true;
	}
	PHY_HD const uint8_t *issue_probe_s(const RefIndex &R) const { return R.S + pr_pos; }
	PHY_HD void consume_probe_s(const U4 &sw)
	{
		uint32_t len, less;
		if (window_cmp(sw, &len, &less)) {
			pr_len = len;
			pr_less = less;
			st = pr_ret;
		} else {
			start_ext(EXT_PROBE, 0, pr_pos);
		}
	}

	// ── phase EXT: 32 more bytes of one comparison; true = hand over to the wave ──
	PHY_HD void issue_ext(const RefIndex &R, const uint8_t **a) const
	{
		a[0] = Q + q + e_pos;
		a[1] = Q + q + e_pos + 16;
		a[2] = R.S + e_p + e_pos;
		a[3] = R.S + e_p + e_pos + 16;
	}
	PHY_HD bool consume_ext(const RefIndex &R, const Data &d)
	{
		uint32_t n = qlen - q;
		uint32_t m = n - e_pos; // > 0
		uint32_t dd = first_diff(d.w[0], d.w[2]);
		uint32_t qb = byte_at(d.w[0], dd & 15u), sb = byte_at(d.w[2], dd & 15u);
		if (dd == 16) {
			uint32_t d2 = first_diff(d.w[1], d.w[3]);
			dd = 16 + d2;
			qb = byte_at(d.w[1], d2 & 15u);
			sb = byte_at(d.w[3], d2 & 15u);
		}
		uint32_t lim = m < 32 ? m : 32;
		if (dd < lim) {
			deliver(R, e_pos + dd, sb < qb ? 1u : 0u);
		} else if (m <= 32) {
			deliver(R, n, 0);
		} else {
			e_pos += 32;
			return e_pos >= EXT_COOP_AT;
		}
		return false;
	}

	// ── phase FIN: anchor(), process.cxx:219-225 ──
	// The only neighbour attaining lmax is unique iff the next suffix outward does
	// not share lmax characters with it: LCP[best] < lmax for the predecessor,
	// LCP[best+1] < lmax for the successor.  Both values travel with the record,
	// clipped to LCP_CLIP; returns true when the clipped value cannot decide and
	// the full LCP array has to be read.
	PHY_HD bool fin_needs_lcp(const RefIndex &R)
	{
		const uint32_t lmax = lp > lsu ? lp : lsu;
		const bool pbest = lp > lsu;
		const uint32_t l = pbest ? (lcp_p & 0xffffu) : (lcp_s >> 16);
		const bool cand = lp != lsu && lmax >= R.threshold;
		if (cand && l == LCP_CLIP && lmax >= LCP_CLIP) return true; // clipped: read the full LCP array
		finish_step(pbest ? pp : psu, lmax, cand && l < lmax);
		return false;
	}
	PHY_HD const uint8_t *issue_lcp(const RefIndex &R) const
	{
		// rank lo-1 (lp>lsu) or rank hi (lsu>lp) is the best neighbour
		return (const uint8_t *)(R.LCP + (lp > lsu ? lo - 1 : hi));
	}
	PHY_HD void consume_lcp(const U4 &v)
	{
		uint32_t lmax = lp > lsu ? lp : lsu;
		uint32_t l = lp > lsu ? v.x : v.y; // LCP[best] / LCP[best+1]
		finish_step(lp > lsu ? pp : psu, lmax, l < lmax);
	}
};

// One trip through the phases for a single chain on the CPU (emulation tests);
// the GPU kernel runs the same phases in the same order for 64 lanes at once.
// `tail` finishes a comparison that ran past EXT_COOP_AT (the wave-cooperative
// part on the GPU).
template <class Tail> PHY_HD void chain_trip(Chain &ch, const RefIndex &R, Tail tail)
{
	Data d;
	if (ch.st == ST_STEP) {
		const uint8_t *a0, *a1 = nullptr;
		uint32_t n = ch.issue_step(R, &a0, &a1);
		U4 qw = load16(a0), sw = {0, 0, 0, 0};
		if (n > 1) sw = load16(a1);
		ch.consume_step(R, qw, sw);
	}
	if (ch.st == ST_T) {
		const uint8_t *a = ch.issue_T(R);
		U4 hdr;
		slot_unpack(load16(a), load16(a + 16), load16(a + 32), load16(a + 48), &hdr, &d);
		ch.consume_T(R, hdr, d);
	}
	if (ch.st == ST_CAND) {
		d.w[0] = load16(R.S + ch.c_pos0);
		d.w[1] = d.w[2] = d.w[3] = d.w[0];
		if (ch.c_n > 1) d.w[1] = load16(R.S + ch.c_pos1);
		if (ch.c_n > 2) d.w[2] = load16(R.S + ch.c_pos2);
		if (ch.c_n > 3) d.w[3] = load16(R.S + ch.c_pos3);
		ch.consume_cand(R, d);
	}
	if (ch.in_gen()) {
		if (ch.gen_advance(R)) {
			if (ch.consume_probe_sa(load16(ch.issue_probe_sa(R)))) ch.consume_probe_s(load16(ch.issue_probe_s(R)));
			if (ch.in_gen()) ch.gen_advance(R); // digest the probe now if it was decided
		}
	}
	if (ch.st == ST_EXT) {
		const uint8_t *a[4];
		ch.issue_ext(R, a);
		for (int i = 0; i < 4; i++) d.w[i] = load16(a[i]);
		if (ch.consume_ext(R, d)) {
			uint32_t len, less;
			tail(ch, &len, &less);
			ch.deliver(R, len, less);
		}
	}
	if (ch.st == ST_FIN) {
		if (ch.fin_needs_lcp(R)) ch.consume_lcp(load16(ch.issue_lcp(R)));
	}
}

// ───────────────────────── phase-A work layout ─────────────────────────
//
// Every query is cut into chunks of C positions (C a multiple of 64).  Chunk c of
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
	const uint32_t *items;     // [nchunks] work order: global chunk ids, runs of one query (hostlogic.hpp: plan_chunks)
	const uint32_t *chunk_query; // [nchunks] query id of each global chunk
	uint32_t nchunks;
	// A query's first qnb[j] chunks have C positions each, the rest Cs (multiples of 64; Cs == C
	// when all are alike): long chunks to start every lane on, short ones for the lanes that
	// finish early (hostlogic.hpp: plan_chunks).
	uint32_t C, Cs;
	uint32_t cap, caps;        // anchor slots per long / short chunk
	const uint32_t *qnb;       // [nq] long chunks of each query
	const uint32_t *qanc0;     // [nq] first anchor slot of each query in spec_anchors
	// speculative logs
	Anchor *spec_anchors;      // [sum over chunks of their slots]: a query's chunks back to back
	uint32_t *spec_cnt;        // [nchunks]
	SpecExit *spec_exit;       // [nchunks]
	uint32_t *visited;         // bitmap over the genome buffer: bit (qoff[j] + q) of query j
	// bridges
	BridgeRec *bridge;         // [nchunks]
	PoolBlock *pool;
	uint32_t pool_blocks;
	uint32_t *pool_next;       // bump allocator
	uint32_t *error;           // set nonzero on pool exhaustion / capacity overflow
	// counters for dynamic work fetch
	uint32_t *fetch;           // [2]
	// lean chains: set nonzero when a speculative chain stopped a comparison that ran a chunk length past its
	// chunk's end (lean_core.h: overruns); such chunks have bit 31 of spec_cnt set until they are resolved
	uint32_t *overrun;
};

PHY_HD bool lucky_eligible(uint32_t q, uint32_t aq, uint32_t as, uint32_t al, const RefIndex &R)
{
	uint32_t advance = q - aq;
	return (as + advance < R.n) && (advance - al <= R.threshold);
}

// Where chunk lc (local index) of query j lies: first position, length, log slots and their base.
struct ChunkGeom {
	uint32_t q0, len, cap, log0;
};
PHY_HD ChunkGeom chunk_geom(const PhaseA &A, uint32_t j, uint32_t lc)
{
	const uint32_t nb = A.qnb[j], base = A.qanc0[j];
	ChunkGeom g;
	if (lc < nb) {
		g.q0 = lc * A.C;
		g.len = A.C;
		g.cap = A.cap;
		g.log0 = base + lc * A.cap;
	} else {
		const uint32_t s = lc - nb;
		g.q0 = nb * A.C + s * A.Cs;
		g.len = A.Cs;
		g.cap = A.caps;
		g.log0 = base + nb * A.cap + s * A.caps;
	}
	return g;
}
// local index of the chunk that holds position q of query j
PHY_HD uint32_t chunk_of_pos(const PhaseA &A, uint32_t j, uint32_t q)
{
	const uint32_t nb = A.qnb[j], split = nb * A.C;
	return q < split ? q / A.C : nb + (q - split) / A.Cs;
}
// word of the visited bitmap that holds position q of the chain's query (genomes start at
// multiples of 64 in the buffer, chunks at multiples of 64 in the genome: a chunk owns its words)
PHY_HD uint32_t visited_word(const PhaseA &A, const Chain &ch, uint32_t q)
{
	return (uint32_t)(((uint64_t)(ch.Q - A.qbase) + q) >> 5);
}

// ── speculative chain driver (one lane) ──
struct SpecLane {
	Chain ch;
	uint32_t gc;        // global chunk id; BRIDGE_END when out of work
	uint32_t q_end;     // chunk end (clipped to the query length)
	uint32_t cnt;       // anchors logged
	uint32_t log0, cap; // the chunk's log in spec_anchors
	uint32_t vis_word;  // visited bits being accumulated
	uint32_t vis_idx;   // word index (global) of vis_word

	PHY_HD void start(const PhaseA &A, uint32_t chunk)
	{
		gc = chunk;
		const uint32_t j = A.chunk_query[chunk];
		const ChunkGeom g = chunk_geom(A, j, chunk - A.qchunk0[j]);
		const uint32_t ql = A.qlen[j], e = g.q0 + g.len;
		q_end = e < ql ? e : ql;
		log0 = g.log0;
		cap = g.cap;
		ch.reset(A.qbase + A.qoff[j], ql, g.q0, 0, 0, 0);
		cnt = 0;
		vis_word = 0;
		vis_idx = visited_word(A, ch, g.q0);
	}

	// Called when ch.st == ST_STEP. Returns false when the chunk is finished.
	PHY_HD bool begin_step(const PhaseA &A)
	{
		if (ch.q >= q_end) {
			A.visited[vis_idx] = vis_word;
			A.spec_cnt[gc] = cnt;
			SpecExit x = {ch.q, ch.lq, ch.ls, ch.ll};
			A.spec_exit[gc] = x;
			return false;
		}
		const uint32_t w = visited_word(A, ch, ch.q);
		if (w != vis_idx) { // positions only grow: the previous word is complete
			A.visited[vis_idx] = vis_word;
			vis_idx = w;
			vis_word = 0;
		}
		vis_word |= 1u << (ch.q & 31);
		return true;
	}

	PHY_HD void step_done(const PhaseA &A)
	{
		if (ch.r_accepted) {
			if (cnt < cap) {
				Anchor a = {ch.r_q, ch.r_s, ch.r_len};
				A.spec_anchors[(size_t)log0 + cnt] = a;
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
	uint32_t qj;       // the query
	uint32_t cur_gc;   // chunk of the speculative log being compared against
	uint32_t cur_q0, cur_len; // its first position and length
	uint32_t cur_log;  // its log in spec_anchors
	uint32_t sp_cnt, sp_idx;
	Anchor Ls;         // last anchor the speculative chain had accepted before ch.q
	uint32_t n;        // anchors accepted by this bridge
	uint32_t first_block, cur_block;

	PHY_HD void start(const PhaseA &A, uint32_t chunk)
	{
		src = chunk;
		uint32_t j = A.chunk_query[chunk];
		qj = j;
		SpecExit x = A.spec_exit[chunk];
		ch.reset(A.qbase + A.qoff[j], A.qlen[j], x.q, x.lq, x.ls, x.ll);
		cur_gc = BRIDGE_END;
		cur_q0 = cur_len = cur_log = 0;
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
		if (cur_gc == BRIDGE_END || ch.q - cur_q0 >= cur_len) { // entered another chunk
			const uint32_t lc = chunk_of_pos(A, qj, ch.q);
			const ChunkGeom g = chunk_geom(A, qj, lc);
			cur_gc = A.qchunk0[qj] + lc;
			cur_q0 = g.q0;
			cur_len = g.len;
			cur_log = g.log0;
			sp_cnt = A.spec_cnt[cur_gc];
			sp_idx = 0;
			Ls.q = Ls.s = Ls.len = 0;
		}
		const uint32_t gc = cur_gc;
		const Anchor *log = A.spec_anchors + (size_t)cur_log;
		while (sp_idx < sp_cnt && log[sp_idx].q < ch.q) {
			Ls = log[sp_idx];
			sp_idx++;
		}
		const uint32_t w = A.visited[visited_word(A, ch, ch.q)];
		if ((w >> (ch.q & 31)) & 1u) {
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

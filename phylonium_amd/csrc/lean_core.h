// lean_core.h — the anchor chain's common step on 2-bit packed operands.
//
// Same results as anchor_core.h's Chain (which restates esa::get_match_cached,
// lcp and the lucky_anchor / anchor lambdas of anchor_homologies,
// /root/reference/src/esa.cxx:361-563, src/process.cxx:171-282), organised for the
// GPU differently:
//
//  * Queries and the subject are held a second time as 2-bit codes, 16 bases per
//    dword (Q2, S2), with the positions of their non-ACGT bytes in sorted lists
//    (QBAD, SBAD).  A step whose 16-base query window is pure ACGT — every step
//    except the handful next to a contig join or the query's end — compares
//    codes: one xor + count-leading-zeros per candidate instead of byte loops.
//  * A lane is always in exactly one phase, and every phase is "one batch of
//    loads from up to three addresses (pA: 32 B, pB: 32 B, pY: 8 B), then
//    digest".  The kernel issues that one batch for all 64 lanes whatever
//    phases they are in — one memory round trip per loop trip — and only the
//    digest is per phase.  Phases: STEP (the k-mer's 16-byte slot + lucky window),
//    SEARCH (slot again after a failed long lucky check), SCAN (four SAX records of a
//    bucket with more than two members), EXT (128 more bases of one
//    comparison), REFILL (16 dwords of the query into the lane's ring).
//  * Everything else — a window that touches '!' or the query's end, two
//    candidates that both share 16 bases with the query, buckets beyond
//    LEAN_SCAN_MAX, a clipped LCP that matters — goes to the *slow resolver*,
//    which answers one whole step from the raw bytes by the definition
//    (longest match at the insertion point, unique iff the LCP array says so).
//    On the GPU the wavefront resolves it together (lean_kernels.hip:
//    coop_resolve); lean_resolve_scalar below is the same definition in plain
//    loops for the CPU emulation tests.
//
// Plain C++ over raw pointers: compiled by hipcc for gfx950 and by g++ for
// tests/emul (test infrastructure; the product never runs the CPU build).
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "anchor_core.h"

namespace phy {

// packed companions of RefIndex / PhaseA
struct LeanIndex {
	const uint32_t *S2;       // S as 2-bit codes, base i of word w = S[16w + i] in bits 31-2i..30-2i; non-ACGT -> 0
	const uint32_t *SBAD;     // sorted positions of the non-ACGT bytes of S, then n (the end of S): nsb entries
	uint32_t nsb;
	uint32_t sb_end;        // = n, the list's last entry
	uint32_t sb_first;        // SBAD[0] (with nsb == 2 — one contig, only '#' and the end — no lookup touches memory)
	const uint32_t *Q2;       // the genome buffer in the same packing: word w = buffer bytes [16w, 16w+16)
	const uint32_t *QBAD;     // positions (genome-relative, sorted per genome) of the genomes' non-ACGT bytes
	const uint32_t *qbad_off; // [nq+1]: query j's list is QBAD[qbad_off[j] .. qbad_off[j+1])
	uint32_t force_slow;      // tests: every step through the slow resolver
	unsigned long long *dbg;  // builds with PHY_LEAN_TIMING: per-segment cycle sums of the chain kernels (else unused)
	// The reference's 6-mer interval cache holds over-deep intervals on some tiny multi-contig subjects
	// (src/esa.cxx:174-199; hostlogic.hpp: esa_cache_quirks): entry e = {prefix, k | depth << 8, lo, hi} — a query
	// window whose first 6 bytes are nucleotides and start with the k nucleotides `prefix` is matched by the reference
	// as if its first `depth` bytes were what the suffixes of ranks [lo, hi) share.  With entries present every step
	// goes through the slow resolver (force_slow is set with them), which reproduces that.
	const U4 *quirk;
	uint32_t nquirk;
};

static const uint32_t LEAN_NO_QUIRK = 0xffffffffu;
// the cache entry the window Q (n bytes left in the query) falls under: get_match_cached, src/esa.cxx:542-563
PHY_HD uint32_t lean_quirk_lookup(const LeanIndex &X, const uint8_t *Q, uint32_t n)
{
	if (!X.nquirk || n <= 6u) return LEAN_NO_QUIRK; // qlen <= CACHE_LENGTH: the plain search (esa.cxx:544)
	uint32_t key = 0;
	for (uint32_t i = 0; i < 6u; i++) {
		const uint32_t c = nuc_code(Q[i]);
		if (c > 3u) return LEAN_NO_QUIRK; // no key: the plain search (esa.cxx:552-554)
		key = (key << 2) | c;
	}
	for (uint32_t e = 0; e < X.nquirk; e++) {
		const uint32_t k = X.quirk[e].y & 0xffu;
		if ((key >> (2u * (6u - k))) == X.quirk[e].x) return e;
	}
	return LEAN_NO_QUIRK;
}

enum LeanPhase : uint32_t { LP_STEP = 0, LP_SEARCH, LP_SCAN, LP_EXT, LP_REFILL, LP_SLOW, LP_SLOWEXT,
							 LP_LOOK }; // (LP_LOOK: a lane of the bridge kernel working out a position ahead of a walker, lean_kernels.hip)
static const uint32_t LEAN_LOOK_MAX_LIVE = 8; // the bridge kernel looks ahead when a wavefront has at most this many walkers left

static const uint32_t LEAN_RING_WORDS = 16;         // dwords of the query a lane keeps at hand (256 bases)
static const uint32_t LEAN_EXT_BASES = 128;         // bases per EXT trip
static const uint32_t LEAN_EXT_COOP = 16 + 32 * 128; // a comparison this long is handed to the wavefront
static const uint32_t NO_BAD = 0xffffffffu;
static const uint32_t LEAN_OVERRUN_BIT = 0x80000000u; // in spec_cnt: the chunk's last anchor and exit are lower bounds
static const uint32_t LEAN_LINK_BIT = 0x40000000u;    // ... and its match is continued by the next chunk's (GPU resolve, pass 1)
static const uint32_t LEAN_COUNT_MASK = 0x3fffffffu;

// why a step left the packed path (counted by the CPU emulation build only)
enum LeanSlowWhy : uint32_t { SW_FORCED = 0, SW_QUERY_END, SW_QUERY_BAD, SW_BUCKET, SW_PEND_MANY, SW_CLIP, SW_EXT_QBAD, SW_COUNT };
#if !defined(__HIP_DEVICE_COMPILE__) && defined(LEAN_COUNT_WHY)
static unsigned long long g_lean_why[SW_COUNT];
#define LEAN_WHY(code) (g_lean_why[code]++)
#else
#define LEAN_WHY(code) ((void)0)
#endif

PHY_HD uint32_t popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__popc(x);
#else
	return (uint32_t)__builtin_popcount(x);
#endif
}

// number of leading bases two 16-base codes share
PHY_HD uint32_t lead_eq(uint32_t x) { return x ? clz32(x) >> 1 : 16u; }

// 16 bases starting `o` bases into w0 (o in 0..15), continuing in w1
PHY_HD uint32_t code_window(uint32_t w0, uint32_t w1, uint32_t o)
{
	// (one 64-bit shift, no case for o = 0: with the case the compiler puts the second word's read — an LDS read of the
	// lane's ring in the chain kernels — under a branch that costs every wavefront more than the sixteenth of its lanes
	// that could skip the read saves)
	return (uint32_t)(((((uint64_t)w0 << 32) | w1) << (2u * o)) >> 32);
}

// A pure-ACGT 16-base query window against a suffix record (code of its first 16 bytes with
// the bases after the first non-ACGT byte zeroed, sv = number of leading ACGT bytes).
// The byte that ends a suffix's valid prefix ('!', '#' or the NUL after S) is smaller than
// any nucleotide, so a record that runs out before it differs is shorter AND smaller.
PHY_HD void rec_cmp(uint32_t qcode, uint32_t code, uint32_t sv, uint32_t *len, uint32_t *less, uint32_t *pend)
{
	const uint32_t d = lead_eq(qcode ^ code);
	const uint32_t l = d < sv ? d : sv;
	*len = l;
	*pend = l == 16u ? 1u : 0u;
	*less = sv < 16u ? (code <= qcode ? 1u : 0u) : (code < qcode ? 1u : 0u);
}

PHY_HD uint32_t meta_of_sax(const U4 &r) // a SAX record's z, w in the slot's one-dword form (anchor_core.h: slot_make)
{
	return (r.z & 31u) | ((r.w & LCP_CLIP) << 5) | (((r.w >> 16) & LCP_CLIP) << 18);
}

struct LeanLane {
	uint32_t qw0;            // first Q2 word of the query (its byte offset in the genome buffer / 16)
	uint32_t qlen;
	uint32_t q, lq, ls, ll;  // this_pos_Q, last_pos_Q, last_pos_S, last_length (process.cxx:203-212)
	uint32_t ph;
	bool fin, r_accepted;    // a step has just finished: r_* are valid
	uint32_t r_q, r_s, r_len;
	uint32_t qcode;          // the window's 16 bases
	uint32_t e_kind, e_pos, e_p, e_meta; // EXT: lucky or candidate, bases known equal, subject position, candidate's meta
	uint32_t s_rank, s_last, p_len, p_pos, p_meta, npend; // bucket walk: next rank, last rank that matters, predecessor so far, pending records seen
	uint32_t qb_next, qb_idx, qb_end;    // next non-ACGT position of the query at or after q (NO_BAD: none)
	uint32_t sb_lo, sb_hi;   // S positions [sb_lo, sb_hi) are clean and sb_hi is not (cache of the last SBAD lookup)
	uint32_t wb, we;         // the ring holds words [wb, we) of the query, word w in slot w mod LEAN_RING_WORDS
	// A speculative chain needs a match's exact length only while the match ends inside its chunk: a
	// match that is still running at position q_cap (chunk end + one chunk length) ends the chunk
	// whatever its length, and is left open ("overrun", resolved later: lean_overrun_*).  Without the
	// cap every chunk of a genome that equals the reference over megabases would compare to the end of
	// that stretch — O(L^2 / C) bytes.  NO_BAD: no cap (bridges, which are the true chain).
	uint32_t q_cap;
	bool ovr;                // the step that just finished was cut at q_cap: r_len is a lower bound

	PHY_HD void reset(uint32_t word0, uint32_t query_len, uint32_t q0, uint32_t a_q, uint32_t a_s, uint32_t a_len)
	{
		qw0 = word0;
		qlen = query_len;
		q = q0;
		lq = a_q;
		ls = a_s;
		ll = a_len;
		ph = LP_STEP;
		fin = false;
		sb_lo = 1;
		sb_hi = 0;
		wb = we = 0;
		qb_next = NO_BAD;
		qb_idx = qb_end = 0;
		q_cap = NO_BAD;
		ovr = false;
	}
	// may the running comparison (e_pos bases equal so far) stop here?  Only when the answer no longer
	// depends on its length: a lucky anchor past the threshold, or a candidate whose neighbours' LCPs
	// (both sides, clipped) are already below what has been verified — unique whatever it grows to.
	PHY_HD bool may_cut(uint32_t verified) const
	{
		if (q_cap == NO_BAD || q + verified < q_cap) return false;
		if (e_kind == EXT_LUCKY) return true;
		const uint32_t l1 = (e_meta >> 5) & LCP_CLIP, l2 = (e_meta >> 18) & LCP_CLIP;
		return l1 < verified && l2 < verified && l1 != LCP_CLIP && l2 != LCP_CLIP;
	}
	PHY_HD void finish_cut(uint32_t verified)
	{
		finish(e_p, verified, true);
		ovr = true;
	}
	PHY_HD bool lucky_ok(const RefIndex &R) const
	{
		const uint32_t advance = q - lq;
		return (ls + advance < R.n) && (advance - ll <= R.threshold); // process.cxx:228-235
	}
	PHY_HD void finish(uint32_t pos, uint32_t len, bool accepted)
	{
		r_q = q;
		r_s = pos;
		r_len = len;
		r_accepted = accepted;
		lq = accepted ? q : lq; // process.cxx:275-277
		ls = accepted ? pos : ls;
		ll = accepted ? len : ll;
		q += len + 1; // process.cxx:281
		ph = LP_STEP;
		fin = true;
	}
	PHY_HD void start_ext(uint32_t kind, uint32_t pos, uint32_t meta)
	{
		e_kind = kind;
		e_pos = 16;
		e_p = pos;
		e_meta = meta;
		ph = LP_EXT;
	}
	// the query's bad-position cursor follows q (positions only grow)
	PHY_HD void qbad_seek(const LeanIndex &X)
	{
		while (qb_next < q) {
			qb_idx++;
			qb_next = qb_idx < qb_end ? X.QBAD[qb_idx] : NO_BAD;
		}
	}
	PHY_HD void qbad_start(const LeanIndex &X, uint32_t j)
	{
		uint32_t lo = X.qbad_off[j], hi = X.qbad_off[j + 1];
		qb_end = hi;
		while (lo < hi) { // first entry >= q
			const uint32_t mid = lo + ((hi - lo) >> 1);
			if (X.QBAD[mid] < q) lo = mid + 1;
			else hi = mid;
		}
		qb_idx = lo;
		qb_next = lo < qb_end ? X.QBAD[lo] : NO_BAD;
	}
	// smallest non-ACGT position of S at or after p (p < n; the list ends with n)
	PHY_HD uint32_t sbad_next(const LeanIndex &X, uint32_t p)
	{
		if (p >= sb_lo && p < sb_hi) return sb_hi;
		if (X.nsb == 2) { // a single-contig subject: '#' in the middle and the end
			const bool left = p <= X.sb_first;
			sb_lo = left ? 0u : X.sb_first + 1u;
			sb_hi = left ? X.sb_first : X.sb_end;
			return sb_hi;
		}
		uint32_t lo = 0, hi = X.nsb - 1; // the answer exists: SBAD[nsb-1] = n > p
		while (lo < hi) {
			const uint32_t mid = lo + ((hi - lo) >> 1);
			if (X.SBAD[mid] < p) lo = mid + 1;
			else hi = mid;
		}
		sb_hi = X.SBAD[lo];
		sb_lo = lo ? X.SBAD[lo - 1] + 1u : 0u;
		return sb_hi;
	}
};

// what a trip loads for a lane: 32 bytes at pA, 32 at pB, 8 at pY (byte offsets from the named base)
enum LeanBase : uint32_t { LB_NONE = 0, LB_SLOT, LB_SAX, LB_Q2, LB_S2 };
struct LeanAddr {
	uint32_t baseA, baseB, baseY;
	uint64_t offA, offB, offY;
};

// anchor(), process.cxx:219-225, from the two neighbours of the insertion point: (length, position,
// meta) of the predecessor and the successor (length 0: none)
PHY_HD void lean_fin(LeanLane &ln, const RefIndex &R, uint32_t lp, uint32_t pp, uint32_t mp, uint32_t lsu, uint32_t ps,
					 uint32_t ms)
{
	const bool pbest = lp > lsu;
	const uint32_t lmax = pbest ? lp : lsu;
	const bool cand = lp != lsu && lmax >= R.threshold;
	const uint32_t m = pbest ? mp : ms;
	const uint32_t l = pbest ? (m >> 5) & LCP_CLIP : (m >> 18) & LCP_CLIP; // LCP[best] resp. LCP[best+1], clipped
	if (cand && l == LCP_CLIP && lmax >= LCP_CLIP) { // the clipped value cannot decide
		LEAN_WHY(SW_CLIP);
		ln.ph = LP_SLOW;
		return;
	}
	ln.finish(pbest ? pp : ps, lmax, cand && l < lmax);
}

// a finished comparison (EXT, or the wavefront's long compare) goes back to whoever asked for it
PHY_HD void lean_deliver(LeanLane &ln, const RefIndex &R, uint32_t len, uint32_t less)
{
	if (ln.e_kind == EXT_LUCKY) {
		if (len >= R.threshold) ln.finish(ln.e_p, len, true); // process.cxx:241
		else ln.ph = LP_SEARCH;
		return;
	}
	// the one candidate that shares >= 16 bases with the query is the best neighbour; it is the
	// predecessor iff it is smaller than the query
	const uint32_t l = less ? (ln.e_meta >> 5) & LCP_CLIP : (ln.e_meta >> 18) & LCP_CLIP;
	const bool cand = len >= R.threshold;
	if (cand && l == LCP_CLIP && len >= LCP_CLIP) {
		LEAN_WHY(SW_CLIP);
		ln.ph = LP_SLOW;
		return;
	}
	ln.finish(ln.e_p, len, cand && l < len);
}

// One group of up to four consecutive ranks of the bucket walk (the slot's records, then SAX records
// four at a time): `nex` of them exist, `more` ranks follow.  In rank order the suffixes smaller
// than the query come first, then those that share all 16 bases of the window ("pending": only a
// longer comparison can place them), then the larger ones.  The walk remembers the last smaller
// record (the predecessor so far) and the first pending one, and ends at the first larger record
// — the successor — or when the ranks run out.
PHY_HD void lean_group(LeanLane &ln, const RefIndex &R, uint32_t nex, bool more, uint32_t pos0, uint32_t pos1, uint32_t pos2,
					   uint32_t pos3, uint32_t code0, uint32_t code1, uint32_t code2, uint32_t code3, uint32_t meta0,
					   uint32_t meta1, uint32_t meta2, uint32_t meta3)
{
	uint32_t len0, len1, len2, len3, less = 0, pend = 0, ls_, pe;
	rec_cmp(ln.qcode, code0, meta0 & 31u, &len0, &ls_, &pe);
	less |= ls_;
	pend |= pe;
	rec_cmp(ln.qcode, code1, meta1 & 31u, &len1, &ls_, &pe);
	less |= ls_ << 1;
	pend |= pe << 1;
	rec_cmp(ln.qcode, code2, meta2 & 31u, &len2, &ls_, &pe);
	less |= ls_ << 2;
	pend |= pe << 2;
	rec_cmp(ln.qcode, code3, meta3 & 31u, &len3, &ls_, &pe);
	less |= ls_ << 3;
	pend |= pe << 3;
	const uint32_t exist = (1u << nex) - 1u;
	const uint32_t stop = exist & ~less & ~pend;                // larger than the query
	const uint32_t before = stop ? (stop & (0u - stop)) - 1u : exist; // the records ahead of the successor
	const uint32_t lm = less & ~pend & before, pm = pend & before;
	if (lm) {
		const uint32_t i = 31u - clz32(lm);
		ln.p_len = sel4(len0, len1, len2, len3, i);
		ln.p_pos = sel4(pos0, pos1, pos2, pos3, i);
		ln.p_meta = sel4(meta0, meta1, meta2, meta3, i);
	}
	if (pm) {
		if (ln.npend == 0) {
			const uint32_t i = ctz32(pm);
			ln.e_p = sel4(pos0, pos1, pos2, pos3, i);
			ln.e_meta = sel4(meta0, meta1, meta2, meta3, i);
		}
		ln.npend += popc32(pm);
	}
	if (!stop && more) { // all smaller or pending, and the bucket goes on
		ln.ph = LP_SCAN;
		return;
	}
	if (ln.npend >= 2u) { // several suffixes share the window's 16 bases (a repeat): bytes decide
		LEAN_WHY(SW_PEND_MANY);
		ln.ph = LP_SLOW;
		return;
	}
	if (ln.npend == 1u) { // the one that does is the best neighbour on either side: extend it
		ln.start_ext(EXT_CAND, ln.e_p, ln.e_meta);
		return;
	}
	const uint32_t is = stop ? ctz32(stop) : 0u;
	lean_fin(ln, R, ln.p_len, ln.p_pos, ln.p_meta, stop ? sel4(len0, len1, len2, len3, is) : 0u, sel4(pos0, pos1, pos2, pos3, is),
			 sel4(meta0, meta1, meta2, meta3, is));
}

// The slot of the window's k-mer (anchor_core.h: slot_make) — anchor(), process.cxx:219-225, for a bucket of up to
// two suffixes in one go; written with selects for the three common types (none, one, two members), which share a
// wavefront in any mix.
PHY_HD void lean_search(LeanLane &ln, const RefIndex &R, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
	const uint32_t type = w0 & 7u;
	if (type >= SLOT_MANY) {
		if (type == SLOT_MANY) { // the bucket's members, four SAX records a trip (its predecessor and successor share
			ln.p_len = ln.p_pos = ln.p_meta = 0; // fewer than k bases with the window: never the better neighbour)
			ln.npend = 0;
			ln.s_rank = w1;
			ln.s_last = w2 - 1u;
			ln.ph = LP_SCAN;
		} else {
			LEAN_WHY(SW_BUCKET);
			ln.ph = LP_SLOW;
		}
		return;
	}
	const bool one = type == SLOT_ONE, two = type == SLOT_TWO;
	const uint32_t tmask = 0xffffffffu >> (2u * R.k);
	const uint32_t qt = ln.qcode & tmask;
	// the first member: its whole code (one member) or the code's tail behind the k-mer (two)
	const uint32_t x1 = one ? ln.qcode ^ w2 : qt ^ (w2 & 0xffffu);
	const uint32_t x2 = qt ^ (w2 >> 16);
	const uint32_t sv1 = one ? w3 & 31u : (w0 >> 3) & 31u, sv2 = (w0 >> 8) & 31u;
	const uint32_t d1 = x1 ? clz32(x1) >> 1 : 16u, d2 = x2 ? clz32(x2) >> 1 : 16u;
	const uint32_t l1 = d1 < sv1 ? d1 : sv1;
	const uint32_t l2 = two ? (d2 < sv2 ? d2 : sv2) : 0u;
	const uint32_t c12 = two ? (w0 >> 13) & LCP_CLIP : 0u; // what the two members share (one member: nothing to share)
	const bool pend1 = l1 == 16u, pend2 = l2 == 16u;
	if (type == SLOT_EMPTY) {
		const uint32_t l = (w0 >> 3) & 31u;
		if (l >= R.threshold) { // (thresholds below k: tiny subjects) predecessor or successor could be an anchor
			LEAN_WHY(SW_BUCKET);
			ln.ph = LP_SLOW;
			return;
		}
		ln.finish(0u, l, false);
		return;
	}
	if (pend1 && pend2) { // both share the window's 16 bases (a repeat): bytes decide
		LEAN_WHY(SW_PEND_MANY);
		ln.ph = LP_SLOW;
		return;
	}
	if (pend1 || pend2) { // the one that does is the best neighbour on either side: extend it.  Its LCP towards the other
		// member is c12; towards the bucket's predecessor / successor it is below k, i.e. below what an extension verifies
		const uint32_t meta = one ? w3 : pend1 ? (sv1 | (c12 << 18)) : (sv2 | (c12 << 5));
		ln.start_ext(EXT_CAND, pend1 ? w1 : w3, meta);
		return;
	}
	const bool first = l1 > l2;
	const uint32_t l = first ? l1 : l2;
	ln.finish(first ? w1 : w3, l, l >= R.threshold && l > c12);
}

// four SAX records of ranks s_rank .. s_rank+3
PHY_HD void lean_scan(LeanLane &ln, const RefIndex &R, const U4 &r0, const U4 &r1, const U4 &r2, const U4 &r3)
{
	const uint32_t left = ln.s_last - ln.s_rank + 1; // >= 1
	ln.s_rank += 4;
	lean_group(ln, R, left < 4 ? left : 4, left > 4, r0.x, r1.x, r2.x, r3.x, r0.y, r1.y, r2.y, r3.y, meta_of_sax(r0), meta_of_sax(r1),
			   meta_of_sax(r2), meta_of_sax(r3));
}

// EXT: Q words qw[0..8) starting at the word that holds position q + e_pos, S words sw[0..9)
// starting at the word that holds the matching subject position.  Compares up to 128 bases.
PHY_HD void lean_ext(LeanLane &ln, const RefIndex &R, const LeanIndex &X, const uint32_t *qw, const uint32_t *sw)
{
	const uint32_t back = (ln.q + ln.e_pos) & 15u; // bases of the first Q word that are known equal already
	const uint32_t e0 = ln.e_pos - back;
	const uint32_t sp = ln.e_p + e0, so = sp & 15u;
	uint32_t dd = LEAN_EXT_BASES, lessbit = 0;
#pragma unroll
	for (int w = 7; w >= 0; w--) { // descending: the first differing word wins
		const uint32_t s = code_window(sw[w], sw[w + 1], so);
		const uint32_t x = qw[w] ^ s;
		if (x) {
			dd = 16u * (uint32_t)w + (clz32(x) >> 1);
			lessbit = s < qw[w] ? 1u : 0u;
		}
	}
	const uint32_t n_left = ln.qlen - (ln.q + e0);      // bases of the query from the hop's start
	const uint32_t ds = ln.sbad_next(X, sp) - sp;       // clean bases of S from there
	const uint32_t dq = ln.qb_next - (ln.q + e0);       // clean bases of the query (NO_BAD: plenty)
	uint32_t lim = LEAN_EXT_BASES;
	lim = n_left < lim ? n_left : lim;
	lim = ds < lim ? ds : lim;
	lim = dq < lim ? dq : lim;
	if (dd < lim) {
		lean_deliver(ln, R, e0 + dd, lessbit);
	} else if (lim == LEAN_EXT_BASES) {
		ln.e_pos = e0 + LEAN_EXT_BASES;
		if (ln.may_cut(ln.e_pos)) ln.finish_cut(ln.e_pos);
		else if (ln.e_pos >= LEAN_EXT_COOP) ln.ph = LP_SLOWEXT;
	} else if (dq <= n_left && dq <= ds) {
		LEAN_WHY(SW_EXT_QBAD);
		ln.ph = LP_SLOW; // the match runs into a '!' of the query: bytes decide
	} else if (n_left <= ds) {
		lean_deliver(ln, R, ln.qlen - ln.q, 0); // the query ends first: it is a prefix of the suffix
	} else {
		lean_deliver(ln, R, e0 + ds, 1); // S has a byte below 'A' here, the query a nucleotide
	}
}

// STEP, part 1 (before the loads): can this step take the packed path, and is its window at hand?
// Returns the phase the lane is in for this trip (STEP, REFILL or SLOW).
PHY_HD uint32_t lean_step_phase(const LeanLane &ln, const LeanIndex &X)
{
	const uint32_t n = ln.qlen - ln.q;
	if (X.force_slow || n < 17u || ln.qb_next - ln.q < 16u) {
		LEAN_WHY(X.force_slow ? SW_FORCED : n < 17u ? SW_QUERY_END : SW_QUERY_BAD);
		return LP_SLOW;
	}
	const uint32_t w = ln.q >> 4;
	if (w < ln.wb || w + 1 >= ln.we) return LP_REFILL;
	return LP_STEP;
}

// addresses of a lane's loads for this trip (ph is STEP, SEARCH, SCAN, EXT or REFILL; qcode is set)
PHY_HD LeanAddr lean_addr(const LeanLane &ln, const RefIndex &R)
{
	LeanAddr a;
	a.baseA = a.baseB = a.baseY = LB_NONE;
	a.offA = a.offB = a.offY = 0;
	if (ln.ph == LP_STEP || ln.ph == LP_SEARCH) {
		a.baseA = LB_SLOT;
		a.offA = (uint64_t)(ln.qcode >> (2u * (16u - R.k))) * 16u;
		if (ln.ph == LP_STEP && ln.lucky_ok(R)) {
			a.baseY = LB_S2;
			a.offY = (uint64_t)((ln.ls + (ln.q - ln.lq)) >> 4) * 4u;
		}
	} else if (ln.ph == LP_SCAN) {
		a.baseA = a.baseB = LB_SAX;
		a.offA = (uint64_t)ln.s_rank * 16u;
		a.offB = a.offA + 32;
	} else if (ln.ph == LP_EXT) {
		const uint32_t e0 = ln.e_pos - ((ln.q + ln.e_pos) & 15u);
		a.baseA = LB_Q2;
		a.offA = ((uint64_t)ln.qw0 + ((ln.q + e0) >> 4)) * 4u;
		a.baseB = a.baseY = LB_S2;
		a.offB = (uint64_t)((ln.e_p + e0) >> 4) * 4u;
		a.offY = a.offB + 32;
	} else if (ln.ph == LP_REFILL) {
		a.baseA = a.baseB = LB_Q2;
		a.offA = ((uint64_t)ln.qw0 + (ln.q >> 4)) * 4u;
		a.offB = a.offA + 32;
	}
	return a;
}

// STEP, part 2: d[0..4) = the slot, y0/y1 = the two S2 words of the lucky window
// The step's outcomes side by side — lucky_anchor (process.cxx:227-242) hit or running on, else anchor() (process.cxx:219-225)
// on the slot: an empty bucket, one or two members decided by their codes, one of them to extend, a long bucket to walk, the
// bytes to decide — and the lane's state written with selects.  A wavefront's lanes take all of these in any mix: as cases
// (below) every trip runs every case's code once, each behind its own exec mask and with the lane's fields copied at every
// join; written flat the same arithmetic is a third of the instructions.  (The fields of another phase — e_*, s_*, p_* — and
// r_* are dead while the lane is in STEP: they are written whatever the outcome.)
// with_lucky = false: the slot's part alone (lean_search: SEARCH, and the bridge kernel's look-ahead) — one body for the lanes
// of a trip that are in either
PHY_HD void lean_step_any(LeanLane &ln, const RefIndex &R, const LeanIndex &X, const uint32_t *d, uint32_t y0, uint32_t y1, bool with_lucky)
{
	const uint32_t thr = R.threshold, q = ln.q;
	const bool lucky = with_lucky && ln.lucky_ok(R);
	const uint32_t try_s = ln.ls + (q - ln.lq);
	if (lucky && !(try_s >= ln.sb_lo && try_s < ln.sb_hi)) ln.sbad_next(X, try_s); // (seldom: the next '#' of S behind try_s changes)
	const uint32_t dcode = lead_eq(ln.qcode ^ code_window(y0, y1, try_s & 15u));
	const uint32_t ds = ln.sb_hi - try_s;
	const uint32_t lw = lucky ? (dcode < ds ? dcode : ds) : 0u;
	const bool l_ext = lw >= 16u, l_fin = lucky && !l_ext && lw >= thr, l_any = l_ext || l_fin;
	// the slot (lean_search)
	const uint32_t w0 = d[0], w1 = d[1], w2 = d[2], w3 = d[3];
	const uint32_t type = w0 & 7u;
	const bool one = type == SLOT_ONE, two = type == SLOT_TWO, empty = type == SLOT_EMPTY, many = type == SLOT_MANY;
	const uint32_t tmask = 0xffffffffu >> (2u * R.k);
	const uint32_t qt = ln.qcode & tmask;
	const uint32_t x1 = one ? ln.qcode ^ w2 : qt ^ (w2 & 0xffffu);
	const uint32_t x2 = qt ^ (w2 >> 16);
	const uint32_t sv1 = one ? w3 & 31u : (w0 >> 3) & 31u, sv2 = (w0 >> 8) & 31u;
	const uint32_t d1 = lead_eq(x1), d2 = lead_eq(x2);
	const uint32_t l1 = d1 < sv1 ? d1 : sv1;
	const uint32_t l2 = two ? (d2 < sv2 ? d2 : sv2) : 0u;
	const uint32_t c12 = two ? (w0 >> 13) & LCP_CLIP : 0u;
	const bool pend1 = l1 == 16u, pend2 = l2 == 16u, reg = one || two;
	const uint32_t le = (w0 >> 3) & 31u; // an empty bucket: what its neighbours share with the window
	const bool first = l1 > l2;
	const uint32_t l = first ? l1 : l2;
	const bool s_fin = (empty && le < thr) || (reg && !pend1 && !pend2);
	const bool s_ext = reg && (pend1 != pend2);
	// what the step comes to
	const bool fin = l_fin || (!l_any && s_fin), ext = l_ext || (!l_any && s_ext), scan = !l_any && many;
	const uint32_t pos = l_fin ? try_s : empty ? 0u : first ? w1 : w3;
	const uint32_t len = l_fin ? lw : empty ? le : l;
	const bool acc = l_fin || (fin && reg && l >= thr && l > c12);
	ln.r_q = q; // LeanLane::finish
	ln.r_s = pos;
	ln.r_len = len;
	ln.r_accepted = acc;
	ln.lq = acc ? q : ln.lq;
	ln.ls = acc ? pos : ln.ls;
	ln.ll = acc ? len : ln.ll;
	ln.q = fin ? q + len + 1u : q;
	ln.fin = fin;
	ln.ph = fin ? (uint32_t)LP_STEP : ext ? (uint32_t)LP_EXT : scan ? (uint32_t)LP_SCAN : (uint32_t)LP_SLOW;
	ln.e_kind = l_ext ? (uint32_t)EXT_LUCKY : (uint32_t)EXT_CAND; // LeanLane::start_ext
	ln.e_pos = 16;
	ln.e_p = l_ext ? try_s : pend1 ? w1 : w3;
	const uint32_t m_two = (pend1 ? sv1 : sv2) | (c12 << (pend1 ? 18u : 5u)); // the other member's LCP on the candidate's other side
	ln.e_meta = l_ext ? 0u : one ? w3 : m_two;
	ln.p_len = ln.p_pos = ln.p_meta = 0; // a bucket walk's start (lean_search)
	ln.npend = 0;
	ln.s_rank = w1;
	ln.s_last = w2 - 1u;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(LEAN_COUNT_WHY)
	if (!fin && !ext && !scan) LEAN_WHY(!l_any && reg && pend1 && pend2 ? SW_PEND_MANY : SW_BUCKET);
#endif
}
PHY_HD void lean_step(LeanLane &ln, const RefIndex &R, const LeanIndex &X, const uint32_t *d, uint32_t y0, uint32_t y1)
{
	lean_step_any(ln, R, X, d, y0, y1, true);
}

// ───────────────── the slow resolver's definition, in plain loops (CPU emulation) ─────────────────
// One whole step from the raw bytes: lucky_anchor, else the longest match at the query suffix's
// insertion point in the suffix array, unique iff the LCP array says the next suffix outward does
// not share it (anchor_core.h's header).
#if !defined(__HIP_DEVICE_COMPILE__)
inline void lean_resolve_scalar(LeanLane &ln, const uint8_t *qbase, const RefIndex &R, const LeanIndex &X)
{
	const uint8_t *Q = qbase + ((uint64_t)ln.qw0 << 4) + ln.q;
	const uint32_t n = ln.qlen - ln.q;
	uint32_t skip = 0; // bytes the comparisons take as matched (an over-deep cache entry's depth, else 0)
	auto cmp = [&](uint32_t sa, uint32_t *len, uint32_t *less) {
		uint32_t i = skip;
		while (i < n && Q[i] == R.S[sa + i]) i++; // S ends in zero bytes, query bytes are never zero
		*len = i;
		*less = (i < n && R.S[sa + i] < Q[i]) ? 1u : 0u;
	};
	uint32_t len, less;
	if (ln.lucky_ok(R)) {
		const uint32_t try_s = ln.ls + (ln.q - ln.lq);
		cmp(try_s, &len, &less);
		if (len >= R.threshold) {
			const uint32_t cap_rel = (ln.q_cap != NO_BAD && ln.q_cap > ln.q && ln.q_cap - ln.q < n) ? ln.q_cap - ln.q : NO_BAD;
			if (cap_rel != NO_BAD && len >= cap_rel) { // as the wavefront's resolver: a speculative chain stops at its cap
				ln.finish(try_s, cap_rel, true);
				ln.ovr = true;
				return;
			}
			ln.finish(try_s, len, true);
			return;
		}
	}
	uint32_t lo = 0, hi = R.n, in_lo = 0, in_hi = R.n;
	const uint32_t qe = lean_quirk_lookup(X, Q, n);
	if (qe != LEAN_NO_QUIRK) {
		// get_match_from(query, qlen, ij.l, ij) with the cache's over-deep ij (esa.cxx:556-562): the search goes on
		// below the interval from depth ij.l, whatever the query holds before that
		skip = X.quirk[qe].y >> 8;
		lo = in_lo = X.quirk[qe].z;
		hi = in_hi = X.quirk[qe].w;
	}
	while (lo < hi) {
		const uint32_t mid = lo + ((hi - lo) >> 1);
		cmp(R.SAX[mid].x, &len, &less);
		if (less) lo = mid + 1;
		else hi = mid;
	}
	uint32_t lp = 0, pp = 0, lsu = 0, ps = 0;
	if (lo > in_lo) { // (a rank outside an over-deep interval shares less than its depth: never the better neighbour)
		pp = R.SAX[lo - 1].x;
		cmp(pp, &lp, &less);
	}
	if (lo < in_hi) {
		ps = R.SAX[lo].x;
		cmp(ps, &lsu, &less);
	}
	const bool pbest = lp > lsu;
	const uint32_t lmax = pbest ? lp : lsu;
	const bool cand = lp != lsu && lmax >= R.threshold;
	const uint32_t l = pbest ? R.LCP[lo - 1] : R.LCP[lo + 1];
	ln.finish(pbest ? pp : ps, lmax, cand && l < lmax);
	// (the GPU resolver also cuts a search result at the cap when it is unique whichever side it lies on;
	// here the exact answer is at hand, and a cut one would only be made exact again)
}
inline void lean_resolve_ext_scalar(LeanLane &ln, const uint8_t *qbase, const RefIndex &R)
{
	const uint8_t *Q = qbase + ((uint64_t)ln.qw0 << 4) + ln.q;
	const uint32_t n = ln.qlen - ln.q;
	uint32_t i = ln.e_pos - ((ln.q + ln.e_pos) & 15u);
	const uint32_t lim = ln.may_cut(ln.q_cap - ln.q) ? ln.q_cap - ln.q : n; // (q_cap NO_BAD: may_cut is false)
	while (i < n && i < lim && Q[i] == R.S[ln.e_p + i]) i++;
	if (i < n && i >= lim) {
		ln.finish_cut(i);
		return;
	}
	lean_deliver(ln, R, i, (i < n && R.S[ln.e_p + i] < Q[i]) ? 1u : 0u);
}
#endif

// ───────────────── chunk drivers (the same logs, exits and bridges as anchor_core.h) ─────────────────

struct VisDirect { // finished words of the visited bitmap go straight to memory
	uint32_t *visited;
	PHY_HD void put(uint32_t idx, uint32_t bits) { visited[idx] = bits; }
	PHY_HD void close() {}
	PHY_HD bool cheap(uint32_t) const { return true; }
	PHY_HD void put_cheap(uint32_t idx, uint32_t bits) { put(idx, bits); }
};

PHY_HD uint32_t lean_visited_word(const LeanLane &ln, uint32_t q)
{
	return (ln.qw0 >> 1) + (q >> 5); // genomes start at multiples of 64 bytes: qw0 is a multiple of 4
}

struct LeanSpec {
	LeanLane ln;
	uint32_t gc, q_end, cnt, log0, vis_word, vis_idx; // (the log's capacity is the plan's A.cap)
	uint32_t q_end_full; // the chunk's end on the grid (not clipped to the query's length)
	// The log is written two anchors — 32 bytes, what the memory writes without reading first; a chunk's log starts on
	// such a boundary — at a time: an anchor on its own is a partial write there (the line has left the L2 long before
	// the chain's next anchor, 14 trips on), eleven million of them a pass on C3.  The one that waits:
	Anchor pend;

	// the work item of a chunk and the descriptor of a query, as lean_work_kernel stores them
	static PHY_HD WorkItem make_item(const PhaseA &A, const LeanIndex &X, uint32_t chunk)
	{
		const uint32_t j = A.chunk_query[chunk], q0 = (chunk - A.qchunk0[j]) * A.C;
		uint32_t lo = X.qbad_off[j], hi = X.qbad_off[j + 1];
		const uint32_t end = hi;
		while (lo < hi) { // first entry >= q0
			const uint32_t mid = lo + ((hi - lo) >> 1);
			if (X.QBAD[mid] < q0) lo = mid + 1;
			else hi = mid;
		}
		WorkItem w = {chunk, j, lo, lo < end ? X.QBAD[lo] : NO_BAD};
		return w;
	}
	static PHY_HD QDesc make_qdesc(const PhaseA &A, const LeanIndex &X, uint32_t j)
	{
		QDesc d = {A.qchunk0[j], A.qlen[j], (uint32_t)(A.qoff[j] >> 4), A.qanc0[j], X.qbad_off[j + 1], 0u, 0u, 0u};
		return d;
	}
	PHY_HD void start(const PhaseA &A, const LeanIndex &X, uint32_t chunk)
	{
		const WorkItem w = make_item(A, X, chunk);
		start_desc(A, w, make_qdesc(A, X, w.j));
	}
	// ... without a load: everything comes with the item and the descriptor
	PHY_HD void start_desc(const PhaseA &A, const WorkItem &w, const QDesc &d)
	{
		gc = w.chunk;
		const uint32_t lc = w.chunk - d.qchunk0, q0 = lc * A.C, e = q0 + A.C;
		q_end = e < d.qlen ? e : d.qlen;
		q_end_full = e;
		log0 = d.qanc0 + lc * A.cap;
		ln.reset(d.qword0, d.qlen, q0, 0, 0, 0);
		ln.qb_idx = w.qb_idx;
		ln.qb_end = d.qb_end;
		ln.qb_next = w.qb_next;
		ln.q_cap = e + A.C < e ? NO_BAD : e + A.C;
		cnt = 0;
		vis_word = 0;
		vis_idx = lean_visited_word(ln, q0);
	}
	// called when ln.ph == LP_STEP; false when the chunk is finished.  `vis` takes the finished words of the
	// visited bitmap: put(word index, bits) for one that is complete, close() at the chunk's end.  (The GPU
	// collects 64 bytes of them in LDS before they go out — a word on its own is evicted from the L2, which the
	// slot fetches turn over every few microseconds, long before the lane writes its neighbour, and every such
	// eviction is a partial-line write at the memory; the CPU emulation writes them straight through.)
	template <class Vis> PHY_HD bool begin_step(const PhaseA &A, const LeanIndex &X, Vis &vis)
	{
		if (ln.q >= q_end) {
			vis.put(vis_idx, vis_word);
			vis.close();
			log_close(A);
			A.spec_cnt[gc] = cnt | (ln.ovr ? LEAN_OVERRUN_BIT : 0u); // only a chunk's last step can be cut
			if (ln.ovr) *A.overrun = 1;
			SpecExit x = {ln.q, ln.lq, ln.ls, ln.ll};
			A.spec_exit[gc] = x;
			return false;
		}
		const uint32_t w = lean_visited_word(ln, ln.q);
		if (w != vis_idx) {
			vis.put(vis_idx, vis_word);
			vis_idx = w;
			vis_word = 0;
		}
		vis_word |= 1u << (ln.q & 31);
		ln.qbad_seek(X);
		return true;
	}
	// begin_step where it costs next to nothing — inside the chunk, the finished word of the visited bitmap in the group
	// the sink is collecting — for the chain kernel's STEP-only trips; false: nothing was done, begin_step has to be
	template <class Vis> PHY_HD bool begin_step_fast(const LeanIndex &X, Vis &vis)
	{
		if (ln.q >= q_end) return false;
		const uint32_t w = lean_visited_word(ln, ln.q);
		if (w != vis_idx) {
			if (!vis.cheap(vis_idx)) return false;
			vis.put_cheap(vis_idx, vis_word);
			vis_idx = w;
			vis_word = 0;
		}
		vis_word |= 1u << (ln.q & 31);
		ln.qbad_seek(X);
		return true;
	}
	// the anchor still waiting when the chunk ends
	PHY_HD void log_close(const PhaseA &A)
	{
		const uint32_t n = cnt < A.cap ? cnt : A.cap;
		if (n & 1u) A.spec_anchors[(size_t)log0 + (n - 1u)] = pend;
	}
	PHY_HD void step_done(const PhaseA &A)
	{
		// (the anchor that waits for its pair, kept with selects: as cases the three words are copied at every join)
		const bool acc = ln.r_accepted, room = cnt < A.cap, second = (cnt & 1u) != 0u;
		if (acc && room && second) {
			U4 *dst = (U4 *)(A.spec_anchors + (size_t)log0 + (cnt - 1u));
			dst[0] = U4{pend.q, pend.s, pend.len, 0u};
			dst[1] = U4{ln.r_q, ln.r_s, ln.r_len, 0u};
		}
		if (acc && !room) *A.error = 1;
		const bool keep = acc && room && !second;
		pend.q = keep ? ln.r_q : pend.q;
		pend.s = keep ? ln.r_s : pend.s;
		pend.len = keep ? ln.r_len : pend.len;
		cnt += acc ? 1u : 0u;
	}
};

struct LeanBridge {
	LeanLane ln;
	uint32_t src, qj, cur_gc, cur_q0, cur_len, cur_log, sp_cnt, sp_idx;
	Anchor Ls;
	uint32_t n, first_block, cur_block;
	// what every step would otherwise fetch again: the query's chunk grid, the position of the
	// speculative log's next anchor, the word of the visited bitmap last looked at
	uint32_t g_anc0, g_chunk0, nx_q, vw_idx, vw_word;
	// two words of the visited bitmap fetched ahead (the chain kernel brings them in beside the step's slot: a walker's next
	// position is 13 bases on on average, so two steps in five cross into the next word — a load the next slot's address
	// would have to wait for, in a kernel whose length is its longest walk's dependent loads): words pv_base, pv_base + 1
	static const uint32_t PV_NONE = 0x80000000u; // (no word index comes near: the bitmap has a bit per byte of a 32-bit arena)
	uint32_t pv_base, pv0, pv1;

	PHY_HD void start(const PhaseA &A, const LeanIndex &X, uint32_t chunk)
	{
		src = chunk;
		const uint32_t j = A.chunk_query[chunk];
		qj = j;
		const SpecExit x = A.spec_exit[chunk];
		g_anc0 = A.qanc0[j];
		g_chunk0 = A.qchunk0[j];
		ln.reset((uint32_t)(A.qoff[j] >> 4), A.qlen[j], x.q, x.lq, x.ls, x.ll);
		ln.qbad_start(X, j);
		cur_gc = BRIDGE_END;
		cur_q0 = cur_len = cur_log = 0;
		sp_cnt = sp_idx = 0;
		Ls.q = Ls.s = Ls.len = 0;
		nx_q = 0xffffffffu;
		vw_idx = 0xffffffffu;
		vw_word = 0;
		pv_base = PV_NONE;
		pv0 = pv1 = 0;
		n = 0;
		first_block = cur_block = NO_BLOCK;
	}
	// A bridge that is past its first begin_step — started, in the chunk it stands in, not merged there — as 32 words:
	// what lean_bridge_prepare_kernel leaves for the lanes of the bridge kernel, so that taking a bridge up is one
	// batch of loads instead of the chain of dependent ones start() + begin_step() are (work item -> query -> exit
	// state -> chunk -> log -> visited word), paid inside a trip by every lane of the wavefront.  w[24..32) are the
	// eight words of the query from the one that holds position q: the lane's ring, filled.
	static const uint32_t PACKED_WORDS = 32, PACKED_RING = 8;
	PHY_HD void pack(uint32_t *w) const
	{
		w[0] = src, w[1] = ln.qw0, w[2] = ln.qlen, w[3] = ln.q, w[4] = ln.lq, w[5] = ln.ls, w[6] = ln.ll;
		w[7] = ln.qb_next, w[8] = ln.qb_idx, w[9] = ln.qb_end;
		w[10] = cur_gc, w[11] = sp_cnt, w[12] = sp_idx, w[13] = Ls.q, w[14] = Ls.s, w[15] = Ls.len;
		w[16] = g_anc0, w[17] = g_chunk0, w[18] = nx_q, w[19] = vw_idx, w[20] = vw_word;
		w[21] = w[22] = w[23] = 0;
	}
	PHY_HD void unpack(const PhaseA &A, const uint32_t *w)
	{
		src = w[0];
		qj = 0; // (only start() needs the query's index)
		ln.reset(w[1], w[2], w[3], w[4], w[5], w[6]);
		ln.qb_next = w[7], ln.qb_idx = w[8], ln.qb_end = w[9];
		cur_gc = w[10], sp_cnt = w[11], sp_idx = w[12];
		Ls.q = w[13], Ls.s = w[14], Ls.len = w[15];
		g_anc0 = w[16], g_chunk0 = w[17], nx_q = w[18], vw_idx = w[19], vw_word = w[20];
		const uint32_t lc = cur_gc - g_chunk0; // chunk_geom
		cur_q0 = lc * A.C;
		cur_len = A.C;
		cur_log = g_anc0 + lc * A.cap;
		n = 0;
		first_block = cur_block = NO_BLOCK;
		pv_base = PV_NONE;
		pv0 = pv1 = 0;
		ln.wb = ln.q >> 4; // the ring as the record's last eight words fill it
		ln.we = ln.wb + PACKED_RING;
	}
	PHY_HD void finish(const PhaseA &A, uint32_t target, uint32_t idx_m)
	{
		BridgeRec *b = &A.bridge[src];
		b->target = target;
		b->idx_m = idx_m;
		b->n = n;
		b->block = first_block;
	}
	PHY_HD bool begin_step(const PhaseA &A, const LeanIndex &X, const RefIndex &R)
	{
		if (ln.q >= ln.qlen) {
			finish(A, BRIDGE_END, 0);
			return false;
		}
		if (cur_gc == BRIDGE_END || ln.q - cur_q0 >= cur_len) { // entered another chunk (chunk_of_pos + chunk_geom)
			const uint32_t lc = ln.q / A.C; // chunk_of_pos + chunk_geom
			cur_q0 = lc * A.C;
			cur_len = A.C;
			cur_log = g_anc0 + lc * A.cap;
			cur_gc = g_chunk0 + lc;
			sp_cnt = A.spec_cnt[cur_gc];
			sp_idx = 0;
			Ls.q = Ls.s = Ls.len = 0;
			nx_q = sp_cnt ? A.spec_anchors[(size_t)cur_log].q : 0xffffffffu;
		}
		const uint32_t wi = lean_visited_word(ln, ln.q);
		if (wi != vw_idx) {
			const uint32_t ahead = wi - pv_base;
			vw_idx = wi;
			if (ahead < 2u) vw_word = ahead ? pv1 : pv0;
			else vw_word = A.visited[wi];
		}
		if ((vw_word >> (ln.q & 31)) & 1u) {
			// The chunk's own chain stood here too: the same chain from here on iff the two are in equivalent states,
			// which takes that chain's last anchor before q.  The cursor into its log is moved up only now — a walker
			// that passes five anchors a step through homologous sequence would otherwise pay a chain of dependent
			// loads for each in every trip, and the wavefront with it; most walkers look at the log once, where they merge.
			const Anchor *log = A.spec_anchors + (size_t)cur_log;
			while (nx_q < ln.q) {
				Ls = log[sp_idx];
				sp_idx++;
				nx_q = sp_idx < sp_cnt ? log[sp_idx].q : 0xffffffffu;
			}
			const bool eb = lucky_eligible(ln.q, ln.lq, ln.ls, ln.ll, R);
			const bool es = lucky_eligible(ln.q, Ls.q, Ls.s, Ls.len, R);
			bool merged = false;
			if (!eb && !es) merged = true;
			else if (eb && es && (ln.ls - ln.lq == Ls.s - Ls.q) && (ln.lq + ln.ll == Ls.q + Ls.len)) merged = true;
			if (merged) {
				finish(A, cur_gc, sp_idx);
				return false;
			}
		}
		ln.qbad_seek(X);
		return true;
	}
	template <class Alloc> PHY_HD void step_done(const PhaseA &A, Alloc alloc)
	{
		if (!ln.r_accepted) return;
		Anchor a = {ln.r_q, ln.r_s, ln.r_len};
		if (n < BRIDGE_INLINE) {
			A.bridge[src].a[n] = a;
		} else {
			const uint32_t k = (n - BRIDGE_INLINE) % POOL_BLOCK;
			if (k == 0) {
				const uint32_t nb = alloc();
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

// ───────────────── overruns: the open-ended last match of a speculative chunk ─────────────────
// A cut match (q_s, pos, verified) lies on the diagonal pos - q_s and is known to cover the whole
// next chunk.  If that chunk's chain also ends in a cut match that starts at the chunk's first
// position on the same diagonal, the two are the same maximal match from there on and END AT THE
// SAME POSITION.  So the ends are handed down a run of such chunks from its last member, whose own
// open end is at most two chunk lengths of comparing (the reference's lcp, process.cxx:171-184).
// A genome that equals the reference over megabases costs O(L) bytes this way instead of O(L^2/C).
struct LeanOverrun {
	uint32_t flagged, q_s, pos, verified;
};
PHY_HD LeanOverrun lean_overrun_of(const PhaseA &A, uint32_t gc)
{
	const SpecExit x = A.spec_exit[gc];
	LeanOverrun o = {A.spec_cnt[gc] & LEAN_OVERRUN_BIT, x.lq, x.ls, x.ll};
	return o;
}
// does chunk `nxt` (the next chunk of the same query, first position nxt_q0) continue `o`'s match?
PHY_HD bool lean_overrun_links(const LeanOverrun &o, const LeanOverrun &nxt, uint32_t nxt_q0)
{
	return o.flagged && nxt.flagged && nxt.q_s == nxt_q0 && nxt.pos - nxt.q_s == o.pos - o.q_s;
}
// the match's end is known: fix the chunk's exit state, its last anchor, and clear the flag
PHY_HD void lean_overrun_close(const PhaseA &A, uint32_t j, uint32_t gc, const LeanOverrun &o, uint32_t end_q, bool clear_flags = true)
{
	const uint32_t len = end_q - o.q_s;
	const uint32_t cnt = A.spec_cnt[gc] & LEAN_COUNT_MASK;
	SpecExit x = {o.q_s + len + 1u, o.q_s, o.pos, len};
	A.spec_exit[gc] = x;
	A.spec_anchors[(size_t)chunk_geom(A, j, gc - A.qchunk0[j]).log0 + cnt - 1u].len = len;
	if (clear_flags) A.spec_cnt[gc] = cnt;
}
#if !defined(__HIP_DEVICE_COMPILE__)
// CPU emulation: all of query j's overruns, last chunk first
inline void lean_overrun_resolve_query(const PhaseA &A, const RefIndex &R, uint32_t j)
{
	const uint32_t c0 = A.qchunk0[j], c1 = A.qchunk0[j + 1];
	const uint8_t *Q = A.qbase + A.qoff[j];
	const uint32_t qlen = A.qlen[j];
	LeanOverrun nxt = {0, 0, 0, 0};
	uint32_t nxt_end = 0, nxt_q0 = 0;
	for (uint32_t gc = c1; gc-- > c0;) {
		const LeanOverrun o = lean_overrun_of(A, gc);
		uint32_t end = 0;
		if (o.flagged) {
			if (gc + 1 < c1 && lean_overrun_links(o, nxt, nxt_q0)) {
				end = nxt_end;
			} else {
				uint32_t i = o.verified;
				while (o.q_s + i < qlen && Q[o.q_s + i] == R.S[o.pos + i]) i++;
				end = o.q_s + i;
				if (getenv("EMUL_OVERRUN_TRACE"))
					fprintf(stderr, "direct: query %u chunk %u q_s %u pos %u verified %u -> len %u; next flagged %u q_s %u pos %u q0 %u\n", j, gc - c0,
							o.q_s, o.pos, o.verified, i, nxt.flagged ? 1u : 0u, nxt.q_s, nxt.pos, nxt_q0);
			}
			lean_overrun_close(A, j, gc, o, end);
		}
		nxt = o;
		nxt_end = end;
		nxt_q0 = chunk_geom(A, j, gc - c0).q0;
	}
}
#endif

// ───────────────── one trip of one lane on the CPU (emulation tests) ─────────────────
#if !defined(__HIP_DEVICE_COMPILE__)
struct LeanTables { // what the kernel reaches through the four bases
	const uint8_t *slot, *sax, *q2, *s2;
};
inline const uint8_t *lean_ptr(const LeanTables &T, uint32_t base, uint64_t off)
{
	switch (base) {
		case LB_SLOT: return T.slot + off;
		case LB_SAX: return T.sax + off;
		case LB_Q2: return T.q2 + off;
		case LB_S2: return T.s2 + off;
		default: return T.s2;
	}
}
// ring: the lane's LEAN_RING_WORDS dwords
inline void lean_trip_cpu(LeanLane &ln, uint32_t *ring, const uint8_t *qbase, const RefIndex &R, const LeanIndex &X,
						  const LeanTables &T, uint64_t *slow_steps)
{
	if (ln.ph == LP_STEP) {
		ln.ph = lean_step_phase(ln, X);
		if (ln.ph == LP_STEP) {
			const uint32_t w = ln.q >> 4;
			ln.qcode = code_window(ring[w % LEAN_RING_WORDS], ring[(w + 1) % LEAN_RING_WORDS], ln.q & 15u);
		}
	}
	if (ln.ph == LP_SLOW) {
		lean_resolve_scalar(ln, qbase, R, X);
		(*slow_steps)++;
		return;
	}
	if (ln.ph == LP_SLOWEXT) {
		lean_resolve_ext_scalar(ln, qbase, R);
		return;
	}
	const LeanAddr a = lean_addr(ln, R);
	uint32_t d[16], y[2];
	memcpy(d, lean_ptr(T, a.baseA, a.offA), 32);
	memcpy(d + 8, lean_ptr(T, a.baseB, a.offB), 32);
	memcpy(y, lean_ptr(T, a.baseY, a.offY), 8);
	switch (ln.ph) {
		case LP_STEP: lean_step(ln, R, X, d, y[0], y[1]); break;
		case LP_SEARCH: lean_step_any(ln, R, X, d, y[0], y[1], false); break; // (as the kernels digest it)
		case LP_SCAN: {
			U4 r[4];
			memcpy(r, d, 64);
			lean_scan(ln, R, r[0], r[1], r[2], r[3]);
			break;
		}
		case LP_EXT: {
			uint32_t sw[9];
			memcpy(sw, d + 8, 32);
			sw[8] = y[0];
			ln.wb = (ln.q + (ln.e_pos - ((ln.q + ln.e_pos) & 15u))) >> 4;
			ln.we = ln.wb + 8;
			for (uint32_t i = 0; i < 8; i++) ring[(ln.wb + i) % LEAN_RING_WORDS] = d[i]; // the query words double as the ring's new content
			lean_ext(ln, R, X, d, sw);
			break;
		}
		case LP_REFILL:
			ln.wb = ln.q >> 4;
			ln.we = ln.wb + 16;
			for (uint32_t i = 0; i < 16; i++) ring[(ln.wb + i) % LEAN_RING_WORDS] = d[i];
			ln.ph = LP_STEP;
			break;
		default: break;
	}
}
#endif

// ───────────────── packing (host versions for the emulation; the product packs on the device) ─────────────────
#if !defined(__HIP_DEVICE_COMPILE__)
// 2-bit codes of `n` bytes (n a multiple of 16 is not required: the last word is padded with zero codes)
inline void lean_pack_host(const uint8_t *s, size_t n, uint32_t *out, size_t out_words)
{
	for (size_t w = 0; w < out_words; w++) {
		uint32_t c = 0;
		for (uint32_t i = 0; i < 16; i++) {
			const size_t p = 16 * w + i;
			const uint32_t v = p < n ? nuc_code(s[p]) : 4u;
			c |= (v < 4 ? v : 0u) << (30u - 2u * i);
		}
		out[w] = c;
	}
}
#endif

} // namespace phy

// anchor_core.h — phase A's common types and small helpers: the reference index as the
// kernels see it (RefIndex, SAX records), the work layout of the speculative chunk chains and
// their bridges (PhaseA), and the fold of accepted anchors into homologies.
//
// Replaces, with identical results (together with lean_core.h, which holds the chain's step):
//   esa::get_match_cached / get_match / get_match_from / get_interval
//       (/root/reference/src/esa.cxx:361-563)
//   lcp                         (src/process.cxx:171-184)
//   the lucky_anchor / anchor lambdas and the position chain of
//   anchor_homologies           (src/process.cxx:198-282), the homology bookkeeping (:246-292)
//
// The reference walks a child-table ESA (SA+LCP+CLD+FVC, 26 B/entry, ~25
// dependent reads per match).  What the chain needs from a match is only
// (length of the longest prefix of the query suffix that occurs in S, whether it
// occurs exactly once, where) — SURVEY §3.3.  Here that is answered by a k-mer
// slot table over the suffix array: the longest match is max(lcp(query, pred),
// lcp(query, succ)) at the query's insertion point, and it is unique iff exactly one
// neighbour attains it and the LCP array says the next suffix outward does not share it.
//
// Everything here is plain C++ over raw pointers so that the same code is
// compiled by hipcc for gfx950 and by g++ for the CPU emulation harness under
// tests/emul (test infrastructure; the product never runs it).
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
	const U4 *SLOT;      // 4^k slots of 16 bytes, one per k-mer: see slot_make
	const uint32_t *T;   // 4^k + 1 bucket bounds: the suffixes that start with k-mer c have ranks [T[c], T[c+1])
	uint32_t n;          // |S| = 2L+1
	uint32_t k;          // bucket k-mer length (1..14)
	uint32_t threshold;  // minimum anchor length
};

struct Anchor {
	uint32_t q, s, len; // this_pos_Q, this_pos_S, this_length of an accepted anchor
	uint32_t pad;       // (16 bytes: four of them are a 64-byte store — the speculative chains write their logs that way)
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
// that z and w also fit the one dword a k-mer slot has for them (slot_make, lean_core.h: meta_of_sax).
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

// Slot of k-mer c: 16 bytes — ONE load answers a whole step for a bucket of up to two suffixes, which is all but a
// few per cent of the k-mers at k = ceil(log4 |S|).  The suffixes that start with the k-mer ("members", ranks
// [lo, hi) = [T[c], T[c+1])) share at least k bases with a query window that starts with it; the bucket's
// predecessor (rank lo - 1) and successor (rank hi) share fewer than k, and how many is a property of the k-mer
// alone.  Word 0 carries a type in bits 0..2:
//   SLOT_EMPTY  no member.  The longest match is the better of predecessor and successor: bits 3..7.
//   SLOT_ONE    one member: words 1..3 are its record — SA, the 2-bit codes of its first 16 bytes, and
//               valid length (5 bits) | LCP[r] (13) | LCP[r+1] (13), both clipped (sax_record).  Whatever side of
//               the query it lies on, the neighbour across shares fewer than k bases, so it is the best neighbour
//               and, with both of its LCPs below k, unique.
//   SLOT_TWO    two members (k >= 8: what of their codes lies behind the k-mer fits 16 bits each): word 1 = SA of the
//               first, word 3 = SA of the second, word 2 = the codes' tails (first | second << 16), word 0 also holds
//               their valid lengths (bits 3..7, 8..12) and lcp(first, second) clipped to 13 bits (bits 13..25).
//               With l1, l2 the query's matches with the two: the longer one is the best neighbour, and it is unique
//               iff it is longer than what the two share — wherever the query's insertion point lies:
//                 before both:  l2 = min(l1, c12), best = first, unique iff LCP[its rank + 1] = c12 < l1;
//                 between:      c12 = min(l1, l2), the other LCPs (towards predecessor / successor) are below k;
//                 behind both:  l1 = min(l2, c12), best = second, unique iff LCP[its rank] = c12 < l2.
//   SLOT_MANY   3 .. LEAN_SCAN_MAX members: word 1 = the first member's rank, word 2 = the rank behind the last; the
//               chain walks their SAX records.
//   SLOT_SLOW   more than that: left to the wavefront's resolver.
// 16 rather than 64 bytes (rounds 1-3: {lo, hi} and four 12-byte records): a step then costs one load instruction
// and one translation instead of four — what the random-row rate of a table beyond ~3 GB is made of
// (profiles/r04_gather_bench.jsonl: 16 GB, 64-byte rows in four loads 16.5 G rows/s, 16-byte rows 39.5) — the table
// is a quarter the size (k = 14: 4.3 GB instead of 17), and the digest is two comparisons instead of four.
enum SlotType : uint32_t { SLOT_EMPTY = 0, SLOT_ONE = 1, SLOT_TWO = 2, SLOT_MANY = 3, SLOT_SLOW = 4 };
static const uint32_t LEAN_SCAN_MAX = 24; // buckets with more members go to the slow resolver

// min(number of leading bases the two 16-base codes share, sv)
PHY_HD uint32_t code_lcp(uint32_t a, uint32_t b, uint32_t sv)
{
	const uint32_t x = a ^ b;
	const uint32_t d = x ? clz32(x) >> 1 : 16u;
	return d < sv ? d : sv;
}

// the slot of k-mer `c` of an index with n suffixes: lo = T[c], hi = T[c+1]; sax(r) = the SAX record of rank r.
// The ranks [lo, hi) are the suffixes that start with the k-mer — the members — followed by those that end (in '!',
// '#' or S's end) inside their first k bytes on a prefix of the next k-mers: T counts such a suffix before every k-mer
// that shares its prefix (index_kernels.hip: kmer_hist_kernel), i.e. at the end of the range in front.  Like the
// successor proper they share fewer than k bases with a window that starts with the k-mer: what follows the last
// member is "the successor" whichever it is.
template <class Sax> PHY_HD U4 slot_make(uint64_t c, uint32_t k, uint32_t lo, uint32_t hi, uint32_t n, Sax sax)
{
	const uint32_t kcode = (uint32_t)(c << (2u * (16u - k)));
	U4 out = {0, 0, 0, 0};
	if (hi - lo > LEAN_SCAN_MAX) {
		out.x = SLOT_SLOW;
		return out;
	}
	uint32_t members = 0;
	while (lo + members < hi) {
		const U4 r = sax(lo + members);
		if (r.z < k || (r.y >> (2u * (16u - k))) != (uint32_t)c) break;
		members++;
	}
	if (members == 0) {
		uint32_t l = 0;
		if (lo > 0) {
			const U4 p = sax(lo - 1);
			l = code_lcp(kcode, p.y, p.z);
		}
		if (lo < n) {
			const U4 q = sax(lo);
			const uint32_t l2 = code_lcp(kcode, q.y, q.z);
			l = l2 > l ? l2 : l;
		}
		out.x = SLOT_EMPTY | (l << 3);
	} else if (members == 1) {
		const U4 r = sax(lo);
		out.x = SLOT_ONE;
		out.y = r.x;
		out.z = r.y;
		out.w = (r.z & 31u) | ((r.w & LCP_CLIP) << 5) | (((r.w >> 16) & LCP_CLIP) << 18);
	} else if (members == 2 && k >= 8) {
		const U4 a = sax(lo), b = sax(lo + 1);
		const uint32_t tmask = 0xffffffffu >> (2u * k); // the codes behind the k-mer: 32 - 2k <= 16 bits
		out.x = SLOT_TWO | ((a.z & 31u) << 3) | ((b.z & 31u) << 8) | ((b.w & LCP_CLIP) << 13); // b.w's low half: LCP[lo + 1]
		out.y = a.x;
		out.z = (a.y & tmask) | ((b.y & tmask) << 16);
		out.w = b.x;
	} else {
		out.x = SLOT_MANY;
		out.y = lo;
		out.z = lo + members;
	}
	return out;
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

enum ExtKind : uint32_t { EXT_LUCKY = 0, EXT_CAND = 1 };

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

// The speculative kernel's work queue, ready to start from (lean_kernels.hip: lean_work_kernel makes both with the plan):
// an item = a chunk, its query and where the query's list of non-ACGT positions stands at the chunk's first position;
// a query's descriptor = everything LeanSpec::start needs of it — so that a lane takes its next chunk without a chain
// of dependent loads (item -> query -> six tables -> a binary search), which stalls the other 63 lanes of its wavefront.
struct WorkItem {
	uint32_t chunk, j, qb_idx, qb_next;
};
struct QDesc {
	uint32_t qchunk0, qlen, qword0, qanc0, qb_end, pad0, pad1, pad2;
};

struct PhaseA {
	// inputs
	const uint8_t *qbase;      // all genomes, each followed by >= 64 zero bytes
	const uint64_t *qoff;      // [nq] byte offset of genome j in qbase
	const uint32_t *qlen;      // [nq]
	const uint32_t *qchunk0;   // [nq+1] first global chunk id of each query
	const uint32_t *items;     // [nchunks] work order: global chunk ids, runs of one query (hostlogic.hpp: plan_chunks)
	const uint32_t *chunk_query; // [nchunks] query id of each global chunk
	const WorkItem *work;      // [nchunks] the work order as items to start from
	const QDesc *qdesc;        // [nq]
	uint32_t nchunks;
	uint32_t C;                // positions per chunk (a multiple of 64; a query's last chunk is cut by its length)
	uint32_t cap;              // anchor slots per chunk
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
	// bridges that need walking, as left by lean_bridge_prepare_kernel: records of LeanBridge::PACKED_WORDS words,
	// *bridge_todo of them (lean_core.h); the others merged where their chunk's chain ended
	uint32_t *bridge_start;
	uint32_t *bridge_todo;
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
	ChunkGeom g;
	g.q0 = lc * A.C;
	g.len = A.C;
	g.cap = A.cap;
	g.log0 = A.qanc0[j] + lc * A.cap;
	return g;
}
// local index of the chunk that holds position q of a query
PHY_HD uint32_t chunk_of_pos(const PhaseA &A, uint32_t, uint32_t q) { return q / A.C; }
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

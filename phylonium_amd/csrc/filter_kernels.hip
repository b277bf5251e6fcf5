// filter_kernels.hip — phase A's last step on the device: reverseEh, the sort by
// projected start and filter_overlaps_max for one query per block
// (/root/reference/src/process.cxx:438-443, 354-401; src/process.h:72-80).
//
// The reference sorts with std::sort, which is not stable: when two homologies of
// a query share a projected start, their order — and with it the chain the filter
// picks — is whatever libstdc++'s introsort leaves.  For lists without such a tie
// the order is unique, and that is what this kernel handles; a list with a tie,
// or one that does not fit the block's LDS, is flagged and done by the host
// exactly as before (hostlogic.hpp: sort_filter_order / sort_and_filter).
//
// The chain DP is the O(n log n) form of hostlogic.hpp::filter_overlaps_max: the
// candidates of entry i are a prefix of the pile ordered by end, and a running
// (max score, smallest index) over that order gives the reference's predecessor.
// Sorting is parallel (bitonic, in LDS); the DP, a dependent scan, goes 64 entries
// at a time (see the kernel).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace phy {

static const uint32_t FILT_MAX = 4096;   // entries per query the block's LDS holds
static const uint32_t FILT_THREADS = 1024;
static const uint32_t FILT_WAVES = FILT_THREADS / 64;
static const uint16_t NONE16 = 0xffff;
static const uint32_t FLAG_HOST = 1;    // flag[j]: the host does this query (equal starts, or too long a list)
static const uint32_t FLAG_GENERAL = 2; // ... the general kernel below does (an entangled stretch of more than SEG_MAX entries)
static const uint32_t SEG_MAX = 48;     // longest stretch of mutually entangled entries one thread works out
static const uint32_t FLAG_LONG = 3;    // ... the long-list kernel does (more than FILT_MAX entries: genomes of tens of Mbp)
static const uint32_t LONG_MAX_N = 1u << 17; // entries per list the long-list kernel takes (its scratch slot holds that many)
static const uint32_t LONG_SLOTS = 96;       // scratch slots: that many long lists are worked on at a time; the rest go to the host

// (score, pile index) as one integer whose maximum is "highest score, then smallest index":
// exactly the reference's choice of predecessor (the first k with the largest score,
// process.cxx:374-381).  Scores are sums of lengths (< 2^32), indices < 4096.  0 = nothing.
static __device__ __forceinline__ uint64_t sk_key(uint32_t score, uint32_t idx) { return (uint64_t)score << 12 | (4095u - idx); }
static __device__ __forceinline__ uint32_t sk_index(uint64_t key) { return 4095u - (uint32_t)(key & 4095u); }

struct FilterShared {
	uint64_t by_start[FILT_MAX]; // start << 32 | raw index, sorted: pile order
	uint64_t by_end[FILT_MAX];   // end << 32 | pile position, sorted; later K: sk_key of the finished entries by end position
	uint32_t len[FILT_MAX];      // by pile position
	uint32_t score[FILT_MAX];
	uint16_t pred[FILT_MAX];     // pile position of the predecessor, NONE16 = none
	uint16_t endpos[FILT_MAX];   // position of pile entry p in end order
	uint16_t npred[FILT_MAX];    // P(i): how many entries end at or before entry i starts
	uint16_t jump[FILT_MAX];     // pointer doubling over pred for the backtrack
	uint8_t keep[FILT_MAX];
	uint64_t wred[FILT_WAVES];
	uint32_t wsum[FILT_WAVES];
	uint32_t tie, base, top;
};

// in-place bitonic sort of n2 (a power of two) keys in LDS
static __device__ void bitonic_sort(uint64_t *a, uint32_t n2)
{
	for (uint32_t k = 2; k <= n2; k <<= 1)
		for (uint32_t j = k >> 1; j > 0; j >>= 1) {
			for (uint32_t t = threadIdx.x; t < n2; t += FILT_THREADS) {
				const uint32_t x = t ^ j;
				if (x > t) {
					const uint64_t u = a[t], v = a[x];
					const bool up = (t & k) == 0;
					if ((u > v) == up) {
						a[t] = v;
						a[x] = u;
					}
				}
			}
			// partners less than 64 apart sit in the same wavefront (t and t ^ j differ only in the
			// lane bits), and a wavefront's LDS accesses execute in order: no block barrier needed
			if (j >= 64 || j == 1) __syncthreads();
			else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		}
}

static __device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
	return (uint64_t)(uint32_t)__shfl((int)(uint32_t)v, src, 64) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64) << 32;
}

// raw: fold_kernel's output (index_reference, index_query, length before reverseEh), query j at
// raw[raw_base[j] .. + raw_cnt[j]).  Output: the filtered list in device form at
// out[rng[2j] .. rng[2j+1]) — slots are handed out by an atomic counter, their order is
// irrelevant — or flag[j] = 1 when the host has to do this query.  ref_local: the query that
// is the reference itself (its list is [(0,0,L)] iff L/2 >= threshold, process.cxx:285-292),
// or 0xffffffff.
//
// The chain DP  score(i) = len(i) + max{score(k) : end(k) <= start(i)}  is a dependent scan, and
// done by one lane it costs ~700 cycles per entry (every LDS access a round trip).  Here one
// wavefront takes 64 entries per round (see the loop).
__global__ __launch_bounds__(FILT_THREADS) void sort_filter_kernel(const RawHom *__restrict__ raw,
																	const uint64_t *__restrict__ raw_base,
																	const uint32_t *__restrict__ raw_cnt, uint32_t j0, uint32_t border,
																	uint32_t threshold, uint32_t ref_local,
																	DevHom *__restrict__ out, uint32_t *__restrict__ rng,
																	uint32_t *__restrict__ total, uint32_t *__restrict__ flag,
																	uint32_t only_flagged)
{
	__shared__ FilterShared sh;
	const uint32_t j = j0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	if (only_flagged && flag[j] != FLAG_GENERAL) return; // second pass: the lists the segment-wise kernel handed over
	const RawHom *r = raw + raw_base[j];
	const uint32_t n = raw_cnt[j];
	if (j == ref_local) {
		if (tid == 0) {
			uint32_t b = 0, w = border / 2 >= threshold ? 1u : 0u;
			if (w) {
				b = atomicAdd(total, 1u);
				out[b] = DevHom{0u, 0u, border, 0u};
			}
			rng[2 * j] = b;
			rng[2 * j + 1] = b + w;
			flag[j] = 0;
		}
		return;
	}
	if (n > FILT_MAX) {
		if (tid == 0) {
			flag[j] = FLAG_HOST;
			rng[2 * j] = rng[2 * j + 1] = 0; // an empty list until the host has done this query
		}
		return;
	}
	uint32_t n2 = 1;
	while (n2 < n) n2 <<= 1;
	if (tid == 0) sh.tie = 0;
	// reverseEh (process.h:72-80): a hit in the reverse half of S projects to 2L+1-len-iref
	for (uint32_t t = tid; t < n2; t += FILT_THREADS) {
		uint64_t key = ~0ull; // padding sorts last
		if (t < n) {
			const RawHom h = r[t];
			const uint32_t start = h.iref >= border ? 2u * border + 1u - h.len - h.iref : h.iref;
			key = (uint64_t)start << 32 | t;
		}
		sh.by_start[t] = key;
	}
	__syncthreads();
	bitonic_sort(sh.by_start, n2);
	for (uint32_t p = tid; p < n; p += FILT_THREADS) {
		const uint64_t key = sh.by_start[p];
		if (p + 1 < n && (uint32_t)(sh.by_start[p + 1] >> 32) == (uint32_t)(key >> 32)) sh.tie = 1; // benign race: same value
		const uint32_t l = r[(uint32_t)key].len;
		sh.len[p] = l;
		sh.by_end[p] = ((key >> 32) + l) << 32 | p;
	}
	for (uint32_t p = n + tid; p < n2; p += FILT_THREADS) sh.by_end[p] = ~0ull;
	__syncthreads();
	if (sh.tie) {
		if (tid == 0) {
			flag[j] = FLAG_HOST;
			rng[2 * j] = rng[2 * j + 1] = 0;
		}
		return;
	}
	bitonic_sort(sh.by_end, n2);
	// end order: position of every entry, and P(i) = #{q : end(q) <= start(i)} by binary search
	for (uint32_t q = tid; q < n; q += FILT_THREADS) sh.endpos[(uint32_t)sh.by_end[q] & 0xffffu] = (uint16_t)q;
	for (uint32_t i = tid; i < n; i += FILT_THREADS) {
		const uint32_t start_i = (uint32_t)(sh.by_start[i] >> 32);
		uint32_t lo = 0, hi = n; // first q with end(q) > start_i
		while (lo < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			if ((uint32_t)(sh.by_end[mid] >> 32) <= start_i) lo = mid + 1;
			else hi = mid;
		}
		sh.npred[i] = (uint16_t)lo;
		sh.keep[i] = 0;
	}
	__syncthreads();
	uint64_t *K = sh.by_end; // the ends are not needed any more (end = start + len): now sk_key of every finished entry, by end position
	for (uint32_t q = tid; q < n; q += FILT_THREADS) K[q] = 0;
	__syncthreads();
	if (wave == 0) {
		// One wavefront, 64 entries per round, no block barriers.  The candidates of entry i are the
		// entries at end positions < P(i); P is nondecreasing in i, and an entry that ends before
		// the round's first entry starts is already finished (its index is smaller), so
		//   G    = best key at end positions < gpos, kept in a register and advanced by scans of K,
		//   base = max(G, K[P(first) .. P(i)-1])  — a prefix maximum over a few dozen positions,
		// cover everything finished in earlier rounds, and the predecessors inside the round are
		// settled lane after lane with v_readlane broadcasts.
		uint64_t G = 0;
		uint32_t gpos = 0;
		for (uint32_t a = 0; a < n; a += 64) {
			const uint32_t cnt = n - a < 64 ? n - a : 64;
			const uint32_t i = a + lane;
			const bool on = lane < cnt;
			uint32_t start_i = 0, len_i = 0, np = 0, ep = 0;
			if (on) {
				start_i = (uint32_t)(sh.by_start[i] >> 32);
				len_i = sh.len[i];
				np = sh.npred[i];
				ep = sh.endpos[i];
			}
			const uint32_t end_i = start_i + len_i;
			const uint32_t np0 = (uint32_t)__builtin_amdgcn_readlane((int)np, 0);
			const uint32_t np_last = (uint32_t)__builtin_amdgcn_readlane((int)np, (int)(cnt - 1));
			for (uint32_t q0 = gpos; q0 < np0; q0 += 64) {
				uint64_t v = q0 + lane < np0 ? K[q0 + lane] : 0;
				for (uint32_t d = 32; d > 0; d >>= 1) {
					const uint64_t o = shfl64(v, (int)(lane ^ d));
					if (o > v) v = o;
				}
				if (v > G) G = v;
			}
			if (np0 > gpos) gpos = np0;
			uint64_t running = G;
			for (uint32_t q0 = np0; q0 < np_last; q0 += 64) {
				uint64_t v = q0 + lane < np_last ? K[q0 + lane] : 0;
				for (uint32_t d = 1; d < 64; d <<= 1) { // inclusive prefix maximum
					const uint64_t o = shfl64(v, (int)(lane >= d ? lane - d : lane));
					if (lane >= d && o > v) v = o;
				}
				const bool need = on && np > q0;
				const uint32_t src = need ? (np - 1 - q0 < 63 ? np - 1 - q0 : 63) : 0;
				const uint64_t t = shfl64(v, (int)src);
				if (need && t > running) running = t;
			}
			// Inside the round plain scores do: lanes are visited in index order and only a strictly
			// higher score replaces the running one, which is the smallest-index rule again.
			uint32_t run_score = (uint32_t)(running >> 12), run_idx = running ? sk_index(running) : 0xffffu;
			uint32_t score_i = 0;
			for (uint32_t s = 0; s < cnt; s++) {
				// lane s has seen every earlier lane's result: its own is final now
				if (lane == s) score_i = run_score + len_i;
				const uint32_t score_s = (uint32_t)__builtin_amdgcn_readlane((int)score_i, (int)s);
				const uint32_t end_s = (uint32_t)__builtin_amdgcn_readlane((int)end_i, (int)s);
				if (lane > s && end_s <= start_i && score_s > run_score) {
					run_score = score_s;
					run_idx = a + s;
				}
			}
			if (on) {
				sh.score[i] = score_i;
				sh.pred[i] = (uint16_t)run_idx;
				K[ep] = sk_key(score_i, i);
			}
		}
	}
	__syncthreads();
	// std::max_element over (0, score[0..n)): the first maximum — the largest (score, smallest index)
	{
		uint64_t best = 0;
		for (uint32_t i = tid; i < n; i += FILT_THREADS) {
			const uint64_t k2 = sk_key(sh.score[i], i);
			if (k2 > best) best = k2;
		}
		for (uint32_t d = 32; d > 0; d >>= 1) {
			const uint64_t o = shfl64(best, (int)(lane ^ d));
			if (o > best) best = o;
		}
		if (lane == 0) sh.wred[wave] = best;
		__syncthreads();
		if (tid == 0) {
			uint64_t b2 = 0;
			for (uint32_t w2 = 0; w2 < FILT_WAVES; w2++)
				if (sh.wred[w2] > b2) b2 = sh.wred[w2];
			sh.top = n ? sk_index(b2) : 0xffffffffu; // lengths are positive, so any entry beats the initial 0
		}
		__syncthreads();
	}
	// backtrack from the top along pred: reachability by pointer doubling
	for (uint32_t i = tid; i < n; i += FILT_THREADS) sh.jump[i] = sh.pred[i];
	if (tid == 0 && n) sh.keep[sh.top] = 1;
	__syncthreads();
	for (uint32_t round = 0; (1u << round) < n; round++) {
		for (uint32_t i = tid; i < n; i += FILT_THREADS)
			if (sh.keep[i] && sh.jump[i] != NONE16) sh.keep[sh.jump[i]] = 1;
		__syncthreads();
		uint16_t nj[FILT_MAX / FILT_THREADS];
		uint32_t c = 0;
		for (uint32_t i = tid; i < n; i += FILT_THREADS, c++) {
			const uint16_t j1 = sh.jump[i];
			nj[c] = j1 == NONE16 ? NONE16 : sh.jump[j1];
		}
		__syncthreads();
		c = 0;
		for (uint32_t i = tid; i < n; i += FILT_THREADS, c++) sh.jump[i] = nj[c];
		__syncthreads();
	}
	// the kept entries leave in pile order: block-wide prefix count of keep[]
	uint32_t done = 0, base = 0;
	{
		uint32_t mine = 0;
		for (uint32_t i = tid; i < n; i += FILT_THREADS) mine += sh.keep[i];
		for (uint32_t d = 32; d > 0; d >>= 1) mine += (uint32_t)__shfl((int)mine, (int)(lane ^ d), 64);
		if (lane == 0) sh.wsum[wave] = mine;
		__syncthreads();
		if (tid == 0) {
			uint32_t w = 0;
			for (uint32_t w2 = 0; w2 < FILT_WAVES; w2++) w += sh.wsum[w2];
			const uint32_t b = w ? atomicAdd(total, w) : 0u;
			sh.base = b;
			rng[2 * j] = b;
			rng[2 * j + 1] = b + w;
			flag[j] = 0;
		}
		__syncthreads();
		base = sh.base;
	}
	for (uint32_t p0 = 0; p0 < n; p0 += FILT_THREADS) {
		const uint32_t p = p0 + tid;
		const uint32_t k = (p < n && sh.keep[p]) ? 1u : 0u;
		const uint64_t m = __ballot(k);
		if (lane == 0) sh.wsum[wave] = (uint32_t)__popcll(m);
		__syncthreads();
		uint32_t off = done + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
		for (uint32_t w2 = 0; w2 < wave; w2++) off += sh.wsum[w2];
		if (k) {
			const uint64_t key = sh.by_start[p];
			const RawHom h = r[(uint32_t)key];
			out[base + off] = DevHom{(uint32_t)(key >> 32), h.iq, h.len, h.iref >= border ? 1u : 0u};
		}
		for (uint32_t w2 = 0; w2 < FILT_WAVES; w2++) done += sh.wsum[w2];
		__syncthreads();
	}
}

// ── the same filter, stretch by stretch ──
// In pile order (by projected start) put a cut before entry i when every earlier entry ends at or before
// i starts (prefix maximum of the ends).  The stretches between cuts are independent: every entry of an
// earlier stretch is a candidate predecessor of every entry of a later one, and any candidate inside the
// own stretch outscores them all, so the reference's DP (process.cxx:354-401) restricted to a stretch
// gives the same predecessors inside it, the stretch's first best-scoring entry is where the chain of all
// later stretches continues, and the kept set is the union of the stretches' own best chains.  After
// anchoring most stretches are single entries (homologies rarely overlap on the reference), so instead
// of a dependent scan over the whole list (the kernel above: 0.25 ms) every thread works out the few
// short stretches that start at its entries.  A stretch longer than SEG_MAX hands the query to the
// kernel above (flag 2); equal starts or more than FILT_MAX entries to the host (flag 1), as there.
struct SegShared {
	uint64_t by_start[FILT_MAX]; // start << 32 | raw index, sorted: pile order
	uint32_t len[FILT_MAX];
	uint32_t score[FILT_MAX];    // first the prefix maximum of the ends, then the stretch-local scores
	uint16_t pred[FILT_MAX];
	uint8_t keep[FILT_MAX];      // bit 0 kept, bit 1 a stretch starts here
	uint32_t wmax[FILT_WAVES], wsum[FILT_WAVES];
	uint32_t tie, general, base;
};

__global__ __launch_bounds__(FILT_THREADS) void sort_filter_seg_kernel(const RawHom *__restrict__ raw,
																		const uint64_t *__restrict__ raw_base,
																		const uint32_t *__restrict__ raw_cnt, uint32_t j0, uint32_t border,
																		uint32_t threshold, uint32_t ref_local,
																		DevHom *__restrict__ out, uint32_t *__restrict__ rng,
																		uint32_t *__restrict__ total, uint32_t *__restrict__ flag,
																		uint32_t long_ok)
{
	__shared__ SegShared sh;
	const uint32_t j = j0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const RawHom *r = raw + raw_base[j];
	const uint32_t n = raw_cnt[j];
	if (j == ref_local) {
		if (tid == 0) {
			uint32_t b = 0, w = border / 2 >= threshold ? 1u : 0u;
			if (w) {
				b = atomicAdd(total, 1u);
				out[b] = DevHom{0u, 0u, border, 0u};
			}
			rng[2 * j] = b;
			rng[2 * j + 1] = b + w;
			flag[j] = 0;
		}
		return;
	}
	if (n > FILT_MAX) {
		if (tid == 0) {
			flag[j] = (long_ok && n <= LONG_MAX_N) ? FLAG_LONG : FLAG_HOST;
			rng[2 * j] = rng[2 * j + 1] = 0;
		}
		return;
	}
	uint32_t n2 = 1;
	while (n2 < n) n2 <<= 1;
	if (tid == 0) sh.tie = sh.general = 0;
	for (uint32_t t = tid; t < n2; t += FILT_THREADS) { // reverseEh (process.h:72-80)
		uint64_t key = ~0ull;
		if (t < n) {
			const RawHom h = r[t];
			const uint32_t start = h.iref >= border ? 2u * border + 1u - h.len - h.iref : h.iref;
			key = (uint64_t)start << 32 | t;
		}
		sh.by_start[t] = key;
	}
	__syncthreads();
	bitonic_sort(sh.by_start, n2);
	// lengths by pile position, equal starts, and the prefix maximum of the ends (4 consecutive entries per thread)
	const uint32_t p4 = tid * 4u;
	uint32_t e[4], run = 0;
#pragma unroll
	for (uint32_t u = 0; u < 4; u++) {
		const uint32_t p = p4 + u;
		e[u] = 0;
		if (p < n) {
			const uint64_t key = sh.by_start[p];
			if (p + 1 < n && (uint32_t)(sh.by_start[p + 1] >> 32) == (uint32_t)(key >> 32)) sh.tie = 1; // benign race: same value
			const uint32_t l = r[(uint32_t)key].len;
			sh.len[p] = l;
			e[u] = (uint32_t)(key >> 32) + l;
		}
		run = e[u] > run ? e[u] : run;
		e[u] = run; // inclusive maximum inside the thread
	}
	uint32_t incl = run;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
		if ((int)lane >= d && t > incl) incl = t;
	}
	if (lane == 63) sh.wmax[wave] = incl;
	__syncthreads();
	if (sh.tie) {
		if (tid == 0) {
			flag[j] = FLAG_HOST;
			rng[2 * j] = rng[2 * j + 1] = 0;
		}
		return;
	}
	uint32_t before = (uint32_t)__shfl_up((int)incl, 1, 64); // maximum of the ends of all entries of earlier threads
	if (lane == 0) before = 0;
	for (uint32_t w2 = 0; w2 < wave; w2++) before = sh.wmax[w2] > before ? sh.wmax[w2] : before;
#pragma unroll
	for (uint32_t u = 0; u < 4; u++) {
		const uint32_t p = p4 + u;
		if (p < n) {
			const uint32_t prev_max = u ? (e[u - 1] > before ? e[u - 1] : before) : before;
			sh.keep[p] = prev_max <= (uint32_t)(sh.by_start[p] >> 32) ? 2u : 0u; // a stretch starts here
		}
	}
	__syncthreads();
	// every thread: the stretches that start at its entries
#pragma unroll 1
	for (uint32_t u = 0; u < 4; u++) {
		const uint32_t a = p4 + u;
		if (a >= n || !(sh.keep[a] & 2u)) continue;
		uint32_t b = a + 1;
		while (b < n && !(sh.keep[b] & 2u)) b++;
		if (b - a == 1) {
			sh.keep[a] |= 1u;
			continue;
		}
		if (b - a > SEG_MAX) {
			sh.general = 1;
			continue;
		}
		uint32_t top = a, top_score = 0;
		for (uint32_t i = a; i < b; i++) { // filter_overlaps_max inside the stretch: first best predecessor, first best end
			const uint32_t start_i = (uint32_t)(sh.by_start[i] >> 32);
			uint32_t best = 0, bk = NONE16;
			for (uint32_t k = a; k < i; k++) {
				const uint32_t end_k = (uint32_t)(sh.by_start[k] >> 32) + sh.len[k];
				const uint32_t sc = sh.score[k];
				if (end_k <= start_i && sc > best) {
					best = sc;
					bk = k;
				}
			}
			const uint32_t sc_i = best + sh.len[i];
			sh.score[i] = sc_i;
			sh.pred[i] = (uint16_t)bk;
			if (sc_i > top_score) {
				top_score = sc_i;
				top = i;
			}
		}
		for (uint32_t i = top; i != NONE16; i = sh.pred[i]) sh.keep[i] |= 1u;
	}
	__syncthreads();
	if (sh.general) {
		if (tid == 0) {
			flag[j] = FLAG_GENERAL;
			rng[2 * j] = rng[2 * j + 1] = 0;
		}
		return;
	}
	// the kept entries leave in pile order
	uint32_t done = 0, base = 0;
	{
		uint32_t mine = 0;
		for (uint32_t i = tid; i < n; i += FILT_THREADS) mine += sh.keep[i] & 1u;
		for (uint32_t d = 32; d > 0; d >>= 1) mine += (uint32_t)__shfl((int)mine, (int)(lane ^ d), 64);
		if (lane == 0) sh.wsum[wave] = mine;
		__syncthreads();
		if (tid == 0) {
			uint32_t w = 0;
			for (uint32_t w2 = 0; w2 < FILT_WAVES; w2++) w += sh.wsum[w2];
			const uint32_t b = w ? atomicAdd(total, w) : 0u;
			sh.base = b;
			rng[2 * j] = b;
			rng[2 * j + 1] = b + w;
			flag[j] = 0;
		}
		__syncthreads();
		base = sh.base;
	}
	for (uint32_t p0 = 0; p0 < n; p0 += FILT_THREADS) {
		const uint32_t p = p0 + tid;
		const uint32_t k = (p < n && (sh.keep[p] & 1u)) ? 1u : 0u;
		const uint64_t m = __ballot(k);
		if (lane == 0) sh.wsum[wave] = (uint32_t)__popcll(m);
		__syncthreads();
		uint32_t off = done + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
		for (uint32_t w2 = 0; w2 < wave; w2++) off += sh.wsum[w2];
		if (k) {
			const uint64_t key = sh.by_start[p];
			const RawHom h = r[(uint32_t)key];
			out[base + off] = DevHom{(uint32_t)(key >> 32), h.iq, h.len, h.iref >= border ? 1u : 0u};
		}
		for (uint32_t w2 = 0; w2 < FILT_WAVES; w2++) done += sh.wsum[w2];
		__syncthreads();
	}
}


// ── lists of more than FILT_MAX entries (queries of tens of Mbp: ~330 homologies per Mbp) ──
// The same stretch-wise filter with the list in global memory.  A list takes a scratch slot (a list that finds none
// goes to the host); its keys are sorted by an LSD radix sort spread over the whole device — four passes over the bits
// a projected start can have, a pass is two launches over (slot, tile of FILT_MAX keys): the tiles' digit counts, then
// every tile ranks its keys (stable: a wavefront's keys by ballot, the wavefronts in index order) and scatters them
// behind the smaller digits of the list and the same digit of the tiles before it.  Then one block per list: prefix
// maximum of the ends, cuts, stretches and output tile by tile with a carry.  (Round 2 sorted inside that one block:
// 2000 barrier-separated passes for C5's 33 k entries per list; round 3 ran a bitonic network over the device, eleven
// launches over lists padded to a power of two: 0.46 of the filter's 0.74 ms on C5.)
struct LongMeta {
	uint32_t n, j, pad0, pad1;
};
static const uint32_t LONG_TILES = LONG_MAX_N / FILT_MAX; // tiles per slot
static const uint32_t RADIX_BINS = 256;
struct LongScratch {
	uint64_t *keys;   // [LONG_SLOTS][LONG_MAX_N] start << 32 | raw index; sorted when the passes are through
	uint64_t *keys2;  // the passes' other buffer
	uint32_t *counts; // [LONG_SLOTS][LONG_TILES][RADIX_BINS] a pass's digit counts by tile
	uint32_t *ends;   // [LONG_SLOTS][LONG_MAX_N] end of the entry at that pile position
	LongMeta *meta;   // [LONG_SLOTS] the list in the slot: entries, query
	uint32_t *next_slot;
};

// a slot for every long list (rng[2j] holds it)
__global__ __launch_bounds__(64) void long_slots_kernel(const uint32_t *__restrict__ raw_cnt, uint32_t j0, uint32_t nq, uint32_t *__restrict__ rng,
														 uint32_t *__restrict__ flag, LongScratch S)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nq) return;
	const uint32_t j = j0 + t;
	if (flag[j] != FLAG_LONG) return;
	const uint32_t slot = atomicAdd(S.next_slot, 1u);
	if (slot >= LONG_SLOTS) {
		flag[j] = FLAG_HOST;
		return;
	}
	S.meta[slot] = LongMeta{raw_cnt[j], j, 0u, 0u};
	rng[2 * j] = slot;
}
// the keys (reverseEh, process.h:72-80): block = (slot, tile)
__global__ __launch_bounds__(FILT_THREADS) void long_keys_kernel(const RawHom *__restrict__ raw, const uint64_t *__restrict__ raw_base,
																  uint32_t border, LongScratch S)
{
	const uint32_t slot = blockIdx.x / LONG_TILES, base = (blockIdx.x % LONG_TILES) * FILT_MAX, tid = threadIdx.x;
	const uint32_t used = *S.next_slot < LONG_SLOTS ? *S.next_slot : LONG_SLOTS;
	if (slot >= used) return;
	const LongMeta M = S.meta[slot];
	if (base >= M.n) return;
	const RawHom *r = raw + raw_base[M.j];
	uint64_t *K = S.keys + (size_t)slot * LONG_MAX_N;
#pragma unroll
	for (uint32_t u = 0; u < FILT_MAX / FILT_THREADS; u++) {
		const uint32_t t = base + u * FILT_THREADS + tid;
		if (t < M.n) {
			const RawHom h = r[t];
			const uint32_t start = h.iref >= border ? 2u * border + 1u - h.len - h.iref : h.iref;
			K[t] = (uint64_t)start << 32 | t;
		}
	}
}
// a pass's digit counts: block = (slot, tile); `from2`: the pass reads keys2
__global__ __launch_bounds__(FILT_THREADS) void long_radix_count_kernel(LongScratch S, uint32_t shift, uint32_t mask, uint32_t from2)
{
	__shared__ uint32_t hist[RADIX_BINS];
	const uint32_t slot = blockIdx.x / LONG_TILES, tile = blockIdx.x % LONG_TILES, base = tile * FILT_MAX, tid = threadIdx.x;
	const uint32_t used = *S.next_slot < LONG_SLOTS ? *S.next_slot : LONG_SLOTS;
	if (slot >= used) return;
	const uint32_t n = S.meta[slot].n;
	if (base >= n) return;
	if (tid < RADIX_BINS) hist[tid] = 0;
	__syncthreads();
	const uint64_t *K = (from2 ? S.keys2 : S.keys) + (size_t)slot * LONG_MAX_N;
	for (uint32_t t = base + tid; t < base + FILT_MAX && t < n; t += FILT_THREADS)
		atomicAdd(&hist[(uint32_t)(K[t] >> (32u + shift)) & mask], 1u);
	__syncthreads();
	if (tid < RADIX_BINS) S.counts[((size_t)slot * LONG_TILES + tile) * RADIX_BINS + tid] = hist[tid];
}
// ... and the scatter: key i of the tile (index order: round r = i / 1024, wavefront, lane) goes behind the list's
// smaller digits, the same digit of the tiles before, and the same digit of the keys before it in this tile
__global__ __launch_bounds__(FILT_THREADS) void long_radix_scatter_kernel(LongScratch S, uint32_t shift, uint32_t mask, uint32_t from2)
{
	constexpr uint32_t ROUNDS = FILT_MAX / FILT_THREADS;
	// counts, then offsets inside the tile's keys of that digit, of a (round, wavefront)'s digits (16-bit: 32 KB, so that
	// every tile of C5's 64 lists is resident at once)
	__shared__ uint16_t cnt[ROUNDS][FILT_WAVES][RADIX_BINS];
	__shared__ uint32_t gbase[RADIX_BINS], wsum4[RADIX_BINS / 64];
	const uint32_t slot = blockIdx.x / LONG_TILES, tile = blockIdx.x % LONG_TILES, base = tile * FILT_MAX, tid = threadIdx.x;
	const uint32_t lane = tid & 63u, wave = tid >> 6;
	const uint32_t used = *S.next_slot < LONG_SLOTS ? *S.next_slot : LONG_SLOTS;
	if (slot >= used) return;
	const uint32_t n = S.meta[slot].n;
	if (base >= n) return;
	const uint32_t ntiles = (n + FILT_MAX - 1) / FILT_MAX;
	const uint64_t *K = (from2 ? S.keys2 : S.keys) + (size_t)slot * LONG_MAX_N;
	uint64_t *D = (from2 ? S.keys : S.keys2) + (size_t)slot * LONG_MAX_N;
	uint64_t key[ROUNDS];
#pragma unroll
	for (uint32_t r = 0; r < ROUNDS; r++) { // (on their way while the counts are read)
		const uint32_t i = base + r * FILT_THREADS + tid;
		key[r] = i < n ? K[i] : 0ull;
	}
	for (uint32_t t = tid; t < ROUNDS * FILT_WAVES * RADIX_BINS / 2; t += FILT_THREADS) ((uint32_t *)&cnt[0][0][0])[t] = 0;
	// where this tile's keys of digit `tid` begin: the list's smaller digits (an exclusive scan over the 256 digits: four
	// wavefronts, then their sums) and the same digit of the tiles before
	uint32_t mine = 0;
	if (tid < RADIX_BINS) {
		const uint32_t *C = S.counts + (size_t)slot * LONG_TILES * RADIX_BINS + tid;
		uint32_t before = 0, all = 0, v[LONG_TILES];
#pragma unroll
		for (uint32_t t2 = 0; t2 < LONG_TILES; t2++) v[t2] = t2 < ntiles ? C[(size_t)t2 * RADIX_BINS] : 0u; // (all in flight at once)
#pragma unroll
		for (uint32_t t2 = 0; t2 < LONG_TILES; t2++) {
			if (t2 < tile) before += v[t2];
			all += v[t2];
		}
		uint32_t incl = all;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t u = (uint32_t)__shfl_up((int)incl, d, 64);
			if ((int)lane >= d) incl += u;
		}
		mine = incl - all + before;
		if (lane == 63) wsum4[wave] = incl;
	}
	__syncthreads();
	if (tid < RADIX_BINS) {
		for (uint32_t w2 = 0; w2 < wave; w2++) mine += wsum4[w2];
		gbase[tid] = mine;
	}
	uint32_t rank[ROUNDS];
#pragma unroll
	for (uint32_t r = 0; r < ROUNDS; r++) {
		const bool valid = base + r * FILT_THREADS + tid < n;
		const uint32_t dg = (uint32_t)(key[r] >> (32u + shift)) & mask;
		unsigned long long peers = __ballot(valid);
#pragma unroll
		for (uint32_t bit = 0; bit < 8; bit++) {
			const unsigned long long m = __ballot((dg >> bit) & 1u);
			peers &= ((dg >> bit) & 1u) ? m : ~m;
		}
		rank[r] = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
		if (valid && rank[r] == 0) cnt[r][wave][dg] = (uint16_t)__popcll(peers);
	}
	__syncthreads();
	if (tid < RADIX_BINS) {
		uint32_t run = 0;
#pragma unroll
		for (uint32_t r = 0; r < ROUNDS; r++)
#pragma unroll
			for (uint32_t w2 = 0; w2 < FILT_WAVES; w2++) {
				const uint32_t u = cnt[r][w2][tid];
				cnt[r][w2][tid] = (uint16_t)run;
				run += u;
			}
	}
	__syncthreads();
#pragma unroll
	for (uint32_t r = 0; r < ROUNDS; r++) {
		if (base + r * FILT_THREADS + tid < n) {
			const uint32_t dg = (uint32_t)(key[r] >> (32u + shift)) & mask;
			D[gbase[dg] + cnt[r][wave][dg] + rank[r]] = key[r];
		}
	}
}

__global__ __launch_bounds__(FILT_THREADS) void sort_filter_long_kernel(const RawHom *__restrict__ raw,
																		 const uint64_t *__restrict__ raw_base,
																		 const uint32_t *__restrict__ raw_cnt, uint32_t j0, uint32_t border,
																		 DevHom *__restrict__ out, uint32_t *__restrict__ rng,
																		 uint32_t *__restrict__ total, uint32_t *__restrict__ flag,
																		 LongScratch S)
{
	__shared__ uint32_t wmax[FILT_WAVES], wsum[FILT_WAVES];
	__shared__ uint32_t s_tie, s_general, s_carry, s_base;
	__shared__ uint8_t s_keep[LONG_MAX_N];
	const uint32_t j = j0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	if (flag[j] != FLAG_LONG) return;
	const RawHom *r = raw + raw_base[j];
	const uint32_t n = raw_cnt[j];
	const uint32_t slot = rng[2 * j]; // (long_prepare_kernel)
	if (tid == 0) s_tie = s_general = 0;
	uint64_t *K = S.keys + (size_t)slot * LONG_MAX_N;
	uint32_t *E = S.ends + (size_t)slot * LONG_MAX_N;
	uint8_t *KP = s_keep; // bit 0 kept, bit 1 a stretch starts here — by pile position, the whole list in LDS
	// ends, equal starts, prefix maximum of the ends and the cuts, tile by tile
	if (tid == 0) s_carry = 0;
	__syncthreads();
	for (uint32_t base = 0; base < n; base += FILT_MAX) {
		const uint32_t p4 = base + tid * 4u;
		uint32_t e[4], st[4], run = 0;
#pragma unroll
		for (uint32_t u = 0; u < 4; u++) {
			const uint32_t p = p4 + u;
			e[u] = 0;
			st[u] = 0;
			if (p < n) {
				const uint64_t key = K[p];
				st[u] = (uint32_t)(key >> 32);
				if (p + 1 < n && (uint32_t)(K[p + 1] >> 32) == st[u]) s_tie = 1;
				e[u] = st[u] + r[(uint32_t)key].len;
				E[p] = e[u];
			}
			run = e[u] > run ? e[u] : run;
			e[u] = run;
		}
		uint32_t incl = run;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
			if ((int)lane >= d && t > incl) incl = t;
		}
		if (lane == 63) wmax[wave] = incl;
		__syncthreads();
		uint32_t before = (uint32_t)__shfl_up((int)incl, 1, 64);
		if (lane == 0) before = 0;
		for (uint32_t w2 = 0; w2 < wave; w2++) before = wmax[w2] > before ? wmax[w2] : before;
		before = s_carry > before ? s_carry : before;
#pragma unroll
		for (uint32_t u = 0; u < 4; u++) {
			const uint32_t p = p4 + u;
			if (p < n) {
				const uint32_t prev_max = u ? (e[u - 1] > before ? e[u - 1] : before) : before;
				KP[p] = prev_max <= st[u] ? 2u : 0u;
			}
		}
		__syncthreads();
		if (tid == FILT_THREADS - 1) {
			uint32_t m = s_carry;
			for (uint32_t w2 = 0; w2 < FILT_WAVES; w2++) m = wmax[w2] > m ? wmax[w2] : m;
			s_carry = m;
		}
		__syncthreads();
	}
	if (s_tie) {
		if (tid == 0) {
			flag[j] = FLAG_HOST;
			rng[2 * j] = 0; // (held the slot) an empty list until the host has done this query
		}
		return;
	}
	// the stretches that start at this thread's entries
	for (uint32_t a = tid; a < n; a += FILT_THREADS) {
		if (!(KP[a] & 2u)) continue;
		uint32_t b = a + 1;
		while (b < n && !(KP[b] & 2u)) b++;
		if (b - a == 1) {
			KP[a] |= 1u;
			continue;
		}
		if (b - a > SEG_MAX) {
			s_general = 1;
			continue;
		}
		uint32_t score[SEG_MAX], start_[SEG_MAX], end_[SEG_MAX];
		uint8_t pred[SEG_MAX];
		uint32_t top = 0, top_score = 0;
		const uint32_t m = b - a;
		for (uint32_t i = 0; i < m; i++) {
			start_[i] = (uint32_t)(K[a + i] >> 32);
			end_[i] = E[a + i];
			uint32_t best = 0, bk = 0xffu;
			for (uint32_t k2 = 0; k2 < i; k2++)
				if (end_[k2] <= start_[i] && score[k2] > best) {
					best = score[k2];
					bk = k2;
				}
			score[i] = best + (end_[i] - start_[i]);
			pred[i] = (uint8_t)bk;
			if (score[i] > top_score) {
				top_score = score[i];
				top = i;
			}
		}
		for (uint32_t i = top; i != 0xffu; i = pred[i]) KP[a + i] |= 1u;
	}
	__syncthreads();
	if (s_general) { // an entangled stretch beyond what a thread takes: the host does this list
		if (tid == 0) {
			flag[j] = FLAG_HOST;
			rng[2 * j] = 0;
		}
		return;
	}
	// output in pile order, tile by tile
	{
		uint32_t mine = 0;
		for (uint32_t i = tid; i < n; i += FILT_THREADS) mine += KP[i] & 1u;
		for (uint32_t d = 32; d > 0; d >>= 1) mine += (uint32_t)__shfl((int)mine, (int)(lane ^ d), 64);
		if (lane == 0) wsum[wave] = mine;
		__syncthreads();
		if (tid == 0) {
			uint32_t w = 0;
			for (uint32_t w2 = 0; w2 < FILT_WAVES; w2++) w += wsum[w2];
			const uint32_t b = w ? atomicAdd(total, w) : 0u;
			s_base = b;
			rng[2 * j] = b;
			rng[2 * j + 1] = b + w;
			flag[j] = 0;
		}
		__syncthreads();
	}
	const uint32_t obase = s_base;
	uint32_t done = 0;
	for (uint32_t p0 = 0; p0 < n; p0 += FILT_THREADS) {
		const uint32_t p = p0 + tid;
		const uint32_t k = (p < n && (KP[p] & 1u)) ? 1u : 0u;
		const uint64_t mk = __ballot(k);
		if (lane == 0) wsum[wave] = (uint32_t)__popcll(mk);
		__syncthreads();
		uint32_t off = done + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
		for (uint32_t w2 = 0; w2 < wave; w2++) off += wsum[w2];
		if (k) {
			const uint64_t key = K[p];
			const RawHom h = r[(uint32_t)key];
			out[obase + off] = DevHom{(uint32_t)(key >> 32), h.iq, h.len, h.iref >= border ? 1u : 0u};
		}
		for (uint32_t w2 = 0; w2 < FILT_WAVES; w2++) done += wsum[w2];
		__syncthreads();
	}
}

size_t long_filter_scratch_bytes()
{
	return (size_t)LONG_SLOTS * LONG_MAX_N * (8 + 8 + 4) + (size_t)LONG_SLOTS * LONG_TILES * RADIX_BINS * 4 + LONG_SLOTS * sizeof(LongMeta) + 64;
}
uint32_t long_filter_min_entries() { return FILT_MAX; }

void launch_sort_filter_long(const RawHom *raw, const uint64_t *raw_base, const uint32_t *raw_cnt, uint32_t j0, uint32_t j1,
							 uint32_t border, DevHom *out, uint32_t *rng, uint32_t *total, uint32_t *flag, void *scratch,
							 uint32_t *slot_counter, hipStream_t st)
{
	if (j1 <= j0) return;
	LongScratch S;
	S.keys = (uint64_t *)scratch;
	S.keys2 = S.keys + (size_t)LONG_SLOTS * LONG_MAX_N;
	S.ends = (uint32_t *)(S.keys2 + (size_t)LONG_SLOTS * LONG_MAX_N);
	S.counts = S.ends + (size_t)LONG_SLOTS * LONG_MAX_N;
	S.meta = (LongMeta *)(S.counts + (size_t)LONG_SLOTS * LONG_TILES * RADIX_BINS);
	S.next_slot = slot_counter;
	hipLaunchKernelGGL(long_slots_kernel, dim3((j1 - j0 + 63) / 64), dim3(64), 0, st, raw_cnt, j0, j1 - j0, rng, flag, S);
	const dim3 grid(LONG_SLOTS * LONG_TILES);
	hipLaunchKernelGGL(long_keys_kernel, grid, dim3(FILT_THREADS), 0, st, raw, raw_base, border, S);
	// projected starts lie below 2 * border + 2: four passes (an even number: the sorted keys are back in `keys`) over
	// that many bits
	uint32_t bits = 1;
	while (bits < 32 && ((2ull * border + 1ull) >> bits)) bits++;
	const uint32_t per = (bits + 3) / 4;
	for (uint32_t p = 0; p < 4; p++) {
		const uint32_t shift = p * per, mask = (1u << per) - 1u;
		hipLaunchKernelGGL(long_radix_count_kernel, grid, dim3(FILT_THREADS), 0, st, S, shift, mask, p & 1u);
		hipLaunchKernelGGL(long_radix_scatter_kernel, grid, dim3(FILT_THREADS), 0, st, S, shift, mask, p & 1u);
	}
	hipLaunchKernelGGL(sort_filter_long_kernel, dim3(j1 - j0), dim3(FILT_THREADS), 0, st, raw, raw_base, raw_cnt, j0, border, out, rng,
					   total, flag, S);
}

void launch_sort_filter(const RawHom *raw, const uint64_t *raw_base, const uint32_t *raw_cnt, uint32_t j0, uint32_t j1, uint32_t border,
						uint32_t threshold, uint32_t ref_local, DevHom *out, uint32_t *rng, uint32_t *total, uint32_t *flag,
						hipStream_t st, int variant, int long_ok)
{
	if (j1 <= j0) return;
	if (variant == 0) { // stretch by stretch, then the general kernel for what that one handed over
		hipLaunchKernelGGL(sort_filter_seg_kernel, dim3(j1 - j0), dim3(FILT_THREADS), 0, st, raw, raw_base, raw_cnt, j0, border,
						   threshold, ref_local, out, rng, total, flag, (uint32_t)long_ok);
		hipLaunchKernelGGL(sort_filter_kernel, dim3(j1 - j0), dim3(FILT_THREADS), 0, st, raw, raw_base, raw_cnt, j0, border, threshold,
						   ref_local, out, rng, total, flag, 1u);
	} else {
		hipLaunchKernelGGL(sort_filter_kernel, dim3(j1 - j0), dim3(FILT_THREADS), 0, st, raw, raw_base, raw_cnt, j0, border, threshold,
						   ref_local, out, rng, total, flag, 0u);
	}
}

} // namespace phy

// hostlogic.hpp — host-side (CPU) pieces of the path that the north star keeps
// on the host cores: suffix-array construction, the index tables derived from
// it, the anchor threshold, and the per-query sort + chain filter whose result
// depends on libstdc++'s std::sort tie order.
//
// Citations are to /root/reference.
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/phylonium_amd.h"
#include "anchor_core.h"

namespace phy {

// ───────────────────────── suffix array: SA-IS ─────────────────────────
// Induced sorting (Nong, Zhang & Chan 2009).  Stands in for
// divsufsort64 (src/esa.cxx:74); the suffix array of a string is unique, so the
// result is the same array.  Text must end in a unique smallest symbol 0.

template <class Ch>
static void sais_buckets(const Ch *s, int32_t n, int32_t K, std::vector<int32_t> &bkt, bool end)
{
	std::fill(bkt.begin(), bkt.begin() + K, 0);
	for (int32_t i = 0; i < n; i++) bkt[(size_t)s[i]]++;
	int32_t sum = 0;
	for (int32_t c = 0; c < K; c++) {
		sum += bkt[(size_t)c];
		bkt[(size_t)c] = end ? sum : sum - bkt[(size_t)c];
	}
}

template <class Ch>
static void sais_induce(const Ch *s, int32_t *SA, int32_t n, int32_t K, const std::vector<uint8_t> &stype,
						std::vector<int32_t> &bkt)
{
	sais_buckets(s, n, K, bkt, false);
	for (int32_t i = 0; i < n; i++) { // L-type, left to right
		int32_t j = SA[i] - 1;
		if (SA[i] > 0 && !stype[(size_t)j]) SA[bkt[(size_t)s[j]]++] = j;
	}
	sais_buckets(s, n, K, bkt, true);
	for (int32_t i = n - 1; i >= 0; i--) { // S-type, right to left
		int32_t j = SA[i] - 1;
		if (SA[i] > 0 && stype[(size_t)j]) SA[--bkt[(size_t)s[j]]] = j;
	}
}

template <class Ch> static void sais_main(const Ch *s, int32_t *SA, int32_t n, int32_t K)
{
	if (n == 1) {
		SA[0] = 0;
		return;
	}
	std::vector<uint8_t> stype((size_t)n);
	stype[(size_t)n - 1] = 1;
	for (int32_t i = n - 2; i >= 0; i--)
		stype[(size_t)i] = (s[i] < s[i + 1] || (s[i] == s[i + 1] && stype[(size_t)i + 1])) ? 1 : 0;
	auto is_lms = [&](int32_t i) { return i > 0 && stype[(size_t)i] && !stype[(size_t)i - 1]; };

	std::vector<int32_t> bkt((size_t)K);
	// 1. sort LMS substrings
	sais_buckets(s, n, K, bkt, true);
	std::fill(SA, SA + n, -1);
	for (int32_t i = 1; i < n; i++)
		if (is_lms(i)) SA[--bkt[(size_t)s[i]]] = i;
	sais_induce(s, SA, n, K, stype, bkt);
	int32_t n1 = 0;
	for (int32_t i = 0; i < n; i++)
		if (is_lms(SA[i])) SA[n1++] = SA[i];
	std::fill(SA + n1, SA + n, -1);
	int32_t name = 0, prev = -1;
	for (int32_t i = 0; i < n1; i++) {
		int32_t pos = SA[i];
		bool diff = false;
		if (prev < 0) {
			diff = true;
		} else {
			for (int32_t d = 0;; d++) {
				if (s[pos + d] != s[prev + d] || stype[(size_t)(pos + d)] != stype[(size_t)(prev + d)]) {
					diff = true;
					break;
				}
				if (d > 0 && (is_lms(pos + d) || is_lms(prev + d))) break;
			}
		}
		if (diff) {
			name++;
			prev = pos;
		}
		SA[n1 + (pos >> 1)] = name - 1;
	}
	for (int32_t i = n - 1, j = n - 1; i >= n1; i--)
		if (SA[i] >= 0) SA[j--] = SA[i];
	// 2. order the LMS suffixes
	int32_t *SA1 = SA, *s1 = SA + n - n1;
	if (name < n1) {
		sais_main<int32_t>(s1, SA1, n1, name);
	} else {
		for (int32_t i = 0; i < n1; i++) SA1[s1[i]] = i;
	}
	// 3. induce the full order
	sais_buckets(s, n, K, bkt, true);
	for (int32_t i = 1, j = 0; i < n; i++)
		if (is_lms(i)) s1[j++] = i;
	for (int32_t i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
	std::fill(SA + n1, SA + n, -1);
	for (int32_t i = n1 - 1; i >= 0; i--) {
		int32_t j = SA[i];
		SA[i] = -1;
		SA[--bkt[(size_t)s[j]]] = j;
	}
	sais_induce(s, SA, n, K, stype, bkt);
}

// Suffix array of the n bytes of `s` in unsigned-byte order (shorter suffix
// first), as divsufsort returns it.  n < 2^31 - 1.
static inline void suffix_array_u32(const uint8_t *s, uint32_t n, uint32_t *sa_out)
{
	if (n == 0) return;
	// dense-rank the alphabet so the bucket array stays small; 0 = sentinel
	int32_t rank[256];
	bool seen[256] = {false};
	for (uint32_t i = 0; i < n; i++) seen[s[i]] = true;
	int32_t K = 1;
	for (int c = 0; c < 256; c++) rank[c] = seen[c] ? K++ : 0;
	std::vector<uint8_t> t((size_t)n + 1);
	for (uint32_t i = 0; i < n; i++) t[i] = (uint8_t)rank[s[i]];
	t[n] = 0;
	std::vector<int32_t> SA((size_t)n + 1);
	sais_main<uint8_t>(t.data(), SA.data(), (int32_t)n + 1, K);
	for (uint32_t i = 0; i < n; i++) sa_out[i] = (uint32_t)SA[(size_t)i + 1]; // SA[0] is the sentinel
}

// ───────────────── suffix array on several host cores ─────────────────
// Bucket the suffixes by their first P symbols (counting sort over position
// chunks), then std::sort every bucket on its own by comparing the text eight
// bytes at a time.  On sequence without long repeats a comparison ends within a
// word or two and the buckets are independent, so the work spreads over the
// cores.  Long repeats make the comparisons deep: the sort counts the words it
// looks at and gives up past a budget (the caller then runs SA-IS, whose time
// does not depend on the repeats).  The array is the same either way.
//
// `s` must be followed by at least 16 zero bytes and contain no zero byte itself.
// par(ntasks, f) runs f(0..ntasks-1) on the caller's threads.

struct SuffixSortGiveUp {};

static inline uint64_t load_be64(const uint8_t *p)
{
	uint64_t v;
	memcpy(&v, p, 8);
	return __builtin_bswap64(v);
}

template <class Par> static bool suffix_array_buckets(const uint8_t *s, uint32_t n, uint32_t *sa, Par &&par, size_t nthreads)
{
	if (n < 2) {
		if (n) sa[0] = 0;
		return true;
	}
	nthreads = std::max<size_t>(1, nthreads);
	const size_t nchunk = std::min<size_t>(nthreads * 4, std::max<size_t>(1, n / 65536));
	const uint32_t per = (uint32_t)((n + nchunk - 1) / nchunk);
	auto chunk_lo = [&](size_t c) { return (uint32_t)std::min<uint64_t>((uint64_t)c * per, n); };

	// alphabet, dense ranks (0 = the padding past the end)
	std::vector<std::array<uint8_t, 256>> seen_c(nchunk);
	par(nchunk, [&](size_t c) {
		auto &seen = seen_c[c];
		seen.fill(0);
		for (uint32_t i = chunk_lo(c), e = chunk_lo(c + 1); i < e; i++) seen[s[i]] = 1;
	});
	uint32_t rank[256];
	uint32_t sigma = 1;
	for (int ch = 0; ch < 256; ch++) {
		bool any = false;
		for (size_t c = 0; c < nchunk; c++) any |= seen_c[c][(size_t)ch] != 0;
		if (any && ch == 0) return false;
		rank[ch] = any ? sigma++ : 0;
	}
	uint32_t P = 1;
	uint64_t nb = sigma;
	while (nb * sigma <= 65536 && P < 8) nb *= sigma, P++;
	const uint32_t top = (uint32_t)(nb / sigma); // weight of the first symbol
	// bucket of the suffix at i = its first P ranks, most significant first
	auto bucket_at = [&](uint32_t i) {
		uint32_t id = 0;
		for (uint32_t t = 0; t < P; t++) id = id * sigma + rank[s[i + t]]; // reads the padding past n as rank 0
		return id;
	};
	// (P <= 8 < 16 bytes of padding)

	std::vector<uint32_t> hist(nchunk * nb, 0);
	par(nchunk, [&](size_t c) {
		uint32_t lo = chunk_lo(c), hi = chunk_lo(c + 1);
		if (lo >= hi) return;
		uint32_t *h = hist.data() + c * nb;
		uint32_t id = bucket_at(hi - 1);
		h[id]++;
		for (uint32_t i = hi - 1; i-- > lo;) {
			id = id / sigma + rank[s[i]] * top;
			h[id]++;
		}
	});
	std::vector<uint32_t> start(nb + 1);
	uint32_t sum = 0, largest = 0;
	for (size_t b = 0; b < nb; b++) {
		start[b] = sum;
		for (size_t c = 0; c < nchunk; c++) {
			sum += hist[c * nb + b];
			hist[c * nb + b] = sum; // end of this chunk's share of the bucket
		}
		largest = std::max(largest, sum - start[b]);
	}
	start[nb] = sum;
	if (n > (1u << 20) && largest > n / 8) return false; // low-complexity text: one bucket would carry the sort
	par(nchunk, [&](size_t c) {
		uint32_t lo = chunk_lo(c), hi = chunk_lo(c + 1);
		if (lo >= hi) return;
		uint32_t *h = hist.data() + c * nb;
		// positions are walked downwards, so each bucket's share of the chunk is filled back to front
		uint32_t id = bucket_at(hi - 1);
		sa[--h[id]] = hi - 1;
		for (uint32_t i = hi - 1; i-- > lo;) {
			id = id / sigma + rank[s[i]] * top;
			sa[--h[id]] = i;
		}
	});

	// sort the buckets, largest work first is not needed: they are many and small
	std::atomic<bool> give_up{false};
	std::atomic<uint64_t> deep_words{0};
	const uint64_t budget = 64ull * n + (1u << 20); // words of 8 bytes; SA-IS costs about as much as 100 per character
	// tasks of about 64k suffixes
	std::vector<uint32_t> task_lo;
	{
		uint32_t acc = 0;
		task_lo.push_back(0);
		for (size_t b = 0; b < nb; b++) {
			acc += start[b + 1] - start[b];
			if (acc >= 65536) {
				task_lo.push_back((uint32_t)b + 1);
				acc = 0;
			}
		}
		if (task_lo.back() != nb) task_lo.push_back((uint32_t)nb);
	}
	par(task_lo.size() - 1, [&](size_t t) {
		uint64_t local = 0;
		auto less = [&](uint32_t a, uint32_t b) {
			if (a == b) return false;
			const uint8_t *x = s + a + P, *y = s + b + P;
			for (uint32_t d = 0;; d += 8) {
				uint64_t u = load_be64(x + d), v = load_be64(y + d);
				if (u != v) return u < v;
				if ((d & 0xff) == 0xf8) { // another 32 words alike
					local += 32;
					if (local >= 65536) {
						uint64_t g = deep_words.fetch_add(local) + local;
						local = 0;
						if (g > budget) throw SuffixSortGiveUp();
					}
					if (give_up.load(std::memory_order_relaxed)) throw SuffixSortGiveUp();
				}
			}
		};
		try {
			for (uint32_t b = task_lo[t]; b < task_lo[t + 1]; b++) {
				if (give_up.load(std::memory_order_relaxed)) return;
				uint32_t lo = start[b], hi = start[b + 1];
				if (hi - lo > 1) std::sort(sa + lo, sa + hi, less);
			}
			if (local) deep_words.fetch_add(local);
		} catch (const SuffixSortGiveUp &) {
			give_up.store(true);
		}
	});
	return !give_up.load();
}

// par() for callers without a thread pool of their own
struct ThreadFan {
	size_t nthreads;
	void operator()(size_t ntasks, const std::function<void(size_t)> &f) const
	{
		size_t nt = std::min(nthreads, ntasks);
		if (nt <= 1) {
			for (size_t i = 0; i < ntasks; i++) f(i);
			return;
		}
		std::atomic<size_t> next{0};
		std::vector<std::thread> th;
		for (size_t t = 0; t < nt; t++)
			th.emplace_back([&] {
				for (size_t i; (i = next.fetch_add(1)) < ntasks;) f(i);
			});
		for (auto &t : th) t.join();
	}
};

// Suffix array on `nthreads` cores when the text allows, SA-IS otherwise.  `s` needs 16 zero bytes after it.
template <class Par> static inline void suffix_array_u32_par(const uint8_t *s, uint32_t n, uint32_t *sa_out, Par &&par, size_t nthreads)
{
	if (nthreads > 1 && n >= (1u << 16) && suffix_array_buckets(s, n, sa_out, par, nthreads)) return;
	suffix_array_u32(s, n, sa_out);
}

// LCP[r] = lcp(suffix SA[r-1], suffix SA[r]) for r in 1..n-1, LCP[0] = LCP[n] = 0
// (Kasai et al.; the reference's init_LCP, src/esa.cxx:305-347, computes the same
// values with the Φ variant and stores -1 at both ends).
static inline void lcp_kasai(const uint8_t *s, uint32_t n, const uint32_t *sa, uint32_t *lcp)
{
	std::vector<uint32_t> rnk((size_t)n);
	for (uint32_t r = 0; r < n; r++) rnk[sa[r]] = r;
	uint32_t h = 0;
	lcp[0] = 0;
	lcp[n] = 0;
	for (uint32_t i = 0; i < n; i++) {
		uint32_t r = rnk[i];
		if (r == 0) {
			h = 0;
			continue;
		}
		uint32_t j = sa[r - 1];
		while (i + h < n && j + h < n && s[i + h] == s[j + h]) h++;
		lcp[r] = h;
		if (h) h--;
	}
}

// Bucket table: T[c] = number of suffixes of s lexicographically smaller than
// the k-mer with code c (A<C<G<T, 2 bits each), c in [0,4^k]; T[4^k] = n.
// Suffixes that hit a non-ACGT byte ('!', '#') or the end of s inside their
// first k bytes sort before every k-mer sharing their ACGT prefix, because
// those bytes are < 'A'.
static inline void kmer_table(const uint8_t *s, uint32_t n, uint32_t k, std::vector<uint32_t> &T)
{
	size_t buckets = (size_t)1 << (2 * k);
	T.assign(buckets + 1, 0);
	uint32_t code = 0, run = 0;
	for (uint32_t ii = n; ii-- > 0;) {
		uint32_t v = nuc_code(s[ii]);
		if (v > 3) {
			run = 0;
			v = 0;
		} else if (run < k) {
			run++;
		}
		code = (v << (2 * (k - 1))) | (code >> 2);
		if (run >= k) {
			T[(size_t)code + 1]++;
		} else {
			uint32_t low = 2 * (k - run);
			uint32_t b = (low >= 32) ? 0u : (code >> low) << low;
			T[b]++;
		}
	}
	uint32_t sum = 0;
	for (size_t c = 0; c <= buckets; c++) {
		sum += T[c];
		T[c] = sum;
	}
}

// One 16-byte record per rank (anchor_core.h: sax_record); +4 zero records of pad.
static inline void build_sax(const uint8_t *s, uint32_t n, const uint32_t *sa, const uint32_t *lcp,
							 std::vector<U4> &out)
{
	out.assign((size_t)n + 4, U4{0, 0, 0, 0});
	for (uint32_t r = 0; r < n; r++) out[r] = sax_record(s, sa[r], lcp[r], lcp[r + 1]);
}

// Slot table for the emulation on the host (the product builds it on the device, abi_reference.hip:
// build_slots_kernel, with the same slot_make).
static inline void build_slots(const std::vector<uint32_t> &T, const std::vector<U4> &sax, uint32_t n, uint32_t k,
							   std::vector<U4> &out)
{
	const size_t codes = (size_t)1 << (2 * k);
	out.assign(codes + 8, U4{0, 0, 0, 0}); // (+8: a trip's load batch reads up to 64 bytes from a slot's address)
	for (size_t c = 0; c < codes; c++) out[c] = slot_make(c, k, T[c], T[c + 1], n, [&](uint32_t r) { return sax[r]; });
}

// Does the reference's 6-mer interval cache hold an entry that claims more than its key shares with S?
// esa::init_cache_dfs (src/esa.cxx:114-201) walks the virtual suffix tree one nucleotide at a time; when a
// child interval is deeper than expected (all suffixes that start with the prefix so far go on identically)
// it fast-forwards along that common stretch — and if it meets a non-ACGT byte there (esa.cxx:174-199) it
// fills every key below the prefix with the child interval and its FULL lcp value.  get_match_cached then
// reports that many matching characters for any query carrying such a key, although the query has a
// nucleotide where S has '!': the reference's answer is longer than the longest match.  It takes a
// nucleotide string of at most 4 characters that occurs at least twice in S and ONLY in front of the same
// contig join — impossible beyond a few kbp of sequence, so the product computes the true longest match
// and this walk (the same DFS over the suffix array: at most 4^5 nodes) says when the reference would not.
// One over-deep cache entry: every 6-mer key that starts with the `k` nucleotides `prefix` (2 bits each, first
// nucleotide in the highest of the 2k bits) holds the interval of ranks [lo, hi) with depth `depth` > k — what its
// suffixes share: the k nucleotides, then a non-nucleotide byte within the next depth - k.
struct CacheQuirk {
	uint32_t prefix, k, depth, lo, hi;
};
static inline std::vector<CacheQuirk> esa_cache_quirks(const uint8_t *S, uint32_t n, const uint32_t *SA)
{
	std::vector<CacheQuirk> found;
	const uint32_t CACHE_LENGTH = 6;
	auto at = [&](uint32_t r, uint32_t off) -> int { // byte at offset off of the suffix of rank r; -1 past the end
		const uint64_t p = (uint64_t)SA[r] + off;
		return p < n ? (int)S[p] : -1;
	};
	// the ranks in [lo, hi) whose suffix has byte c at offset pos (they share their first pos bytes: sorted there)
	auto range = [&](uint32_t lo, uint32_t hi, uint32_t pos, int c, uint32_t *a, uint32_t *b) {
		uint32_t l = lo, h = hi;
		while (l < h) {
			const uint32_t m = l + ((h - l) >> 1);
			if (at(m, pos) < c) l = m + 1;
			else h = m;
		}
		*a = l;
		h = hi;
		while (l < h) {
			const uint32_t m = l + ((h - l) >> 1);
			if (at(m, pos) <= c) l = m + 1;
			else h = m;
		}
		*b = l;
	};
	struct Node {
		uint32_t pos, lo, hi, prefix; // prefix: the pos nucleotides on the way here, 2 bits each
	};
	std::vector<Node> stack;
	stack.push_back(Node{0, 0, n, 0});
	while (!stack.empty()) {
		const Node nd = stack.back();
		stack.pop_back();
		if (nd.pos >= CACHE_LENGTH) continue;
		for (int code = 0; code < 4; code++) {
			uint32_t a, b;
			range(nd.lo, nd.hi, nd.pos, "ACGT"[code], &a, &b);
			if (b - a < 2) continue; // not found, or a singleton: filled as it is (esa.cxx:136-148)
			uint32_t l = nd.pos + 1; // the child's lcp value: what its first and last suffix share
			while (l <= CACHE_LENGTH && at(a, l) >= 0 && at(a, l) == at(b - 1, l)) l++;
			uint32_t prefix = (nd.prefix << 2) | (uint32_t)code;
			if (l <= nd.pos + 1) {
				stack.push_back(Node{nd.pos + 1, a, b, prefix}); // the usual case, esa.cxx:150-155
				continue;
			}
			if (l >= CACHE_LENGTH) continue; // deeper than the cache: filled with the parent, esa.cxx:159-163
			// fast forward (esa.cxx:178-192): the nucleotides the interval shares go into the key; a byte that is
			// none ends it, and the keys below str[0..k) get this interval, whose depth l is beyond k (esa.cxx:195-196)
			uint32_t k = nd.pos + 1;
			bool non_acgt = false;
			for (; k < l; k++) {
				const uint32_t c = nuc_code((uint8_t)at(a, k));
				if (c > 3) {
					non_acgt = true;
					break;
				}
				prefix = (prefix << 2) | c;
			}
			if (non_acgt) found.push_back(CacheQuirk{prefix, k, l, a, b});
			else stack.push_back(Node{l, a, b, prefix});
		}
	}
	return found;
}
static inline bool esa_cache_quirk(const uint8_t *S, uint32_t n, const uint32_t *SA) { return !esa_cache_quirks(S, n, SA).empty(); }

static inline uint32_t choose_k(uint32_t n)
{
	// smallest k with 4^k >= n: at most one suffix per bucket on average, so that
	// nearly every bucket has <= 2 members and takes the four-candidate fast path.
	// Capped at 14 (the k-mer must fit the 16-byte query window; table = 4^k+1 u32).
	uint32_t k = 1;
	while (k < 14 && ((uint64_t)1 << (2 * k)) < (uint64_t)n) k++;
	return k;
}

// ───────────────────────── anchor threshold ─────────────────────────
// src/process.cxx:77-161 (Haubold et al. 2009 shustring distribution).

static inline size_t binomial_coefficient(size_t n, size_t k)
{
	if (n <= 0 || k > n) return 0;
	if (k == 0 || k == n) return 1;
	if (k > n - k) k = n - k;
	size_t res = 1;
	for (size_t i = 1; i <= k; i++) {
		res *= n - k + i;
		res /= i;
	}
	return res;
}

static inline double shuprop(size_t x, double p, size_t l)
{
	double xx = (double)x, ll = (double)l, s = 0.0;
	for (size_t k = 0; k <= x; k++) {
		double kk = (double)k;
		double t = pow(p, kk) * pow(0.5 - p, xx - kk);
		s += pow(2, xx) * (t * pow(1 - t, ll)) * (double)binomial_coefficient(x, k);
		if (s >= 1.0) {
			s = 1.0;
			break;
		}
	}
	return s;
}

static inline size_t min_anchor_length(double p, double g, size_t l)
{
	size_t x = 1;
	while (shuprop(x, g / 2, l) < 1 - p) x++;
	return x;
}

// src/sequence.cxx:152-165
static inline double gc_content(const uint8_t *s, size_t n)
{
	size_t gc = 0;
	for (size_t i = 0; i < n; i++)
		if ((s[i] & 'G' & 'C') == ('G' & 'C')) gc++;
	return (double)gc / (double)n;
}

// src/sequence.cxx:73-103
static inline void revcomp(const uint8_t *in, size_t n, uint8_t *out)
{
	for (size_t k = 0; k < n; k++) {
		uint8_t c = in[n - k - 1];
		out[k] = (c < 'A') ? c : (uint8_t)(c ^ ((c & 2) ? 4 : 21));
	}
}

// ───────────────────────── homologies on the host ─────────────────────────

// homology::reverseEh, src/process.h:72-80. `border` = L = ref.size()/2.
static inline phylo_homology project_homology(const RawHom &r, uint64_t border)
{
	phylo_homology h;
	h.index_reference = r.iref;
	h.index_reference_projected = r.iref;
	h.index_query = r.iq;
	h.length = r.len;
	h.direction = 0;
	h._pad = 0;
	if (h.index_reference >= border) {
		h.index_reference_projected = 2 * border + 1 - h.length - h.index_reference;
		h.direction = 1;
	}
	return h;
}

// filter_overlaps_max, src/process.cxx:354-401, in O(n log n).
// The reference's O(n²) scan picks, for each i, the smallest k with the largest
// score among {k < i : end[k] <= start[i]}.  Because the pile is sorted by start
// and every length is positive, end[k] <= start[i] already implies k < i, so the
// candidates are a prefix of the pile ordered by end; a running (max score,
// smallest index) over that order gives the same predecessor.
static inline void filter_overlaps_max(std::vector<phylo_homology> &pile)
{
	size_t n = pile.size();
	if (n < 2) return;
	std::vector<uint32_t> by_end(n);
	for (size_t i = 0; i < n; i++) by_end[i] = (uint32_t)i;
	std::stable_sort(by_end.begin(), by_end.end(), [&](uint32_t a, uint32_t b) {
		return pile[a].index_reference_projected + pile[a].length <
			   pile[b].index_reference_projected + pile[b].length;
	});
	std::vector<int64_t> score(n), pred(n);
	int64_t best = 0, best_k = -1;
	size_t p = 0;
	for (size_t i = 0; i < n; i++) {
		uint64_t start_i = pile[i].index_reference_projected;
		while (p < n) {
			uint32_t k = by_end[p];
			if (pile[k].index_reference_projected + pile[k].length > start_i) break;
			if (score[k] > best || (score[k] == best && (int64_t)k < best_k)) {
				best = score[k];
				best_k = k;
			}
			p++;
		}
		pred[i] = best_k;
		score[i] = (best_k >= 0 ? score[(size_t)best_k] : 0) + (int64_t)pile[i].length;
	}
	// std::max_element over (0, score[0..n)): first maximum
	int64_t top = 0, idx = -1;
	for (size_t i = 0; i < n; i++)
		if (score[i] > top) {
			top = score[i];
			idx = (int64_t)i;
		}
	std::vector<uint8_t> keep(n, 0);
	while (idx >= 0) {
		keep[(size_t)idx] = 1;
		idx = pred[(size_t)idx];
	}
	size_t w = 0;
	for (size_t r = 0; r < n; r++)
		if (keep[r]) pile[w++] = pile[r];
	pile.resize(w);
}

// src/process.cxx:438-443 — same std::sort call on the same input order.
static inline void sort_and_filter(std::vector<phylo_homology> &hv)
{
	std::sort(hv.begin(), hv.end(), [](const phylo_homology &a, const phylo_homology &b) {
		return a.index_reference_projected < b.index_reference_projected;
	});
	filter_overlaps_max(hv);
}

// The same sort + filter on packed keys, for the per-query lists of phase A.
// get(i, &start, &len) describes entry i.  The order by projected start is unique
// unless two entries share a start; std::sort is not stable, so with such a tie the
// reference's order is whatever libstdc++'s introsort leaves — the function then
// returns false and the caller runs sort_and_filter on the structs.  Otherwise
// `kept` receives the surviving entries (indices into the input) in pile order.
struct SortFilterScratch {
	std::vector<uint64_t> by_start, by_end;
	std::vector<uint32_t> st, len;
	std::vector<int64_t> score;
	std::vector<int32_t> pred;
};

template <class Get>
static inline bool sort_filter_order(size_t n, Get get, SortFilterScratch &w, std::vector<uint32_t> &kept)
{
	kept.clear();
	if (n >= 0x7fffffffu) return false;
	w.by_start.resize(n);
	for (size_t i = 0; i < n; i++) {
		uint64_t s, l;
		get(i, &s, &l);
		if ((s + l) >> 32) return false;
		w.by_start[i] = s << 32 | (uint64_t)i;
	}
	std::sort(w.by_start.begin(), w.by_start.end());
	for (size_t p = 1; p < n; p++)
		if ((w.by_start[p] >> 32) == (w.by_start[p - 1] >> 32)) return false;
	// pile order p: entry by_start[p] & 0xffffffff
	w.st.resize(n);
	w.len.resize(n);
	w.by_end.resize(n);
	for (size_t p = 0; p < n; p++) {
		uint64_t s, l;
		get((size_t)(w.by_start[p] & 0xffffffffu), &s, &l);
		w.st[p] = (uint32_t)s;
		w.len[p] = (uint32_t)l;
		w.by_end[p] = (s + l) << 32 | (uint64_t)p; // ties by pile index = stable
	}
	std::sort(w.by_end.begin(), w.by_end.end());
	w.score.resize(n);
	w.pred.resize(n);
	int64_t best = 0;
	int32_t best_k = -1;
	size_t q = 0;
	for (size_t i = 0; i < n; i++) {
		const uint64_t start_i = w.st[i];
		while (q < n && (w.by_end[q] >> 32) <= start_i) {
			const int32_t k = (int32_t)(w.by_end[q] & 0xffffffffu);
			if (w.score[(size_t)k] > best || (w.score[(size_t)k] == best && k < best_k)) {
				best = w.score[(size_t)k];
				best_k = k;
			}
			q++;
		}
		w.pred[i] = best_k;
		w.score[i] = (best_k >= 0 ? w.score[(size_t)best_k] : 0) + (int64_t)w.len[i];
	}
	int64_t top = 0;
	int32_t idx = -1;
	for (size_t i = 0; i < n; i++)
		if (w.score[i] > top) {
			top = w.score[i];
			idx = (int32_t)i;
		}
	if (n == 1) idx = 0; // a pile of one is returned as it is (process.cxx:358)
	for (; idx >= 0; idx = w.pred[(size_t)idx]) kept.push_back((uint32_t)(w.by_start[(size_t)idx] & 0xffffffffu));
	std::reverse(kept.begin(), kept.end());
	return true;
}

// homology::trim, src/process.h:119-143
static inline phylo_homology trim_homology(const phylo_homology &h, uint64_t s, uint64_t e)
{
	if (e <= s) return h;
	phylo_homology t = h;
	uint64_t hs = h.index_reference_projected, he = hs + h.length;
	uint64_t offset = (s > hs && s < he) ? s - hs : 0;
	uint64_t drift = (he > e && e > hs) ? he - e : 0;
	t.index_reference_projected += offset;
	if (h.direction == 0) {
		t.index_reference += offset;
		t.index_query += offset;
	} else {
		t.index_reference += drift;
		t.index_query += drift;
	}
	t.length = h.length - offset - drift;
	return t;
}

// complete_delete, src/process.cxx:725-776
static inline std::vector<std::vector<phylo_homology>>
complete_delete(const std::vector<std::vector<phylo_homology>> &H)
{
	size_t n = H.size();
	std::vector<std::vector<phylo_homology>> core(n);
	std::vector<size_t> front(n, 0);
	for (;;) {
		bool open = true;
		for (size_t g = 0; g < n; g++)
			if (front[g] >= H[g].size()) open = false;
		if (!open || n == 0) break;
		uint64_t cs = 0, ce = 0;
		size_t arg = 0;
		for (size_t g = 0; g < n; g++) {
			const phylo_homology &h = H[g][front[g]];
			uint64_t s = h.index_reference_projected, e = s + h.length;
			if (g == 0 || s > cs) cs = s;
			if (g == 0 || e < ce) {
				ce = e;
				arg = g;
			}
		}
		if (cs < ce)
			for (size_t g = 0; g < n; g++) core[g].push_back(trim_homology(H[g][front[g]], cs, ce));
		front[arg]++;
	}
	return core;
}

// evo_model::estimate_*, src/evo_model.cxx:100-131
static inline double estimate_raw(uint64_t s, uint64_t h, bool zero_on_error)
{
	if (h == 0) return zero_on_error ? 0.0 : NAN;
	return s / (double)h;
}
static inline double estimate_ani(uint64_t s, uint64_t h, bool zero_on_error)
{
	if (h == 0) return zero_on_error ? 0.0 : NAN;
	return (1.0 - s / (double)h) * 100;
}
static inline double estimate_jc(uint64_t s, uint64_t h, bool zero_on_error)
{
	double d = estimate_raw(s, h, zero_on_error);
	d = -0.75 * log(1.0 - (4.0 / 3.0) * d);
	return d <= 0.0 ? 0.0 : d;
}

// ───────────────────────── phase-A work layout ─────────────────────────

struct ChunkPlan {
	uint32_t C = 0, cap = 0, nchunks = 0;
	uint64_t anchor_slots = 0;         // size of the speculative log array
	std::vector<uint32_t> qchunk0;     // [nq+1]
	std::vector<uint32_t> qanc0;       // [nq] first log slot of each query
	std::vector<uint32_t> chunk_query; // [nchunks]
	std::vector<uint32_t> items;       // [nchunks] work order (see plan_chunks)
};

static const uint32_t CHUNK_MIN = 1536, CHUNK_MAX = 10240;
static const uint32_t ITEM_RUN = 256; // consecutive chunks of one query in the work order (= lanes of a block)

// Chunks per query and the order they are worked on.
// `lanes` = chains the device keeps resident (CUs x blocks per CU x 256).  A chain
// is a dependent sequence, so a chunk's latency is proportional to its length
// whatever the load, and when the work queue runs dry the device idles while the
// last chunks finish.  Measured on MI355X (C3, C4, 29 x 5 Mbp, 1/8 of C3): one
// chunk per lane beats two or three shorter ones (5056 vs 2496 on C3: 5.4 vs
// 6.0 ms), many rounds of long chunks are as good as one (C4: 4096..20096 within
// 4 %), and below one chunk per lane 1536 is the best length (shorter ones only
// add bridge and fold work).  So: the fewest rounds with C <= CHUNK_MAX, the chunk
// count 3 % under a whole number of rounds (every query also ends in a partial
// chunk), and never below CHUNK_MIN.  (A query's tail in shorter chunks, queued
// behind the long ones for the lanes that finish early, was measured in rounds 1
// and 3 and lost both times: DESIGN.md, history.)
// `quantum` (0: not used) = lanes of one block on every CU: below one chunk per lane the chunk count aims at a whole
// number of blocks per CU — 3.2 blocks per CU means a fifth of the CUs works a third longer than the rest (64 x 5 Mbp:
// 2.46 -> 2.24 ms per pass with 3.0).
static inline ChunkPlan plan_chunks(const std::vector<uint32_t> &qlen, uint32_t threshold, uint32_t forced_C,
									uint32_t lanes = 256u * 4u * 256u, uint32_t quantum = 0)
{
	ChunkPlan P;
	uint64_t total = 0;
	for (uint32_t l : qlen) total += l;
	uint32_t C = forced_C;
	if (C == 0) {
		const uint64_t L = lanes ? lanes : 1;
		if (total <= L * CHUNK_MIN) {
			C = CHUNK_MIN;
			if (quantum && total > (uint64_t)quantum * CHUNK_MIN) {
				const uint64_t k = std::min<uint64_t>(total / ((uint64_t)quantum * CHUNK_MIN), L / quantum);
				const uint64_t want = (uint64_t)((double)(k * quantum) * 0.97);
				const uint64_t c = ((total + want - 1) / want + 63) / 64 * 64;
				C = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(c, CHUNK_MIN), CHUNK_MAX);
			}
		} else {
			const uint64_t rounds = (total + L * CHUNK_MAX - 1) / (L * CHUNK_MAX);
			// every query also ends in a partial chunk: aim 3 % under the whole number
			const uint64_t want = (uint64_t)((double)(rounds * L) * 0.97);
			uint64_t c = (total + want - 1) / want;
			c = (c + 63) / 64 * 64;
			C = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(c, CHUNK_MIN), CHUNK_MAX);
		}
	}
	C = (C + 63) / 64 * 64;
	while (C <= 2 * threshold + 32) C += 64; // chunk starts must be lucky-ineligible from (0,0,0)
	P.C = C;
	P.cap = (C / (threshold + 1) + 2 + 3u) & ~3u; // (a multiple of four: a chunk's log starts on a 64-byte boundary)
	size_t nq = qlen.size();
	P.qchunk0.resize(nq + 1);
	P.qanc0.resize(nq);
	uint32_t acc = 0, maxb = 0;
	uint64_t slots = 0;
	for (size_t j = 0; j < nq; j++) {
		P.qchunk0[j] = acc;
		const uint32_t nb = (uint32_t)(((uint64_t)qlen[j] + C - 1) / C);
		if (slots + (uint64_t)nb * P.cap >= 0xffffffffull) return ChunkPlan(); // log slots are 32-bit indices
		P.qanc0[j] = (uint32_t)slots;
		slots += (uint64_t)nb * P.cap;
		acc += nb;
		maxb = std::max(maxb, nb);
	}
	P.anchor_slots = slots;
	P.qchunk0[nq] = acc;
	P.nchunks = acc;
	P.chunk_query.resize(acc);
	for (size_t j = 0; j < nq; j++)
		for (uint32_t c = P.qchunk0[j]; c < P.qchunk0[j + 1]; c++) P.chunk_query[c] = (uint32_t)j;
	// Work order: runs of ITEM_RUN consecutive chunks of one query, the runs dealt round-robin
	// over the queries.  A block's 256 lanes then hold chunks of the same genome, which behave
	// alike (divergence decides how many steps a chunk has and how long its matches run), so
	// the wavefronts stay in the same phases: 8 % faster on C3 than one chunk per query in turn
	// (4.9 vs 5.3 ms), and better than whole queries one after the other when the queue is
	// several rounds long (C4: 18.1 vs 18.8 ms).
	P.items.reserve(acc);
	for (uint32_t r = 0; r < maxb; r += ITEM_RUN)
		for (size_t j = 0; j < nq; j++)
			for (uint32_t e = 0; e < ITEM_RUN; e++)
				if (P.qchunk0[j] + r + e < P.qchunk0[j + 1]) P.items.push_back(P.qchunk0[j] + r + e);
	return P;
}

} // namespace phy

// oracle.cpp — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
//
// A CPU restatement of phylonium's anchor + pairwise-compare hot path, used as
// the parity checker for the HIP implementation.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
// shared object (oracle/liboracle.so).
//
// Parity status: PINNED against (a) the reference's own unit-test literals
// (test/Tprocess.cxx:19-123, test/Tsequence.cxx:5-42) and (b) the known answers
// SURVEY.md §8c recorded from the compiled reference on `simf` inputs; the simf
// inputs are regenerated here by oracle/_ref/simf, built from
// /root/reference/test/simf.cxx (the only reference program that compiles from
// its own sources: src/ and libs/ need an autoconf-generated config.h and the
// un-vendored libdivsufsort64, so they are treated as unbuildable).  The
// reverse-strand (`revseqcmp`) and multi-contig ('!') branches have no
// reference-side known answer; they are pinned by restatement only.
//
// Language: C++17 rather than plain C because the reference sorts homologies
// with libstdc++'s (unstable) std::sort (src/process.cxx:438) and the survivor
// of filter_overlaps_max depends on the tie order; calling the same std::sort
// is the only exact restatement.  The PHYLIP text likewise goes through the
// same iostream formatting calls (src/io.cxx:147-162).
//
// Third-party arithmetic: libdivsufsort64 (unpinned, configure.ac:43-48) is
// called once (src/esa.cxx:74) to build the suffix array of S.  The suffix
// array of a string is unique, so any correct sorter gives the same array; the
// one here is a key sort + comparison fallback, independent of the product's.
//
// Every function cites the reference lines it follows.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <numeric>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t sidx; // saidx64_t in the reference (src/esa.h:16,31-40)

// ───────────────────────── byte kernels ─────────────────────────

// libs/seqcmp.c:13-28 — number of positions where the two strings differ.
static size_t seqcmp_port(const char *a, const char *b, size_t n)
{
	size_t s = 0;
	for (size_t i = 0; i < n; i++) s += (a[i] != b[i]);
	return s;
}

// libs/revseqcmp.h:19-23 — complement test on bits 1..2 of the xor.
static inline int complement_p(char c, char d)
{
	return (((int)c ^ (int)d) & 6) == 4;
}

// libs/revseqcmp.c:15-30 — a[i] against b[n-1-i] under the complement test.
static size_t revseqcmp_port(const char *a, const char *b, size_t n)
{
	size_t s = 0;
	for (size_t i = 0; i < n; i++) s += !complement_p(a[i], b[n - 1 - i]);
	return s;
}

// ── the reference's x86 SIMD bodies and its load-time resolver ──
// The reference binds seqcmp/revseqcmp once, by GNU ifunc, to the widest variant the CPU
// offers (libs/seqcmp.c:32-60, libs/revseqcmp.c:34-63).  bench.py's cpu_baseline times the
// variant that resolver would pick on the box it runs on, so the restatements below follow
// the vector bodies, not only the byte loops above.  Checked against the byte loops on the
// B0 length/offset sweep (tests/test_oracle_self.py).
#if defined(__x86_64__)
#include <immintrin.h>

// libs/seqcmp_avx2.c:23-58 — 32-byte lanes, two per trip (the vector count is rounded down
// to an even number), equal bytes counted through movemask + popcount; byte loop for the rest.
__attribute__((target("avx2,popcnt"))) static size_t seqcmp_avx2_port(const char *a, const char *b, size_t n)
{
	const size_t W = 32;
	const size_t nvec = (n / W) & ~(size_t)1;
	size_t same = 0;
	for (size_t v = 0; v < nvec; v += 2) {
		for (size_t h = 0; h < 2; h++) {
			__m256i x, y;
			memcpy(&x, a + (v + h) * W, W);
			memcpy(&y, b + (v + h) * W, W);
			same += (size_t)__builtin_popcount((unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, y)));
		}
	}
	size_t s = nvec * W - same;
	for (size_t i = nvec * W; i < n; i++) s += (a[i] != b[i]);
	return s;
}

// libs/seqcmp_avx512.c:14-46 — the same on 32-byte lanes with the AVX-512BW/VL mask compare
// (one vector per trip, no rounding to pairs).
__attribute__((target("avx512bw,avx512vl,popcnt"))) static size_t seqcmp_avx512_port(const char *a, const char *b, size_t n)
{
	const size_t W = 32;
	const size_t nvec = n / W;
	size_t same = 0;
	for (size_t v = 0; v < nvec; v++) {
		__m256i x, y;
		memcpy(&x, a + v * W, W);
		memcpy(&y, b + v * W, W);
		same += (size_t)__builtin_popcount((unsigned)_mm256_cmpeq_epi8_mask(x, y));
	}
	size_t s = nvec * W - same;
	for (size_t i = nvec * W; i < n; i++) s += (a[i] != b[i]);
	return s;
}

// libs/revseqcmp_avx2.c:24-46 — vector v of `a` against the 32 bytes of `b` that end at
// n - 32 v, byte-reversed: a per-128-bit-half shuffle reverses inside the halves and the halves
// of `a` are swapped instead of those of `b` (the count does not care which side is permuted);
// complements are the bytes whose xor has bits 1..2 equal to 100b.
__attribute__((target("avx2,popcnt"))) static size_t revseqcmp_avx2_port(const char *a, const char *b, size_t n)
{
	const size_t W = 32;
	const size_t nvec = n / W;
	const __m256i rev = _mm256_set_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10,
										11, 12, 13, 14, 15);
	const __m256i six = _mm256_set1_epi8(6), four = _mm256_set1_epi8(4);
	size_t s = nvec * W;
	for (size_t v = 0; v < nvec; v++) {
		__m256i x, y;
		memcpy(&x, a + v * W, W);
		memcpy(&y, b + (n - (v + 1) * W), W);
		const __m256i yr = _mm256_shuffle_epi8(y, rev);
		const __m256i xs = _mm256_permute2x128_si256(x, x, 1);
		const __m256i hit = _mm256_cmpeq_epi8(_mm256_and_si256(_mm256_xor_si256(xs, yr), six), four);
		s -= (size_t)__builtin_popcount((unsigned)_mm256_movemask_epi8(hit));
	}
	for (size_t i = nvec * W; i < n; i++) s += !complement_p(a[i], b[n - 1 - i]);
	return s;
}
#endif

typedef size_t (*bytecmp_fn)(const char *, const char *, size_t);
static const char *g_seqcmp_name = "generic", *g_revseqcmp_name = "generic";

// libs/seqcmp.c:32-55 — widest first: AVX-512BW+VL, then AVX2, (SSE2 not restated: every
// x86-64 CPU with popcnt that lacks AVX2 falls to the byte loop here), always with popcnt.
static bytecmp_fn pick_seqcmp()
{
#if defined(__x86_64__)
	__builtin_cpu_init();
	if (__builtin_cpu_supports("popcnt") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl")) {
		g_seqcmp_name = "avx512";
		return seqcmp_avx512_port;
	}
	if (__builtin_cpu_supports("popcnt") && __builtin_cpu_supports("avx2")) {
		g_seqcmp_name = "avx2";
		return seqcmp_avx2_port;
	}
#endif
	return seqcmp_port;
}
// libs/revseqcmp.c:34-51 — AVX2, else (SSSE3 not restated) the byte loop.
static bytecmp_fn pick_revseqcmp()
{
#if defined(__x86_64__)
	__builtin_cpu_init();
	if (__builtin_cpu_supports("popcnt") && __builtin_cpu_supports("avx2")) {
		g_revseqcmp_name = "avx2";
		return revseqcmp_avx2_port;
	}
#endif
	return revseqcmp_port;
}
static bytecmp_fn g_seqcmp = pick_seqcmp(), g_revseqcmp = pick_revseqcmp();
static bool g_force_generic = false; // tests: run the path through the byte loops

// ───────────────────────── sequence helpers ─────────────────────────

// src/sequence.cxx:73-103 — reverse complement; bytes below 'A' pass through.
static std::string revcomp_port(const std::string &base)
{
	size_t len = base.size();
	std::string out(len, '\0');
	for (size_t k = 0; k < len; k++) {
		char c = base[len - k - 1];
		out[k] = (c < 'A') ? c : (char)(c ^ ((c & 2) ? 4 : 21));
	}
	return out;
}

// src/sequence.cxx:109-146 — keep ACGTacgt, upper-cased.
static std::string filter_nucl_port(const std::string &base)
{
	std::string out;
	out.reserve(base.size());
	for (unsigned char c : base) {
		switch (c) {
			case 'A': case 'a': out += 'A'; break;
			case 'C': case 'c': out += 'C'; break;
			case 'G': case 'g': out += 'G'; break;
			case 'T': case 't': out += 'T'; break;
			default: break;
		}
	}
	return out;
}

// src/sequence.cxx:152-165
static double gc_content_port(const char *s, size_t n)
{
	size_t gc = 0;
	for (size_t i = 0; i < n; i++) {
		char m = s[i] & 'G' & 'C';
		if (m == ('G' & 'C')) gc++;
	}
	return (double)gc / (double)n;
}

// ───────────────────────── anchor threshold ─────────────────────────

// src/process.cxx:103-125
static size_t binom_port(size_t n, size_t k)
{
	if (n <= 0 || k > n) return 0;
	if (k == 0 || k == n) return 1;
	if (k > n - k) k = n - k;
	size_t r = 1;
	for (size_t i = 1; i <= k; i++) {
		r *= n - k + i;
		r /= i;
	}
	return r;
}

// src/process.cxx:140-161
static double shuprop_port(size_t x, double p, size_t l)
{
	double xx = (double)x, ll = (double)l, s = 0.0;
	for (size_t k = 0; k <= x; k++) {
		double kk = (double)k;
		double t = pow(p, kk) * pow(0.5 - p, xx - kk);
		s += pow(2, xx) * (t * pow(1 - t, ll)) * (double)binom_port(x, k);
		if (s >= 1.0) {
			s = 1.0;
			break;
		}
	}
	return s;
}

// src/process.cxx:77-86
static size_t min_anchor_length_port(double p, double g, size_t l)
{
	size_t x = 1;
	while (shuprop_port(x, g / 2, l) < 1 - p) x++;
	return x;
}

// ───────────────────────── suffix array (stand-in for divsufsort64) ──────────

static inline unsigned sym3(unsigned char c)
{
	switch (c) {
		case 0: return 0;
		case '!': return 1;
		case '#': return 2;
		case 'A': return 3;
		case 'C': return 4;
		case 'G': return 5;
		case 'T': return 6;
		default: return 7;
	}
}

// Sort all suffixes of s[0..n) in unsigned-byte order, shorter suffix first.
// 21 symbols are packed order-preservingly into a 63-bit key; ties are broken
// by direct comparison.  Requires the alphabet {!,#,A,C,G,T}.
static void suffix_sort_port(const unsigned char *s, sidx n, sidx *SA)
{
	const int W = 21;
	std::vector<uint64_t> key((size_t)n);
	uint64_t roll = 0;
	// key[i] = symbols s[i..i+W) (zero padded); build right to left.
	for (sidx i = n - 1; i >= 0; i--) {
		roll = (roll >> 3) | ((uint64_t)sym3(s[i]) << (3 * (W - 1)));
		key[(size_t)i] = roll;
	}
	for (sidx i = 0; i < n; i++) SA[i] = i;
	std::sort(SA, SA + n, [&](sidx a, sidx b) {
		if (key[(size_t)a] != key[(size_t)b]) return key[(size_t)a] < key[(size_t)b];
		if (a == b) return false;
		sidx la = n - a, lb = n - b;
		sidx m = std::min(la, lb);
		if (m > W) {
			int c = memcmp(s + a + W, s + b + W, (size_t)(m - W));
			if (c != 0) return c < 0;
		}
		return la < lb;
	});
}

// ───────────────────────── enhanced suffix array ─────────────────────────

struct ival { // lcp_interval, src/esa.h:31-40
	sidx l, i, j, m;
};

struct esa_port {
	sidx size = 0; // 2L+1
	std::string S;
	std::vector<sidx> SA, LCP, CLD;
	std::vector<char> FVC;
	std::vector<ival> cache;

	static const size_t K = 6; // CACHE_LENGTH, src/esa.cxx:34

	sidx &rchild(sidx idx) { return CLD[(size_t)idx]; }
	sidx &lchild(sidx idx) { return CLD[(size_t)idx - 1]; }
	sidx rchild(sidx idx) const { return CLD[(size_t)idx]; }
	sidx lchild(sidx idx) const { return CLD[(size_t)idx - 1]; }

	// src/esa.cxx:69-81
	esa_port(const std::string &nucl, const int64_t *given_sa)
	{
		size = (sidx)nucl.size() * 2 + 1;
		S = nucl + '#' + revcomp_port(nucl);
		SA.resize((size_t)size);
		if (given_sa) {
			std::copy(given_sa, given_sa + size, SA.begin());
		} else {
			suffix_sort_port((const unsigned char *)S.data(), size, SA.data());
		}
		build_lcp();
		build_cld();
		build_fvc();
		build_cache();
	}

	// src/esa.cxx:305-347 — Φ / permuted-LCP construction.
	void build_lcp()
	{
		sidx len = size;
		LCP.assign((size_t)len + 1, 0);
		LCP[0] = -1;
		LCP[(size_t)len] = -1;
		std::vector<sidx> phi((size_t)len);
		phi[(size_t)SA[0]] = -1;
		for (sidx i = 1; i < len; i++) phi[(size_t)SA[(size_t)i]] = SA[(size_t)i - 1];
		sidx l = 0;
		const char *s = S.c_str(); // reads the terminating NUL like the reference
		for (sidx i = 0; i < len; i++) {
			sidx k = phi[(size_t)i];
			if (k != -1) {
				while (s[k + l] == s[i + l]) l++;
				phi[(size_t)i] = l;
				l--;
				if (l < 0) l = 0;
			} else {
				phi[(size_t)i] = -1;
			}
		}
		for (sidx i = 1; i < len; i++) LCP[(size_t)i] = phi[(size_t)SA[(size_t)i]];
	}

	// src/esa.cxx:256-298 — child table by a stack sweep over LCP.
	void build_cld()
	{
		CLD.assign((size_t)size + 1, 0);
		struct ent {
			sidx idx, lcp;
		};
		std::vector<ent> stack((size_t)size + 1);
		ent *top = stack.data();
		ent last;
		rchild(0) = size;
		top->idx = 0;
		top->lcp = -1;
		for (sidx k = 1; k < size + 1; k++) {
			while (LCP[(size_t)k] < top->lcp) {
				last = *top--;
				while (top->lcp == last.lcp) {
					rchild(top->idx) = last.idx;
					last = *top--;
				}
				if (LCP[(size_t)k] < top->lcp) {
					rchild(top->idx) = last.idx;
				} else {
					lchild(k) = last.idx;
				}
			}
			top++;
			top->idx = k;
			top->lcp = LCP[(size_t)k];
		}
	}

	// src/esa.cxx:239-250 — FVC[i] = S[SA[i] + LCP[i]].
	void build_fvc()
	{
		FVC.assign((size_t)size, 0);
		const char *s = S.c_str();
		for (sidx i = 0; i < size; i++) FVC[(size_t)i] = s[SA[(size_t)i] + LCP[(size_t)i]];
		// i == 0: LCP[0] = -1 ⇒ s[SA[0]-1]; the reference computes that value too
		// (esa.cxx:247-249 overwrites the '\0' stored at :242). SA[0] is the
		// position of '#', never 0, so the read is in range.
	}

	static ssize_t code_of(char c)
	{
		switch (c) {
			case 'A': return 0;
			case 'C': return 1;
			case 'G': return 2;
			case 'T': return 3;
			default: return -1;
		}
	}
	static char char_of(int code) { return "ACGT"[code & 3]; }

	// src/esa.cxx:212-228
	void cache_fill(char *str, size_t pos, ival in)
	{
		if (pos < K) {
			for (int code = 0; code < 4; ++code) {
				str[pos] = char_of(code);
				cache_fill(str, pos + 1, in);
			}
		} else {
			ssize_t code = 0;
			for (size_t i = 0; i < K; ++i) {
				code <<= 2;
				code |= code_of(str[i]);
			}
			cache[(size_t)code] = in;
		}
	}

	// src/esa.cxx:114-201
	void cache_dfs(char *str, size_t pos, ival in)
	{
		if (pos < K && in.i == -1 && in.j == -1) {
			cache_fill(str, pos, in);
			return;
		}
		if (pos >= K) {
			cache_fill(str, pos, in);
			return;
		}
		for (int code = 0; code < 4; ++code) {
			str[pos] = char_of(code);
			ival ij = child(in, str[pos]);
			if (ij.i == -1 && ij.j == -1) {
				cache_fill(str, pos + 1, in);
				continue;
			}
			if (ij.i == ij.j) {
				ij.l = (sidx)pos + 1;
				cache_fill(str, pos + 1, ij);
				continue;
			}
			if (ij.l <= (ssize_t)(pos + 1)) {
				cache_dfs(str, pos + 1, ij);
				continue;
			}
			if ((size_t)ij.l >= K) {
				cache_fill(str, pos + 1, in);
				continue;
			}
			cache_fill(str, pos + 1, in);
			char non_acgt = 0;
			size_t k = pos + 1;
			for (; k < (size_t)ij.l; k++) {
				char c = S[(size_t)(SA[(size_t)ij.i] + (sidx)k)];
				if (code_of(c) < 0) {
					non_acgt = 1;
					break;
				}
				str[k] = c;
			}
			if (non_acgt) {
				cache_fill(str, k, ij);
			} else {
				cache_dfs(str, k, ij);
			}
		}
	}

	// src/esa.cxx:90-101
	void build_cache()
	{
		cache.assign((size_t)1 << (2 * K), ival{0, 0, 0, 0});
		char str[K + 1];
		str[K] = '\0';
		sidx m = lchild(size);
		ival ij = {LCP[(size_t)m], 0, size - 1, m};
		cache_dfs(str, 0, ij);
	}

	// src/esa.cxx:361-427 — child interval of `ij` whose next character is `a`.
	// The reference's goto/do-while is unrolled into one loop; `m`,`l` are
	// narrowed to int exactly as at esa.cxx:374-375.
	ival child(ival ij, char a) const
	{
		const char *s = S.c_str();
		sidx i = ij.i, j = ij.j;
		if (i == j) {
			if (s[SA[(size_t)i] + ij.l] != a) ij.i = ij.j = -1;
			return ij;
		}
		int m = (int)ij.m;
		int l = (int)ij.l;
		char c = s[SA[(size_t)i] + l];
		for (;;) {
			if (c == a) {
				if (i != m - 1) {
					sidx n = lchild(m);
					return ival{LCP[(size_t)n], i, (sidx)m - 1, n};
				}
				return ival{LCP[(size_t)i], i, i, -1};
			}
			if (c > a) break;
			i = m;
			if (i == j) break;
			m = (int)rchild(m);
			if (LCP[(size_t)m] != l) break;
			c = FVC[(size_t)i];
		}
		bool hit = (i != ij.i) ? (FVC[(size_t)i] == a) : (s[SA[(size_t)i] + l] == a);
		if (hit) {
			ij.i = i;
			ij.j = j;
			ij.l = LCP[(size_t)m];
			ij.m = m;
		} else {
			ij.i = ij.j = -1;
		}
		return ij;
	}

	// src/esa.cxx:446-513
	ival match_from(const char *q, size_t qlen, sidx k, ival ij) const
	{
		const char *s = S.c_str();
		if (ij.i == -1 && ij.j == -1) return ij;
		if (ij.i == ij.j) {
			sidx p = SA[(size_t)ij.i];
			size_t kk = (size_t)ij.l;
			for (; kk < qlen && s[p + (sidx)kk]; kk++) {
				if (s[p + (sidx)kk] != q[kk]) {
					ij.l = (sidx)kk;
					return ij;
				}
			}
			ij.l = (sidx)kk;
			return ij;
		}
		ival res = ij;
		do {
			ij = child(ij, q[k]);
			sidx i = ij.i, j = ij.j;
			if (i == -1 && j == -1) {
				res.l = k;
				return res;
			}
			res.i = ij.i;
			res.j = ij.j;
			sidx l = (sidx)qlen;
			if (i < j && ij.l < l) l = ij.l;
			k++;
			for (int p = (int)SA[(size_t)i]; k < l; k++) {
				if (s[p + k] != q[k]) {
					res.l = k;
					return res;
				}
			}
		} while (k < (ssize_t)qlen);
		res.l = (sidx)qlen;
		return res;
	}

	// src/esa.cxx:525-531
	ival match(const char *q, size_t qlen) const
	{
		sidx m = lchild(size);
		ival ij = {LCP[(size_t)m], 0, size - 1, m};
		return match_from(q, qlen, 0, ij);
	}

	// src/esa.cxx:542-563
	ival match_cached(const char *q, size_t qlen) const
	{
		if (qlen <= K) return match(q, qlen);
		ssize_t off = 0;
		for (size_t i = 0; i < K && off >= 0; i++) {
			off <<= 2;
			off |= code_of(q[i]);
		}
		if (off < 0) return match(q, qlen);
		ival ij = cache[(size_t)off];
		if (ij.i == -1 && ij.j == -1) return match(q, qlen);
		return match_from(q, qlen, ij.l, ij);
	}
};

// ───────────────────────── homology algebra ─────────────────────────

struct hom { // class homology, src/process.h:14-144
	int rev = 0;       // direction: 0 forward, 1 reverse
	size_t iref = 0;   // index_reference
	size_t iproj = 0;  // index_reference_projected
	size_t iq = 0;     // index_query
	size_t len = 0;

	hom() = default;
	hom(size_t ir, size_t q, size_t l = 0) : rev(0), iref(ir), iproj(ir), iq(q), len(l) {}

	size_t start() const { return iproj; }
	size_t end() const { return iproj + len; }
	size_t start_query() const { return iq; }
	size_t end_query() const { return iq + len; }
	size_t extend(size_t stride) { return len += stride; }

	// src/process.h:72-80
	void project(size_t reference_length)
	{
		if (iref < reference_length) return;
		iproj = 2 * reference_length + 1 - len - iref;
		rev = 1;
	}
	bool starts_left_of(const hom &o) const { return start() < o.start(); }
	bool ends_left_of(const hom &o) const { return end() <= o.start(); }
	// src/process.h:86-97
	bool overlaps(const hom &o) const
	{
		if (start() == o.start()) return true;
		if (starts_left_of(o)) return !ends_left_of(o);
		if (o.starts_left_of(*this)) return !o.ends_left_of(*this);
		return false;
	}
	// src/process.h:119-143
	hom trim(size_t s, size_t e) const
	{
		if (e <= s) return *this;
		hom that = *this;
		size_t offset = (s > start() && s < end()) ? s - start() : 0;
		size_t drift = (end() > e && e > start()) ? end() - e : 0;
		that.iproj += offset;
		if (!rev) {
			that.iref += offset;
			that.iq += offset;
		} else {
			that.iref += drift;
			that.iq += drift;
		}
		that.len = len - offset - drift;
		return that;
	}
};

// src/process.cxx:171-184
static size_t lcp_port(const char *S, const char *Q, size_t remaining)
{
	size_t n = 0;
	while (n < remaining && S[n] == Q[n]) n++;
	return n;
}

// src/process.cxx:198-295
static std::vector<hom> anchor_homologies_port(const esa_port &ref, size_t threshold,
											   const char *seq, size_t query_length)
{
	std::vector<hom> hv;
	size_t border = (size_t)ref.size / 2;
	size_t last_q = 0, last_s = 0, last_len = 0;
	bool last_right = false;
	size_t this_q = 0, this_s = 0, this_len = 0;
	hom current(0, 0);

	auto anchor = [&]() {
		ival in = ref.match_cached(seq + this_q, query_length - this_q);
		this_len = (size_t)std::max(in.l, (sidx)0);
		this_s = (size_t)ref.SA[(size_t)in.i];
		return in.i == in.j && this_len >= threshold;
	};
	auto lucky = [&]() {
		size_t advance = this_q - last_q;
		size_t gap = this_q - last_q - last_len;
		size_t try_s = last_s + advance;
		if (try_s >= (size_t)ref.size || gap > threshold) return false;
		this_s = try_s;
		this_len = lcp_port(seq + this_q, ref.S.c_str() + try_s, query_length - this_q);
		return this_len >= threshold;
	};

	while (this_q < query_length) {
		if (lucky() || anchor()) {
			size_t end_s = last_s + last_len;
			size_t end_q = last_q + last_len;
			if (this_s > end_s && this_q - end_q == this_s - end_s &&
				(this_s < border) == (last_s < border)) {
				current.extend(this_q - end_q + this_len);
				last_right = true;
			} else {
				if (last_right || last_len / 2 >= threshold) {
					current.project(border);
					hv.push_back(current);
				}
				current = hom(this_s, this_q, this_len);
				last_right = false;
			}
			last_q = this_q;
			last_s = this_s;
			last_len = this_len;
		}
		this_q += this_len + 1;
	}
	if (last_len >= query_length) current = hom(last_s, 0, query_length);
	if (last_right || last_len / 2 >= threshold) {
		current.project(border);
		hv.push_back(current);
	}
	return hv;
}

// src/process.cxx:354-401 (with remove_if_i, :37-64)
static void filter_overlaps_max_port(std::vector<hom> &pile)
{
	if (pile.size() < 2) return;
	size_t size = pile.size();
	std::vector<ssize_t> pred_buf(size + 1, -1), score_buf(size + 1, 0);
	ssize_t *pred = pred_buf.data() + 1, *score = score_buf.data() + 1;
	pred[0] = -1;
	score[0] = (ssize_t)pile[0].len;
	for (ssize_t i = 1; i < (ssize_t)size; i++) {
		ssize_t best = 0, best_k = -1;
		for (ssize_t k = 0; k < i; k++) {
			if (!pile[(size_t)k].ends_left_of(pile[(size_t)i])) continue;
			if (score[k] > best) {
				best = score[k];
				best_k = k;
			}
		}
		pred[i] = best_k;
		score[i] = score[best_k] + (ssize_t)pile[(size_t)i].len;
	}
	std::vector<bool> keep(size, false);
	auto that = std::max_element(score_buf.begin(), score_buf.end());
	ssize_t idx = (that - score_buf.begin()) - 1;
	while (idx >= 0) {
		keep[(size_t)idx] = true;
		idx = pred[idx];
	}
	size_t w = 0;
	for (size_t r = 0; r < size; r++)
		if (keep[r]) pile[w++] = pile[r];
	pile.resize(w);
}

// src/process.cxx:438-443
static void sort_and_filter_port(std::vector<hom> &hv)
{
	std::sort(hv.begin(), hv.end(),
			  [](const hom &a, const hom &b) { return a.starts_left_of(b); });
	filter_overlaps_max_port(hv);
}

// ───────────────────────── tallies ─────────────────────────

struct tally { // evo_model, src/evo_model.h:13-19
	size_t subst = 0, homologs = 0;
	// src/evo_model.cxx:53-59
	void account(const char *a, const char *b, size_t n)
	{
		homologs += n;
		subst += (g_force_generic ? seqcmp_port : g_seqcmp)(a, b, n);
	}
	// src/evo_model.cxx:68-75
	void account_rev(const char *a, const char *b, size_t b_off, size_t n)
	{
		homologs += n;
		subst += (g_force_generic ? revseqcmp_port : g_revseqcmp)(a, b + b_off - n, n);
	}
	// src/evo_model.cxx:81-87
	tally &operator+=(const tally &o)
	{
		homologs += o.homologs;
		subst += o.subst;
		return *this;
	}
};

// src/evo_model.cxx:100-131
static double est_raw(size_t s, size_t h, bool zero_on_error)
{
	if (h == 0) return zero_on_error ? 0.0 : NAN;
	return s / (double)h;
}
static double est_ani(size_t s, size_t h, bool zero_on_error)
{
	if (h == 0) return zero_on_error ? 0.0 : NAN;
	return (1.0 - s / (double)h) * 100;
}
static double est_jc(size_t s, size_t h, bool zero_on_error)
{
	double d = est_raw(s, h, zero_on_error);
	d = -0.75 * log(1.0 - (4.0 / 3.0) * d);
	return d <= 0.0 ? 0.0 : d;
}

// src/process.cxx:620-658
static tally compare_one(const char *sa, const hom &ha, const char *sb, const hom &hb)
{
	tally t;
	if (!ha.overlaps(hb)) return t;
	size_t cs = std::max(ha.start(), hb.start());
	size_t ce = std::min(ha.end(), hb.end());
	size_t n = ce - cs;
	hom hat = ha.trim(cs, ce), hbt = hb.trim(cs, ce);
	if (ha.rev == hb.rev) {
		t.account(sa + hat.start_query(), sb + hbt.start_query(), n);
	} else if (hb.rev) {
		t.account_rev(sa + hat.start_query(), sb, hbt.end_query(), n);
	} else {
		t.account_rev(sb + hbt.start_query(), sa, hat.end_query(), n);
	}
	return t;
}

// src/process.cxx:566-611
static tally compare_lists(const char *sa, const std::vector<hom> &ha, const char *sb,
						   const std::vector<hom> &hb)
{
	tally total;
	auto right = hb.begin();
	std::vector<hom> pile;
	for (const hom &h : ha) {
		auto done = [&h](const hom &o) { return o.ends_left_of(h); };
		auto over = [&h](const hom &o) { return o.overlaps(h); };
		pile.erase(std::remove_if(pile.begin(), pile.end(), done), pile.end());
		right = std::find_if_not(right, hb.end(), done);
		auto far = std::find_if_not(right, hb.end(), over);
		std::copy(right, far, std::back_inserter(pile));
		right = far;
		for (const hom &o : pile) total += compare_one(sa, h, sb, o);
	}
	return total;
}

// src/process.cxx:725-776
static std::vector<std::vector<hom>> complete_delete_port(const std::vector<std::vector<hom>> &H)
{
	size_t n = H.size();
	std::vector<std::vector<hom>> core(n);
	std::vector<size_t> front(n, 0);
	auto all_open = [&]() {
		for (size_t g = 0; g < n; g++)
			if (!(front[g] < H[g].size())) return false;
		return true;
	};
	while (all_open()) {
		size_t cs = 0, ce = 0, arg = 0;
		for (size_t g = 0; g < n; g++) {
			size_t s = H[g][front[g]].start(), e = H[g][front[g]].end();
			if (g == 0 || s > cs) cs = s; // max_element: first maximum; value only
			if (g == 0 || e < ce) {        // min_element: first minimum
				ce = e;
				arg = g;
			}
		}
		if (cs < ce)
			for (size_t g = 0; g < n; g++) core[g].push_back(H[g][front[g]].trim(cs, ce));
		front[arg]++;
	}
	return core;
}

// ───────────────────────── the path: process() ─────────────────────────

struct run_port {
	size_t n = 0;
	size_t ref_idx = 0;
	size_t threshold = 0;
	size_t forced_threshold = 0; // tests: stand in for a longer reference's threshold (0 = process.cxx:416-417)
	double gc = 0;
	std::vector<std::string> seqs;
	std::vector<std::vector<hom>> raw;      // anchor_homologies output
	std::vector<std::vector<hom>> filtered; // after sort + filter (+ complete deletion)
	std::vector<tally> matrix;              // n*n row-major
	esa_port *esa = nullptr;
	double t_esa = 0, t_anchor = 0, t_compare = 0; // seconds, for bench.py's cpu_baseline
	~run_port() { delete esa; }
};

// src/process.cxx:408-556 (no -p output, no progress bar)
static void process_port(run_port &r, bool complete_deletion, const int64_t *sa, int threads,
						 size_t q_begin, size_t q_end, bool do_compare)
{
	size_t N = r.n;
	const std::string &subject = r.seqs[r.ref_idx];
	auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double c0 = clk();
	if (!r.esa) r.esa = new esa_port(subject, sa);
	double c1 = clk();
	r.t_esa = c1 - c0;
	r.gc = gc_content_port(subject.data(), subject.size());
	r.threshold = min_anchor_length_port(0.025, r.gc, (size_t)r.esa->size);
	if (r.forced_threshold) r.threshold = r.forced_threshold;
	r.raw.assign(N, {});
	r.filtered.assign(N, {});
	(void)threads;
#pragma omp parallel for num_threads(threads) schedule(dynamic)
	for (size_t j = q_begin; j < q_end; j++) {
		auto hv = anchor_homologies_port(*r.esa, r.threshold, r.seqs[j].c_str(), r.seqs[j].size());
		r.raw[j] = hv;
		sort_and_filter_port(hv);
		r.filtered[j] = std::move(hv);
	}
	double c2 = clk();
	r.t_anchor = c2 - c1;
	if (complete_deletion) r.filtered = complete_delete_port(r.filtered);
	r.matrix.assign(N * N, tally());
	if (!do_compare) return;
	double c3 = clk();
#pragma omp parallel for num_threads(threads) schedule(dynamic)
	for (size_t i = q_begin; i < q_end; i++) {
		for (size_t j = i + 1; j < q_end; j++) {
			tally t = compare_lists(r.seqs[i].c_str(), r.filtered[i], r.seqs[j].c_str(), r.filtered[j]);
			r.matrix[i * N + j] = t;
			r.matrix[j * N + i] = t;
		}
	}
	r.t_compare = clk() - c3;
}

// src/io.cxx:141-163 — PHYLIP text for one matrix.
static std::string phylip_port(const std::vector<std::string> &names, const std::vector<double> &d,
							   bool ani)
{
	std::ostringstream out;
	size_t N = names.size();
	out << N << std::endl;
	out.precision(4);
	if (ani) out << std::dec;
	else out << std::scientific;
	for (size_t i = 0; i < N; i++) {
		out << names[i];
		for (size_t j = 0; j < N; j++) {
			double v = (i == j) ? 0.0 : d[i * N + j];
			out << "  " << v;
		}
		out << std::endl;
	}
	return out.str();
}

// ───────────────────────── C entry points for ctypes ─────────────────────────

struct orc_hom { // flat mirror of `hom`
	int64_t rev, iref, iproj, iq, len;
};

static void to_flat(const std::vector<hom> &v, orc_hom *out)
{
	for (size_t i = 0; i < v.size(); i++)
		out[i] = orc_hom{v[i].rev, (int64_t)v[i].iref, (int64_t)v[i].iproj, (int64_t)v[i].iq,
						 (int64_t)v[i].len};
}
static std::vector<hom> from_flat(const orc_hom *in, size_t n)
{
	std::vector<hom> v(n);
	for (size_t i = 0; i < n; i++) {
		v[i].rev = (int)in[i].rev;
		v[i].iref = (size_t)in[i].iref;
		v[i].iproj = (size_t)in[i].iproj;
		v[i].iq = (size_t)in[i].iq;
		v[i].len = (size_t)in[i].len;
	}
	return v;
}

extern "C" {

size_t orc_seqcmp(const char *a, const char *b, size_t n) { return seqcmp_port(a, b, n); }
size_t orc_revseqcmp(const char *a, const char *b, size_t n) { return revseqcmp_port(a, b, n); }
// variant: 0 the byte loop, 1 what the reference's resolver would bind on this CPU, 2 AVX2, 3 AVX-512 (seqcmp only);
// returns (size_t)-1 when the CPU lacks the variant
size_t orc_seqcmp_variant(int variant, int rev, const char *a, const char *b, size_t n)
{
	if (variant == 0) return rev ? revseqcmp_port(a, b, n) : seqcmp_port(a, b, n);
	if (variant == 1) return rev ? g_revseqcmp(a, b, n) : g_seqcmp(a, b, n);
#if defined(__x86_64__)
	__builtin_cpu_init();
	if (variant == 2 && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt"))
		return rev ? revseqcmp_avx2_port(a, b, n) : seqcmp_avx2_port(a, b, n);
	if (variant == 3 && !rev && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
		__builtin_cpu_supports("popcnt"))
		return seqcmp_avx512_port(a, b, n);
#endif
	return (size_t)-1;
}
const char *orc_seqcmp_variant_name(int rev) { return rev ? g_revseqcmp_name : g_seqcmp_name; }
void orc_force_generic(int on) { g_force_generic = on != 0; }

void orc_revcomp(const char *in, size_t n, char *out)
{
	std::string r = revcomp_port(std::string(in, n));
	memcpy(out, r.data(), n);
}
size_t orc_filter_nucl(const char *in, size_t n, char *out)
{
	std::string r = filter_nucl_port(std::string(in, n));
	memcpy(out, r.data(), r.size());
	return r.size();
}
double orc_gc_content(const char *s, size_t n) { return gc_content_port(s, n); }
size_t orc_min_anchor_length(double p, double g, size_t l) { return min_anchor_length_port(p, g, l); }
double orc_shuprop(size_t x, double p, size_t l) { return shuprop_port(x, p, l); }

void orc_suffix_array(const char *s, int64_t n, int64_t *sa)
{
	suffix_sort_port((const unsigned char *)s, n, sa);
}

double orc_estimate(int kind, uint64_t subst, uint64_t homologs, int zero_on_error)
{
	switch (kind) {
		case 0: return est_jc(subst, homologs, zero_on_error);
		case 1: return est_raw(subst, homologs, zero_on_error);
		default: return est_ani(subst, homologs, zero_on_error);
	}
}

// --- ESA handle ---
void *orc_esa_create(const char *nucl, size_t n, const int64_t *sa_or_null)
{
	return new esa_port(std::string(nucl, n), sa_or_null);
}
void orc_esa_destroy(void *e) { delete (esa_port *)e; }
int64_t orc_esa_size(void *e) { return ((esa_port *)e)->size; }
void orc_esa_arrays(void *e, int64_t *sa, int64_t *lcp, int64_t *cld, char *fvc, char *s)
{
	esa_port *E = (esa_port *)e;
	if (sa) std::copy(E->SA.begin(), E->SA.end(), sa);
	if (lcp) std::copy(E->LCP.begin(), E->LCP.end(), lcp);
	if (cld) std::copy(E->CLD.begin(), E->CLD.end(), cld);
	if (fvc) std::copy(E->FVC.begin(), E->FVC.end(), fvc);
	if (s) memcpy(s, E->S.data(), E->S.size());
}
// out = {l, i, j, SA[i] or -1}
void orc_esa_match(void *e, const char *q, size_t qlen, int cached, int64_t out[4])
{
	esa_port *E = (esa_port *)e;
	ival r = cached ? E->match_cached(q, qlen) : E->match(q, qlen);
	out[0] = r.l;
	out[1] = r.i;
	out[2] = r.j;
	out[3] = (r.i >= 0) ? E->SA[(size_t)r.i] : -1;
}
// Count 6-mer cache entries whose stored interval does not actually share its
// first `l` characters with the key (the non-ACGT fast-forward quirk at
// esa.cxx:174-199).  Tests use this to know when the reference's cached match
// can differ from the plain longest match.
int64_t orc_esa_cache_quirks(void *e)
{
	esa_port *E = (esa_port *)e;
	int64_t bad = 0;
	for (size_t code = 0; code < E->cache.size(); code++) {
		ival c = E->cache[code];
		if (c.i < 0) continue;
		char key[7];
		for (int t = 0; t < 6; t++) key[t] = "ACGT"[(code >> (2 * (5 - t))) & 3];
		sidx p = E->SA[(size_t)c.i];
		sidx l = std::min<sidx>(c.l, 6);
		for (sidx t = 0; t < l; t++)
			if (E->S[(size_t)(p + t)] != key[t]) {
				bad++;
				break;
			}
	}
	return bad;
}

// anchor_homologies on one query; returns count, writes up to cap entries.
size_t orc_anchor(void *e, size_t threshold, const char *q, size_t qlen, orc_hom *out, size_t cap)
{
	auto hv = anchor_homologies_port(*(esa_port *)e, threshold, q, qlen);
	if (hv.size() <= cap) to_flat(hv, out);
	return hv.size();
}
// std::sort + filter_overlaps_max in place; returns new count.
size_t orc_sort_filter(orc_hom *h, size_t n, int do_sort)
{
	auto v = from_flat(h, n);
	if (do_sort) sort_and_filter_port(v);
	else filter_overlaps_max_port(v);
	to_flat(v, h);
	return v.size();
}
// homology predicates for the Tprocess.cxx literals
int orc_hom_pred(const orc_hom *a, const orc_hom *b, int which)
{
	hom A = from_flat(a, 1)[0], B = from_flat(b, 1)[0];
	switch (which) {
		case 0: return A.starts_left_of(B);
		case 1: return A.ends_left_of(B);
		default: return A.overlaps(B);
	}
}
void orc_hom_trim(const orc_hom *a, size_t s, size_t e, orc_hom *out)
{
	hom r = from_flat(a, 1)[0].trim(s, e);
	std::vector<hom> v{r};
	to_flat(v, out);
}
void orc_hom_project(orc_hom *a, size_t reflen)
{
	hom r = from_flat(a, 1)[0];
	r.project(reflen);
	std::vector<hom> v{r};
	to_flat(v, a);
}
// complete_delete over n lists given as concatenated array + offsets[n+1];
// writes result the same way (out_off[n+1]); returns total count.
size_t orc_complete_delete(size_t n, const orc_hom *in, const size_t *off, orc_hom *out,
						   size_t *out_off, size_t cap)
{
	std::vector<std::vector<hom>> H(n);
	for (size_t g = 0; g < n; g++) H[g] = from_flat(in + off[g], off[g + 1] - off[g]);
	auto core = complete_delete_port(H);
	size_t tot = 0;
	for (size_t g = 0; g < n; g++) {
		out_off[g] = tot;
		if (tot + core[g].size() <= cap) to_flat(core[g], out + tot);
		tot += core[g].size();
	}
	out_off[n] = tot;
	return tot;
}
// compare(list, list) on raw buffers
void orc_compare_lists(const char *sa, const orc_hom *ha, size_t na, const char *sb,
					   const orc_hom *hb, size_t nb, uint64_t out[2])
{
	tally t = compare_lists(sa, from_flat(ha, na), sb, from_flat(hb, nb));
	out[0] = t.subst;
	out[1] = t.homologs;
}

// --- whole path ---
void *orc_run_create(size_t n, const char *const *seq, const size_t *len, size_t ref_idx)
{
	run_port *r = new run_port();
	r->n = n;
	r->ref_idx = ref_idx;
	for (size_t i = 0; i < n; i++) r->seqs.emplace_back(seq[i], len[i]);
	return r;
}
void orc_run_destroy(void *r) { delete (run_port *)r; }
// queries [q_begin, q_end) are anchored and compared among themselves; pass
// 0..n for the full path. sa_or_null: optional precomputed suffix array of S.
void orc_run_process(void *rp, int complete_deletion, const int64_t *sa_or_null, int threads,
					 size_t q_begin, size_t q_end, int do_compare)
{
	process_port(*(run_port *)rp, complete_deletion != 0, sa_or_null, threads < 1 ? 1 : threads,
				 q_begin, q_end, do_compare != 0);
}
void orc_run_times(void *rp, double out[3])
{
	run_port *r = (run_port *)rp;
	out[0] = r->t_esa;
	out[1] = r->t_anchor;
	out[2] = r->t_compare;
}
size_t orc_run_threshold(void *rp) { return ((run_port *)rp)->threshold; }
// The threshold is a function of the reference's length and GC (13-17 for 1 Mbp-100 Mbp);
// forcing it lets small test inputs walk the code paths a 100 Mbp reference takes.
void orc_run_force_threshold(void *rp, size_t threshold) { ((run_port *)rp)->forced_threshold = threshold; }
double orc_run_gc(void *rp) { return ((run_port *)rp)->gc; }
void *orc_run_esa(void *rp) { return ((run_port *)rp)->esa; }
size_t orc_run_hom_count(void *rp, size_t j, int filtered)
{
	run_port *r = (run_port *)rp;
	return (filtered ? r->filtered : r->raw)[j].size();
}
void orc_run_homs(void *rp, size_t j, int filtered, orc_hom *out)
{
	run_port *r = (run_port *)rp;
	to_flat((filtered ? r->filtered : r->raw)[j], out);
}
void orc_run_matrix(void *rp, uint64_t *subst, uint64_t *homologs)
{
	run_port *r = (run_port *)rp;
	for (size_t i = 0; i < r->n * r->n; i++) {
		subst[i] = r->matrix[i].subst;
		homologs[i] = r->matrix[i].homologs;
	}
}

// ── -p FILE: the reference positions of the core alignment with every block's segregating sites ──
// src/process.cxx:684-698 — a[i] != b[i]
static void is_segsite_port(const char *a, const char *b, char *out, size_t n)
{
	for (size_t i = 0; i < n; i++) out[i] = a[i] != b[i];
}
// src/process.cxx:700-708 — a[i] against b[n-1-i] under the complement test
static void is_segsite_rev_port(const char *a, const char *b, char *out, size_t n)
{
	for (size_t i = 0; i < n; i++) out[i] = (((int)a[i] ^ (int)b[n - i - 1]) & 6) != 4;
}
// src/process.cxx:671-712 — the segregating sites of two homologies over their common stretch
static std::vector<char> get_segsites_port(const std::string &sa, const hom &ha, const std::string &sb, const hom &hb)
{
	if (!ha.overlaps(hb)) return {};
	size_t cs = std::max(ha.start(), hb.start()), ce = std::min(ha.end(), hb.end());
	size_t n = ce - cs;
	hom hat = ha.trim(cs, ce), hbt = hb.trim(cs, ce);
	std::vector<char> ret(n, 0);
	if (ha.rev == hb.rev && ha.rev == 0) {
		is_segsite_port(sa.c_str() + hat.start_query(), sb.c_str() + hbt.start_query(), ret.data(), n);
	} else if (ha.rev == hb.rev) {
		is_segsite_port(sa.c_str() + hat.start_query(), sb.c_str() + hbt.start_query(), ret.data(), n);
		std::reverse(ret.begin(), ret.end());
	} else if (hb.rev == 1) {
		is_segsite_rev_port(sa.c_str() + hat.start_query(), sb.c_str() + hbt.end_query() - n, ret.data(), n);
	} else {
		is_segsite_rev_port(sb.c_str() + hbt.start_query(), sa.c_str() + hat.end_query() - n, ret.data(), n);
	}
	return ret;
}
// src/process.cxx:471-513 — the text of the -p file, from the lists after complete deletion
static std::string positions_port(const run_port &r)
{
	std::ostringstream out;
	const std::string &subject = r.seqs[r.ref_idx];
	const auto &homos = r.filtered[0];
	size_t counter = 1;
	for (size_t i = 0; i < homos.size(); i++) {
		const hom &h = homos[i];
		std::vector<char> seg(h.len, 0);
		for (size_t m = 0; m < r.n; m++) {
			auto f = get_segsites_port(r.seqs[0], h, r.seqs[m], r.filtered[m][i]);
			for (size_t t = 0; t < f.size(); t++) seg[t] |= f[t];
		}
		std::vector<size_t> pos;
		for (size_t t = 0; t < seg.size(); t++)
			if (seg[t]) pos.push_back(t);
		size_t start = h.start(), end = h.end();
		out << ">part" << counter++ << "\t(" << (start + 1) << ".." << (end + 1) << ")  " << pos.size();
		for (size_t p : pos) out << "  " << (p + 1);
		out << std::endl;
		out << std::string(subject.begin() + start, subject.begin() + end) << std::endl;
	}
	return out.str();
}
size_t orc_run_positions(void *rp, char *out, size_t cap)
{
	std::string s = positions_port(*(run_port *)rp);
	if (out && cap >= s.size() + 1) memcpy(out, s.c_str(), s.size() + 1);
	return s.size() + 1;
}

// evo_model::bootstrap (src/evo_model.cxx:136-147) for a whole matrix, in print_matrix's order
// (src/io.cxx:192-203): substitutions redrawn from Binomial(homologs, substitutions / homologs) with one
// std::mt19937 running through all cells, `rounds` matrices one after the other.  The reference seeds the
// engine from std::random_device (src/phylonium.cxx:78-91); tests seed both sides alike.
void orc_bootstrap(uint32_t seed, size_t rounds, size_t cells, const uint64_t *subst, const uint64_t *homologs, uint64_t *out)
{
	std::mt19937 prng(seed);
	for (size_t k = 0; k < rounds; k++)
		for (size_t t = 0; t < cells; t++) {
			double rate = subst[t] / (double)homologs[t];
			std::binomial_distribution<> d((int)homologs[t], rate);
			out[k * cells + t] = (uint64_t)d(prng);
		}
}

// PHYLIP text (src/io.cxx:141-233). names: n C strings. kind: 0 jc, 1 raw, 2 ani.
// Returns bytes needed (incl. NUL); writes at most cap bytes.
size_t orc_phylip(size_t n, const char *const *names, const uint64_t *subst,
				  const uint64_t *homologs, int kind, char *out, size_t cap)
{
	std::vector<std::string> nm(names, names + n);
	std::vector<double> d(n * n);
	for (size_t i = 0; i < n * n; i++) d[i] = orc_estimate(kind, subst[i], homologs[i], 0);
	std::string s = phylip_port(nm, d, kind == 2);
	if (s.size() + 1 <= cap) memcpy(out, s.c_str(), s.size() + 1);
	return s.size() + 1;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

} // extern "C"

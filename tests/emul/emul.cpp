// emul.cpp — TEST INFRASTRUCTURE. Compiles the product's host/device chain code
// (phylonium_amd/csrc/anchor_core.h, lean_core.h, hostlogic.hpp) with g++ and runs the
// phase-A pipeline (speculative chunks → bridges → walk + fold → sort + filter)
// serially on the CPU, so the algorithm can be checked against the oracle in
// the GPU-less build container.  The product library never links this file.
#define LEAN_COUNT_WHY 1
#include <cstdio>
#include <vector>

#include "../../phylonium_amd/csrc/hostlogic.hpp"
#include "../../phylonium_amd/csrc/lean_core.h"

using namespace phy;

struct EmulOut {
	std::vector<std::vector<RawHom>> raw;
	std::vector<std::vector<phylo_homology>> filtered;
	uint32_t threshold, k, C, nchunks;
	uint64_t steps_spec = 0, steps_bridge = 0, cmp_calls = 0, pool_used = 0, rounds = 0, slow_steps = 0, overruns = 0;
	int error = 0;
};

// the lean (2-bit) chain of lean_core.h, one lane at a time
template <class Lane, class Begin, class Done>
static void run_lean_lane(Lane &L, const uint8_t *qbase, const RefIndex &R, const LeanIndex &X, const LeanTables &T,
						  Begin begin, Done done, uint64_t *steps, uint64_t *trips, uint64_t *slow, const uint32_t *ring0 = nullptr)
{
	uint32_t ring[LEAN_RING_WORDS] = {0};
	if (ring0) // a bridge taken up from its packed form: the ring holds the record's words [wb, we)
		for (uint32_t i = 0; i < L.ln.we - L.ln.wb; i++) ring[(L.ln.wb + i) % LEAN_RING_WORDS] = ring0[i];
	uint64_t lane_trips = 0;
	for (;;) {
		if (L.ln.fin) {
			done();
			(*steps)++;
			L.ln.fin = false;
		}
		if (L.ln.ph == LP_STEP && !begin()) break;
		static const bool self_check = getenv("EMUL_SELF_CHECK") != nullptr; // every step against the definition
		static LeanLane want;
		static bool have_want = false;
		if (self_check && L.ln.ph == LP_STEP) {
			want = L.ln;
			want.q_cap = NO_BAD; // the definition's answer, uncut
			lean_resolve_scalar(want, qbase, R, X);
			have_want = true;
		}
		lean_trip_cpu(L.ln, ring, qbase, R, X, T, slow);
		if (self_check && L.ln.fin && have_want && !L.ln.ovr) {
			if (L.ln.r_len != want.r_len || L.ln.r_accepted != want.r_accepted || (want.r_accepted && L.ln.r_s != want.r_s)) {
				const uint32_t code = (uint32_t)(L.ln.qcode >> (2u * (16u - R.k)));
				const U4 sl = R.SLOT[code];
				fprintf(stderr, "emul self-check: step at q %u: got (s %u, len %u, acc %d), the definition says (s %u, len %u, acc %d); k-mer %u slot %08x %08x %08x %08x bucket [%u, %u)\n",
						L.ln.r_q, L.ln.r_s, L.ln.r_len, (int)L.ln.r_accepted, want.r_s, want.r_len, (int)want.r_accepted, code, sl.x, sl.y, sl.z, sl.w, R.T[code], R.T[code + 1]);
				abort();
			}
			have_want = false;
		}
		(*trips)++;
		if (++lane_trips > 50000000ull) {
			fprintf(stderr, "emul: lean lane stuck: ph %u q %u qlen %u\n", L.ln.ph, L.ln.q, L.ln.qlen);
			abort();
		}
	}
}

extern "C" {

// mode: bit 1 every step through the chain's slow resolver (bit 0 is ignored: there is one chain, lean_core.h's)
void *emul_run2(size_t n, const char *const *seq, const size_t *len, size_t ref_idx, size_t threshold,
				unsigned forced_C, unsigned forced_k, unsigned mode);
void *emul_run(size_t n, const char *const *seq, const size_t *len, size_t ref_idx, size_t threshold,
			   unsigned forced_C, unsigned forced_k)
{
	const char *m = getenv("EMUL_MODE");
	return emul_run2(n, seq, len, ref_idx, threshold, forced_C, forced_k, m ? (unsigned)atoi(m) : 0u);
}

void *emul_run2(size_t n, const char *const *seq, const size_t *len, size_t ref_idx, size_t threshold,
				unsigned forced_C, unsigned forced_k, unsigned mode)
{
	EmulOut *E = new EmulOut();
	// reference index
	size_t L = len[ref_idx];
	uint32_t ns = (uint32_t)(2 * L + 1);
	std::vector<uint8_t> S((size_t)ns + 64, 0);
	memcpy(S.data(), seq[ref_idx], L);
	S[L] = '#';
	revcomp((const uint8_t *)seq[ref_idx], L, S.data() + L + 1);
	std::vector<uint32_t> SA((size_t)ns + 4, 0), LCP((size_t)ns + 1 + 4, 0), T;
	const bool chatty = getenv("EMUL_PROGRESS") != nullptr;
	if (chatty) fprintf(stderr, "emul: suffix array of %u\n", ns);
	suffix_array_u32(S.data(), ns, SA.data());
	if (chatty) fprintf(stderr, "emul: lcp, tables\n");
	lcp_kasai(S.data(), ns, SA.data(), LCP.data());
	uint32_t k = forced_k ? forced_k : choose_k(ns);
	kmer_table(S.data(), ns, k, T);
	T.resize(T.size() + 4, ns);
	if (threshold == 0) threshold = min_anchor_length(0.025, gc_content((const uint8_t *)seq[ref_idx], L), ns);
	std::vector<U4> SAX;
	build_sax(S.data(), ns, SA.data(), LCP.data(), SAX);
	std::vector<U4> SLOT;
	build_slots(T, SAX, ns, k, SLOT);
	RefIndex R = {S.data(), SAX.data(), LCP.data(), SLOT.data(), T.data(), ns, k, (uint32_t)threshold};
	E->threshold = (uint32_t)threshold;
	E->k = k;

	// genomes, padded
	std::vector<uint64_t> qoff(n);
	std::vector<uint32_t> qlen(n);
	uint64_t tot = 0;
	for (size_t j = 0; j < n; j++) {
		qoff[j] = tot;
		qlen[j] = (uint32_t)len[j];
		// debugging large inputs: the reference against itself costs O(L^2/C) here (the product
		// writes that list directly, phylo_abi.hip) — leave it out
		if (j == ref_idx && getenv("EMUL_SKIP_SELF")) qlen[j] = 0;
		tot += ((len[j] + 63) / 64) * 64 + 64;
	}
	std::vector<uint8_t> qbase(tot, 0);
	for (size_t j = 0; j < n; j++) memcpy(qbase.data() + qoff[j], seq[j], len[j]);

	ChunkPlan P = plan_chunks(qlen, (uint32_t)threshold, forced_C, 256u * 4u * 256u);
	E->C = P.C;
	E->nchunks = P.nchunks;
	std::vector<Anchor> spec_anchors((size_t)P.anchor_slots + 1);
	std::vector<uint32_t> spec_cnt(P.nchunks, 0), visited((size_t)tot / 32 + 8, 0);
	std::vector<SpecExit> spec_exit(P.nchunks);
	std::vector<BridgeRec> bridge(P.nchunks);
	uint32_t pool_blocks = 1024 + P.nchunks;
	std::vector<PoolBlock> pool(pool_blocks);
	uint32_t pool_next = 0, error = 0, fetch[2] = {0, 0}, overrun = 0;
	PhaseA A;
	A.qbase = qbase.data();
	A.qoff = qoff.data();
	A.qlen = qlen.data();
	A.qchunk0 = P.qchunk0.data();
	A.items = P.items.data();
	A.chunk_query = P.chunk_query.data();
	A.nchunks = P.nchunks;
	A.C = P.C;
	A.cap = P.cap;
	A.qanc0 = P.qanc0.data();
	A.spec_anchors = spec_anchors.data();
	A.spec_cnt = spec_cnt.data();
	A.spec_exit = spec_exit.data();
	A.visited = visited.data();
	A.bridge = bridge.data();
	A.pool = pool.data();
	A.pool_blocks = pool_blocks;
	A.pool_next = &pool_next;
	A.error = &error;
	A.fetch = fetch;
	A.overrun = &overrun;

	if (chatty) fprintf(stderr, "emul: index done (k %u), %u chunks of %u\n", k, P.nchunks, P.C);
	// the lean chain's packed tables (the product builds them on the device: lean_kernels.hip)
	// over-deep entries of the reference's 6-mer cache (esa.cxx:174-199): as in the product (phylo_anchor), the lean
	// chains then run with every step through the slow resolver, which reproduces what the reference answers
	const std::vector<CacheQuirk> quirks = getenv("EMUL_NO_QUIRK") ? std::vector<CacheQuirk>() : esa_cache_quirks(S.data(), ns, SA.data());
	std::vector<U4> quirk_tab;
	for (const CacheQuirk &e : quirks) quirk_tab.push_back(U4{e.prefix, e.k | (e.depth << 8), e.lo, e.hi});
	if (!quirk_tab.empty()) mode |= 2u;
	const bool lean = true;
	std::vector<uint32_t> S2, SBAD, Q2, QBAD, qbad_off;
	LeanIndex X = {};
	LeanTables LT = {};
	if (lean) {
		S2.resize((size_t)ns / 16 + 16);
		lean_pack_host(S.data(), ns, S2.data(), S2.size());
		for (uint32_t i = 0; i < ns; i++)
			if (nuc_code(S[i]) > 3) SBAD.push_back(i);
		SBAD.push_back(ns);
		Q2.resize((size_t)tot / 16 + 16);
		lean_pack_host(qbase.data(), tot, Q2.data(), Q2.size());
		qbad_off.assign(n + 1, 0);
		for (size_t j = 0; j < n; j++) {
			qbad_off[j] = (uint32_t)QBAD.size();
			for (uint32_t i = 0; i < qlen[j]; i++)
				if (nuc_code(qbase[qoff[j] + i]) > 3) QBAD.push_back(i);
		}
		qbad_off[n] = (uint32_t)QBAD.size();
		QBAD.push_back(0);
		X.S2 = S2.data();
		X.SBAD = SBAD.data();
		X.nsb = (uint32_t)SBAD.size();
		X.sb_end = ns;
		X.sb_first = SBAD[0];
		X.Q2 = Q2.data();
		X.QBAD = QBAD.data();
		X.qbad_off = qbad_off.data();
		X.force_slow = (mode & 2u) ? 1u : 0u;
		X.quirk = quirk_tab.data();
		X.nquirk = (uint32_t)quirk_tab.size();
		LT.slot = (const uint8_t *)SLOT.data();
		LT.sax = (const uint8_t *)SAX.data();
		LT.q2 = (const uint8_t *)Q2.data();
		LT.s2 = (const uint8_t *)S2.data();
	}
	// K1: speculative chains
	for (uint32_t it = 0; it < P.nchunks; it++) {
		if (lean) {
			LeanSpec ln;
			ln.start(A, X, P.items[it]);
			VisDirect vd = {A.visited};
			run_lean_lane(ln, qbase.data(), R, X, LT, [&] { return ln.begin_step(A, X, vd); }, [&] { ln.step_done(A); },
						  &E->steps_spec, &E->rounds, &E->slow_steps);
			continue;
		}
	}
	// K1b: the open ends of cut comparisons (lean chains)
	E->overruns = 0;
	if (lean && overrun) {
		for (uint32_t gc = 0; gc < P.nchunks; gc++) E->overruns += (spec_cnt[gc] & LEAN_OVERRUN_BIT) ? 1 : 0;
		for (size_t j = 0; j < n; j++) lean_overrun_resolve_query(A, R, (uint32_t)j);
	}
	// K2: bridges
	auto alloc = [&]() -> uint32_t {
		if (pool_next >= pool_blocks) return NO_BLOCK;
		return pool_next++;
	};
	std::vector<uint32_t> bridge_steps; // EMUL_BRIDGE_STATS: how long the dependent chains of the bridge kernel are
	const bool bstats = getenv("EMUL_BRIDGE_STATS") != nullptr;
	for (uint32_t it = 0; it < P.nchunks; it++) {
		const uint64_t before = E->steps_bridge;
		if (lean) {
			// as the product does it: started and taken through its first begin_step by one pass (lean_bridge_prepare_kernel),
			// packed, and taken up from the packed form by the lane that walks it — its ring filled from the record
			LeanBridge first, ln;
			first.start(A, X, P.items[it]);
			if (!first.begin_step(A, X, R)) {
				if (bstats) bridge_steps.push_back(0);
				continue;
			}
			uint32_t w[LeanBridge::PACKED_WORDS];
			first.pack(w);
			for (uint32_t i = 0; i < LeanBridge::PACKED_RING; i++) w[24 + i] = Q2[first.ln.qw0 + (first.ln.q >> 4) + i];
			ln.unpack(A, w);
			run_lean_lane(ln, qbase.data(), R, X, LT, [&] { return ln.begin_step(A, X, R); }, [&] { ln.step_done(A, alloc); },
						  &E->steps_bridge, &E->rounds, &E->slow_steps, w + 24);
			if (bstats) bridge_steps.push_back((uint32_t)(E->steps_bridge - before));
			if (bstats && atoi(getenv("EMUL_BRIDGE_STATS")) > 1 && E->steps_bridge - before >= 30)
				fprintf(stderr, "long bridge: query %u chunk %u steps %llu from q %u to q %u (last anchor q %u s %u len %u)\n", A.chunk_query[P.items[it]],
						P.items[it], (unsigned long long)(E->steps_bridge - before), spec_exit[P.items[it]].q, ln.ln.q, ln.ln.lq, ln.ln.ls, ln.ln.ll);
			continue;
		}
	}
	if (bstats && !bridge_steps.empty()) {
		std::sort(bridge_steps.begin(), bridge_steps.end());
		const size_t m = bridge_steps.size();
		double sum = 0;
		for (uint32_t v : bridge_steps) sum += v;
		fprintf(stderr, "emul: %zu bridges, steps mean %.1f  median %u  p99 %u  p99.9 %u  p99.99 %u  max %u\n", m, sum / m,
				bridge_steps[m / 2], bridge_steps[(size_t)(m * 0.99)], bridge_steps[(size_t)(m * 0.999)],
				bridge_steps[(size_t)(m * 0.9999)], bridge_steps[m - 1]);
	}
	E->pool_used = pool_next;
	E->error = (int)error;
	// K3: walk + fold, then host sort + filter
	uint32_t border = (uint32_t)L;
	E->raw.resize(n);
	E->filtered.resize(n);
	for (size_t j = 0; j < n; j++) {
		FoldState f;
		fold_init(&f);
		RawHom h;
		std::vector<RawHom> &out = E->raw[j];
		if (P.qchunk0[j] < P.qchunk0[j + 1]) {
			uint32_t gc = P.qchunk0[j], idx = 0;
			for (;;) {
				for (uint32_t t = idx; t < spec_cnt[gc]; t++)
					if (fold_anchor(&f, spec_anchors[(size_t)chunk_geom(A, (uint32_t)j, gc - P.qchunk0[j]).log0 + t], border, R.threshold, &h))
						out.push_back(h);
				const BridgeRec &b = bridge[gc];
				uint32_t blk = b.block;
				for (uint32_t t = 0; t < b.n; t++) {
					Anchor a;
					if (t < BRIDGE_INLINE) {
						a = b.a[t];
					} else {
						uint32_t kk = (t - BRIDGE_INLINE) % POOL_BLOCK;
						if (kk == 0 && t != BRIDGE_INLINE) blk = pool[blk].next;
						a = pool[blk].a[kk];
					}
					if (fold_anchor(&f, a, border, R.threshold, &h)) out.push_back(h);
				}
				if (b.target == BRIDGE_END) break;
				gc = b.target;
				idx = b.idx_m;
			}
		}
		if (fold_finish(&f, qlen[j], R.threshold, &h)) out.push_back(h);
		std::vector<phylo_homology> hv;
		for (const RawHom &r : out) hv.push_back(project_homology(r, border));
		sort_and_filter(hv);
		E->filtered[j] = hv;
	}
	return E;
}

void emul_free(void *e) { delete (EmulOut *)e; }
size_t emul_count(void *e, size_t j, int filtered)
{
	EmulOut *E = (EmulOut *)e;
	return filtered ? E->filtered[j].size() : E->raw[j].size();
}
void emul_get_filtered(void *e, size_t j, phylo_homology *out)
{
	EmulOut *E = (EmulOut *)e;
	std::copy(E->filtered[j].begin(), E->filtered[j].end(), out);
}
void emul_get_raw(void *e, size_t j, uint32_t *out)
{
	EmulOut *E = (EmulOut *)e;
	for (size_t i = 0; i < E->raw[j].size(); i++) {
		out[3 * i] = E->raw[j][i].iref;
		out[3 * i + 1] = E->raw[j][i].iq;
		out[3 * i + 2] = E->raw[j][i].len;
	}
}
uint64_t emul_slow_steps(void *e) { return ((EmulOut *)e)->slow_steps; }
uint64_t emul_overruns(void *e) { return ((EmulOut *)e)->overruns; }
// how often each reason sent a step to the slow resolver since the library was loaded (lean_core.h: LeanSlowWhy)
void emul_slow_why(unsigned long long *out)
{
	for (unsigned i = 0; i < SW_COUNT; i++) out[i] = g_lean_why[i];
}
void emul_info(void *e, uint64_t out[8])
{
	EmulOut *E = (EmulOut *)e;
	out[0] = E->threshold;
	out[1] = E->k;
	out[2] = E->C;
	out[3] = E->nchunks;
	out[4] = E->steps_spec;
	out[5] = E->steps_bridge;
	out[6] = E->rounds;
	out[7] = ((uint64_t)E->error << 32) | E->pool_used;
}

// the host walk behind phylo_reference_cache_quirk, on S = seq + '#' + revcomp(seq)
int emul_cache_quirk(const uint8_t *seq, uint32_t len)
{
	const uint32_t ns = 2 * len + 1;
	std::vector<uint8_t> S((size_t)ns + 64, 0);
	memcpy(S.data(), seq, len);
	S[len] = '#';
	revcomp(seq, len, S.data() + len + 1);
	std::vector<uint32_t> SA((size_t)ns + 4, 0);
	suffix_array_u32(S.data(), ns, SA.data());
	return esa_cache_quirk(S.data(), ns, SA.data()) ? 1 : 0;
}

// host-logic entry points for CPU tests
void emul_suffix_array(const uint8_t *s, uint32_t n, uint32_t *sa) { suffix_array_u32(s, n, sa); }
// the several-core bucket sort on its own: 1 = sorted, 0 = gave up (caller would run SA-IS)
int emul_suffix_array_buckets(const uint8_t *s, uint32_t n, uint32_t *sa, uint32_t threads)
{
	std::vector<uint8_t> padded((size_t)n + 16, 0);
	memcpy(padded.data(), s, n);
	ThreadFan fan{threads};
	return suffix_array_buckets(padded.data(), n, sa, fan, threads) ? 1 : 0;
}
void emul_lcp(const uint8_t *s, uint32_t n, const uint32_t *sa, uint32_t *lcp) { lcp_kasai(s, n, sa, lcp); }
size_t emul_kmer_table(const uint8_t *s, uint32_t n, uint32_t k, uint32_t *out)
{
	std::vector<uint32_t> T;
	kmer_table(s, n, k, T);
	if (out) std::copy(T.begin(), T.end(), out);
	return T.size();
}
}

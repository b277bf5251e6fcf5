// abi_anchor.hip — phase A of process() (/root/reference/src/process.cxx:433-458): the chunk plan, the speculative
// chains, overruns and bridges (lean_kernels.hip), the fold (anchor_kernels.hip), sort + chain filter on the device
// (filter_kernels.hip) or on the host cores, and — when the call covers every genome — the projection that phase B
// starts with, queued behind it.
#include "abi_ctx.hpp"

using namespace phy;
using namespace phyabi;

extern "C" {

__global__ void compact_raw_kernel(const RawHom *__restrict__ src, const uint64_t *__restrict__ src_base,
								   const uint32_t *__restrict__ cnt, const uint64_t *__restrict__ dst_base,
								   RawHom *__restrict__ dst)
{
	const uint32_t j = blockIdx.x;
	const RawHom *s = src + src_base[j];
	RawHom *d = dst + dst_base[j];
	for (uint32_t t = threadIdx.x; t < cnt[j]; t += blockDim.x) d[t] = s[t];
}


// What the host wants to know of a phase A, written by the device straight into page-locked memory (one launch instead of
// three copies in front of phase B): h_rng's layout — [0, 2nq) the lists' ranges, [2nq] their total, [2nq + 1, 3nq + 1) the
// filter's flags, then eight of the chains' counters.
__global__ __launch_bounds__(256) void phase_a_report_kernel(const uint32_t *__restrict__ rng, const uint32_t *__restrict__ flt,
															  const uint32_t *__restrict__ misc, uint32_t nq, uint32_t *__restrict__ host_out)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < 2 * nq) host_out[t] = rng[t];
	else if (t < 3 * nq + 1) host_out[t] = flt[t - 2 * nq];
	else if (t < 3 * nq + 9) host_out[t] = misc[t - (3 * nq + 1)];
}

// What the device filter left for queries [q_begin, q_end), once the stream has been waited for and h_rng holds its ranges,
// flags and counters (see anchor_impl): the lists stay in the context's device buffer, the host knows where.
static void adopt_device_lists(phylo_ctx *c, size_t q_begin, size_t q_end, bool tail_eager)
{
	const size_t N = c->n, nq = q_end - q_begin;
	const uint32_t *hr = c->h_rng.p, *dmisc = hr + 3 * nq + 1;
	c->att_homs = c->b_homs.p;
	c->att_rng_on_device = false;
	if (c->att_begin.size() != N) {
		c->att_begin.assign(N, 0);
		c->att_count.assign(N, 0);
	}
	if (c->host_stale.size() != N) c->host_stale.assign(N, 0);
	for (size_t j = 0; j < nq; j++) {
		c->att_begin[q_begin + j] = hr[2 * j];
		c->att_count[q_begin + j] = hr[2 * j + 1] - hr[2 * j];
		c->host_stale[q_begin + j] = 1;
	}
	c->homs_staged = q_begin == 0 && q_end == N;
	c->eager_valid = tail_eager;
	c->stats["n:anchor_calls"] += 1;
	c->stats["count:query_bases"] += c->pend_total;
	c->stats["count:chunks"] += c->pend_nch;
	c->stats["count:filtered_homologies"] += (double)hr[2 * nq];
	c->stats["count:pool_blocks_used"] += dmisc[2];
	c->stats["count:overrun_runs"] += dmisc[5];
	c->stats["count:overrun_bytes_compared"] += dmisc[6];
	c->stats["anchor:chunk"] = c->pend_C;
}

int phyabi::settle_anchor(phylo_ctx *c)
{
	if (!c->anchor_pending || !c->pend_range) return 0;
	HIPOK(c, hipSetDevice(c->device));
	if (sync_stream(c)) return 1;
	const size_t qb = c->pend_qb, qe = c->pend_qe, nq = qe - qb;
	const bool stats_only = c->pend_stats_only;
	c->anchor_pending = c->pend_range = c->pend_stats_only = false;
	const uint32_t *hr = c->h_rng.p, *dmisc = hr + 3 * nq + 1;
	size_t flagged = 0;
	for (size_t j = 0; j < nq; j++) flagged += hr[2 * nq + 1 + j] != 0;
	c->stats["ms:anchor_setup"] += c->pend_t1 - c->pend_t0;
	c->stats["ms:anchor_total"] += c->pend_t2 - c->pend_t0; // (the host's part: the device's time is in the wait for the result)
	c->stats["n:anchor_calls_without_a_wait"] += 1;
	if (stats_only) { // the gathered blocks have replaced this context's view of the lists; what the flags say went round with the blocks
		c->stats["n:anchor_calls"] += 1;
		c->stats["count:query_bases"] += c->pend_total;
		c->stats["count:chunks"] += c->pend_nch;
		c->stats["count:filtered_homologies"] += (double)hr[2 * nq];
		c->stats["count:pool_blocks_used"] += dmisc[2];
		c->stats["count:overrun_runs"] += dmisc[5];
		c->stats["count:overrun_bytes_compared"] += dmisc[6];
		c->stats["anchor:chunk"] = c->pend_C;
		return 0;
	}
	if (dmisc[3]) return c->fail("phase A scratch overflow (code %u: 1 chunk log, 2 bridge pool, 3 homology buffer)", dmisc[3]);
	if (flagged) { // a list needs the host's std::sort: the long way round
		c->stats["count:anchor_block_calls_repeated"] += 1;
		return anchor_impl(c, qb, qe, 0);
	}
	adopt_device_lists(c, qb, qe, false);
	return 0;
}

// defer 1: when this call covers every genome and leaves lists and projection on the device, do not wait for its flags —
// the caller queues phase B behind it and reads them with the result (phylo_anchor_compare); anchor_pending says so.
// defer 2: a range of the queries on its way into the exchange between ranks (phylo_anchor_block_device): the block is
// written behind the filter, nothing is waited for; pend_range says so, settle_anchor reads the flags when somebody asks.
int phyabi::anchor_impl(phylo_ctx *c, size_t q_begin, size_t q_end, int defer)
{
	if (!c) return 1;
	if (settle_anchor(c)) return 1;
	c->anchor_pending = false;
	if (!c->have_ref) return c->fail("phylo_anchor: no reference set");
	if (q_begin > q_end || q_end > c->n) return c->fail("phylo_anchor: bad query range");
	HIPOK(c, hipSetDevice(c->device));
	size_t nq = q_end - q_begin;
	if (nq == 0) return 0;
	// Lists of an earlier call that still live only in this context's device buffer (about to be
	// reused) and lie outside the range computed now are read back first.  Lists attached from a
	// caller's buffer are not: that buffer is borrowed only until this call.
	if (!c->host_stale.empty() && c->att_homs == c->b_homs.p) {
		if (ensure_host_lists(c, 0, q_begin) || ensure_host_lists(c, q_end, c->n)) return 1;
	}
	double t0 = now_ms();
	const bool quirk_mode = c->nquirk > 0 && c->opt_cache_quirk != 0; // (see below, where the chains' tables are set up)

	hipStream_t st = c->stream;
	if (!c->plan_valid || c->plan_qb != q_begin || c->plan_qe != q_end) {
		// chunk plan, output capacities and their device copies: rebuilt only when the
		// query range, the genomes or the reference change
		std::vector<uint32_t> qlen(nq);
		std::vector<uint64_t> qoff(nq);
		for (size_t j = 0; j < nq; j++) {
			qlen[j] = (uint32_t)c->glen[q_begin + j];
			qoff[j] = c->goff[q_begin + j];
			// The subject is one of the queries (src/phylonium.cxx:287).  Against itself the
			// chain is one lucky anchor of the whole length (process.cxx:227-242 at q = 0), so
			// its list is written directly below; as a GPU query it would make every
			// speculative chunk compare to the end of the genome.
			if (q_begin + j == c->ref_idx) qlen[j] = 0;
		}
		// Blocks of the chain kernels per CU, for the plan and for the launch alike: three.  A chain's time is its trips
		// times the trip's duration, the speculative kernel ends with its slowest chain, and a fourth wavefront on a
		// SIMD makes every trip of the other three longer by more than the shorter chunks save (C3: 2.78 against
		// 2.86 ms) — unless the queries go through in several rounds of chunks anyway (C4: 9.90 against 10.11 ms with
		// four) or the slot table is k = 14's 4 GB, whose fetches answer slowly enough for the fourth to pay (C5:
		// 16.0 against 18.7 ms; five: 16.4).
		uint64_t plan_bases = 0;
		for (size_t j = 0; j < nq; j++) plan_bases += qlen[j];
		c->plan_nq_real = 0; // queries that have chunks (the subject among the queries has none)
		for (size_t j = 0; j < nq; j++) c->plan_nq_real += qlen[j] > 0;
		int per_cu_cap = (c->k >= 14 || plan_bases > 2500000000ull) ? 4 : 3;
#ifdef PHY_DEV_HOOKS
		if (const char *e = getenv("PHY_SPEC_PER_CU")) per_cu_cap = std::max(1, atoi(e)); // experiments
#endif
		c->plan_spec_per_cu = per_cu_cap;
		const int resident = std::min(lean_spec_resident_blocks(c->n_cu), per_cu_cap * c->n_cu);
		c->plan = plan_chunks(qlen, c->threshold, c->opt_chunk, (uint32_t)resident * 256u, (uint32_t)c->n_cu * 256u);
		if (!c->plan.C) return c->fail("phase A: more than 2^32 anchor log slots");
		const ChunkPlan &P = c->plan;
		// an emitted homology spans >= 2*threshold query positions
		c->plan_out_base.assign(nq + 1, 0);
		std::vector<uint32_t> out_cap(nq);
		uint64_t raw_total = 0;
		for (size_t j = 0; j < nq; j++) {
			c->plan_out_base[j] = raw_total;
			out_cap[j] = qlen[j] / (2 * c->threshold) + 2;
			raw_total += out_cap[j];
		}
		c->plan_out_base[nq] = raw_total;
		c->plan_raw_total = raw_total;
		uint32_t nchp = P.nchunks;
		HIPOK(c, c->a_qoff.ensure(nq));
		HIPOK(c, c->a_qlen.ensure(nq));
		HIPOK(c, c->a_qchunk0.ensure(nq + 1));
		HIPOK(c, c->a_items.ensure(nchp + 1));
		HIPOK(c, c->a_work.ensure(nchp + 1));
		HIPOK(c, c->a_qdesc.ensure(nq + 1));
		c->work_stale = true;
		HIPOK(c, c->a_chunk_query.ensure(nchp + 1));
		HIPOK(c, c->a_spec_cnt.ensure(nchp + 1));
		// one visited bit per byte of the genome buffer (chains address it by buffer offset)
		HIPOK(c, c->a_visited.ensure((c->goff[c->n - 1] + c->glen[c->n - 1]) / 32 + 8 + 128));
		HIPOK(c, c->a_misc.ensure(32)); // counters and flags
		HIPOK(c, c->a_spec_anchors.ensure(P.anchor_slots + 1));
		HIPOK(c, c->a_qanc0.ensure(nq));
		HIPOK(c, c->a_spec_exit.ensure(nchp + 1));
		HIPOK(c, c->a_bridge.ensure(nchp + 1));
		HIPOK(c, c->a_bridge_start.ensure(((size_t)nchp + 1) * LeanBridge::PACKED_WORDS));
		HIPOK(c, c->a_pool.ensure(nchp / 4 + 4096));
		HIPOK(c, c->a_raw.ensure(raw_total + 1));
		HIPOK(c, c->a_out_base.ensure(nq + 1));
		HIPOK(c, c->a_cmp_base.ensure(nq + 1));
		HIPOK(c, c->a_out_cap.ensure(nq));
		HIPOK(c, c->a_out_cnt.ensure(nq));
		HIPOK(c, hipMemcpyAsync(c->a_qoff.p, qoff.data(), nq * 8, hipMemcpyHostToDevice, st));
		HIPOK(c, hipMemcpyAsync(c->a_qlen.p, qlen.data(), nq * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, hipMemcpyAsync(c->a_qchunk0.p, P.qchunk0.data(), (nq + 1) * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, hipMemcpyAsync(c->a_qanc0.p, P.qanc0.data(), nq * 4, hipMemcpyHostToDevice, st));
		if (nchp) {
			HIPOK(c, hipMemcpyAsync(c->a_items.p, P.items.data(), (size_t)nchp * 4, hipMemcpyHostToDevice, st));
			HIPOK(c, hipMemcpyAsync(c->a_chunk_query.p, P.chunk_query.data(), (size_t)nchp * 4, hipMemcpyHostToDevice, st));
		}
		HIPOK(c, hipMemcpyAsync(c->a_out_base.p, c->plan_out_base.data(), (nq + 1) * 8, hipMemcpyHostToDevice, st));
		HIPOK(c, hipMemcpyAsync(c->a_out_cap.p, out_cap.data(), nq * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, hipStreamSynchronize(st)); // the host vectors above go out of scope
		c->plan_qb = q_begin;
		c->plan_qe = q_end;
		c->plan_valid = true;
	}
	const ChunkPlan &P = c->plan;
	const uint32_t nch = P.nchunks;
	const uint32_t pool_blocks = nch / 4 + 4096;
	uint64_t total = 0;
	for (size_t j = 0; j < nq; j++) total += c->glen[q_begin + j];
	HIPOK(c, hipMemsetAsync(c->a_misc.p, 0, 32 * 4, st));
	if (nch) { // the words of this call's queries (their genomes lie back to back in the buffer)
		// (on 256-byte boundaries: one fill instead of head, body and tail; the neighbours' bits are nobody's between two calls)
		const uint64_t w0 = c->goff[q_begin] / 32 / 64 * 64, w1 = ((c->goff[q_end - 1] + c->glen[q_end - 1]) / 32 + 1 + 63) / 64 * 64;
		// (Clearing them for the next pass on a second stream behind this pass's bridges — beside the fold, or behind the
		// projection beside the pair kernel — was measured: the fill then costs the kernel it runs beside what it saved in
		// front of the chains, C3 fold 0.18 -> 0.21 ms or pairs 0.345 -> 0.364; not kept.  Nor behind the pass's result, with
		// the host waiting for an event recorded in front of the fill — the device clearing while the host turns around:
		// the same to a hundredth of a millisecond, profiles/r05_ab_visited_clear_behind_result.txt.)
		HIPOK(c, hipMemsetAsync(c->a_visited.p + w0, 0, (size_t)(w1 - w0) * 4, st));
	}

	PhaseA A;
	A.qbase = c->d_genomes;
	A.qoff = c->a_qoff.p;
	A.qlen = c->a_qlen.p;
	A.qchunk0 = c->a_qchunk0.p;
	A.items = c->a_items.p;
	A.chunk_query = c->a_chunk_query.p;
	A.work = c->a_work.p;
	A.qdesc = c->a_qdesc.p;
	A.nchunks = nch;
	A.C = P.C;
	A.cap = P.cap;
	A.qanc0 = c->a_qanc0.p;
	A.spec_anchors = c->a_spec_anchors.p;
	A.spec_cnt = c->a_spec_cnt.p;
	A.spec_exit = c->a_spec_exit.p;
	A.visited = c->a_visited.p;
	A.bridge = c->a_bridge.p;
	A.pool = c->a_pool.p;
	A.pool_blocks = pool_blocks;
	A.bridge_start = c->a_bridge_start.p;
	A.bridge_todo = c->a_misc.p + 8;
	A.fetch = c->a_misc.p;          // [0] spec, [1] bridge
	A.pool_next = c->a_misc.p + 2;
	A.error = c->a_misc.p + 3;
	A.overrun = c->a_misc.p + 4;
	RefIndex R = {c->d_S.p, c->d_SAX.p, c->d_LCP.p, c->slot_at, c->d_T.p, c->ns, c->k, c->threshold};
	// A subject on which the reference's 6-mer cache holds over-deep intervals (esa.cxx:174-199): the reference's
	// answers there are reproduced by the lean chains' slow resolver, so every step goes through it (such subjects
	// are a few kbp in several contigs; option "cache_quirk" = 0 computes the true longest matches instead).
	LeanIndex X = {c->d_S2.p, c->d_SBAD.p, c->nsb, c->ns, c->sb_first, c->d_Q2.p, c->d_QBAD.p, c->d_qbad_off.p + q_begin,
				   (uint32_t)(c->lean_force_slow || quirk_mode), nullptr, quirk_mode ? c->d_quirk.p : nullptr, quirk_mode ? c->nquirk : 0u};
	if (c->work_stale && nch) { // the plan is new: its work order as items and descriptors
		launch_lean_work(A, X, c->a_work.p, c->a_qdesc.p, (uint32_t)nq, st);
		HIPOK(c, hipGetLastError());
		c->work_stale = false;
	}
#ifdef PHY_LEAN_TIMING
	static unsigned long long *dbg_buf = nullptr;
	const size_t dbg_words = 16 + 4 * 8192 + 64 + 16;
	if (!dbg_buf) (void)hipMalloc((void **)&dbg_buf, dbg_words * 8);
	(void)hipMemsetAsync(dbg_buf, 0, dbg_words * 8, st);
	X.dbg = dbg_buf;
#endif
	double t1 = now_ms();
#ifdef PHY_DEV_HOOKS
	const bool dbg = getenv("PHY_DEBUG_SYNC") != nullptr; // name the kernel a hang is in
#else
	const bool dbg = false;
#endif
	if (dbg) {
		hipError_t e = hipStreamSynchronize(st);
		fprintf(stderr, "[phylonium_amd] phase A set up (%s): %zu queries, %u chunks of %u, cap %u, k %u, |S| %u, threshold %u\n",
				hipGetErrorString(e), nq, nch, P.C, P.cap, c->k, c->ns, c->threshold);
	}
	auto dbg_sync = [&](const char *what) {
		if (!dbg) return;
		hipError_t e = hipStreamSynchronize(st);
		fprintf(stderr, "[phylonium_amd] %s finished at +%.1f ms (%s), chunks %u of %u positions\n", what, now_ms() - t1,
				hipGetErrorString(e), nch, P.C);
	};
	if (nch) {
		{
			KernelSpan s(c, "anchor_spec");
			launch_lean_spec(A, R, X, c->n_cu, st, c->opt_spec_blocks ? (int)c->opt_spec_blocks : c->plan_spec_per_cu * c->n_cu);
		}
		dbg_sync("anchor_spec");
		{
			KernelSpan s(c, "anchor_overruns");
			launch_lean_overruns(A, R, (uint32_t)nq, st);
		}
	}
	// Sort + chain filter: on the device unless the host is asked for (option "filter" = 1).  Round 1 sent calls
	// with fewer than 128 queries to the host pool (the device's dependent scan took ~0.4 ms whatever the
	// number); stretch by stretch a list takes ~60 us there, and long lists have a kernel of their own.
	const bool device_filter = c->filter_mode != 1;
	const bool full = q_begin == 0 && q_end == c->n;
	const bool tail_eager = device_filter && full && c->backend == 0;
	const uint32_t ref_local = (c->ref_idx >= q_begin && c->ref_idx < q_end) ? (uint32_t)(c->ref_idx - q_begin) : 0xffffffffu;
	const uint32_t tsz_q = project_genomes_per_tile();
	// queries of tens of Mbp leave more entries than a block's LDS holds (~330 per Mbp): their lists go through the
	// long-list kernel and its scratch slots; with shorter queries a list that long is an oddity and goes to the host
	bool long_lists = false;
	if (device_filter && c->opt_filter_kernel == 0) {
		uint64_t longest = 0;
		for (size_t j = 0; j < nq; j++)
			if (q_begin + j != c->ref_idx) longest = std::max<uint64_t>(longest, c->glen[q_begin + j]);
		long_lists = longest > 6000000;
		if (long_lists) HIPOK(c, c->a_long.ensure(long_filter_scratch_bytes()));
	}
	Pileup TP;
	if (device_filter) {
		HIPOK(c, c->b_homs.ensure(c->plan_raw_total + nq + 1));
		HIPOK(c, c->b_hom_rng.ensure(2 * std::max(nq, c->n)));
		HIPOK(c, c->a_flt.ensure(nq + 1));
		HIPOK(c, c->h_rng.ensure(3 * nq + 16));
		if (!nch) HIPOK(c, hipMemsetAsync(c->a_flt.p, 0, 4, st)); // (else: the bridges' prepare kernel zeroes it, below)
		if (tail_eager) {
			if (make_pileup(c, 0, 1, &TP)) return 1;
			HIPOK(c, c->b_flag.ensure(8));
			HIPOK(c, c->b_first.ensure(project_index_entries(TP) + 1));
			if (!nch) HIPOK(c, hipMemsetAsync(c->b_flag.p, 0, 16, st));
			c->eager_five = c->pileup_five && c->opt_pairs_kernel != 0; // (the matrix-core path lists the '!' instead: compare_pileup)
			HIPOK(c, c->b_bang.ensure(2 * (size_t)c->bang_cap + 2));
		}
	}
	{
		hipStream_t sg = st;
		const uint32_t j0 = 0, j1 = (uint32_t)nq;
		if (nch) {
			KernelSpan s(c, "anchor_bridge", sg);
			// (the counters of what follows are zeroed on the way: the fold's list lengths, the filter's total, the projection's flags)
			BridgeZero Z = {{c->a_out_cnt.p + j0, device_filter ? c->a_flt.p : nullptr, (device_filter && tail_eager) ? c->b_flag.p : nullptr},
							{j1 - j0, device_filter ? 1u : 0u, (device_filter && tail_eager) ? 4u : 0u}};
			launch_lean_bridge(A, R, X, c->n_cu, st, Z);
		}
		{
			KernelSpan s(c, "anchor_fold", sg);
			// Blocks per query.  A block's time is its windows' walks (a latency chain every block of the query
			// repeats) plus its share of the anchors; with a block per CU or more there is nothing to gain from
			// splitting (measured, C3's 256 queries with two blocks each: 0.17 -> 0.21 ms), with a handful of queries
			// the idle CUs take a part each (c2like's 29 queries: 0.167 -> 0.100 ms; C5's 64: 2.65 -> 2.28 ms)
			// (counted without the queries that have no chunks — the subject among its queries: their blocks leave at once.  A
			// rank of eight with the subject in its block has 129 queries, 128 of them real: two blocks each fit the 256 CUs,
			// 0.12 ms; counted with the subject it was one block each, 0.19 ms — on the rank that also fetches the result.)
			uint32_t fold_nb = c->opt_fold_blocks;
			const uint32_t nq_real = std::max<uint32_t>(1, (uint32_t)c->plan_nq_real);
			// (a power of two up to sixteen — round 6: eight queries of 100 Mbp, a rank's eighth of C5: 8 blocks each 0.746 ms, 12: 0.735,
			// 16: 0.607, 24: 0.604, 32: 1.14; c5s's sixteen queries 0.238 -> 0.201; four queries of 5 Mbp the same)
			if (!fold_nb) {
				fold_nb = 1;
				while (2 * fold_nb <= 16 && 2 * fold_nb * nq_real <= (uint32_t)c->n_cu) fold_nb *= 2;
			}
			launch_fold(A, j0, j1, c->L, c->threshold, c->a_raw.p, c->a_out_base.p, c->a_out_cap.p, c->a_out_cnt.p, sg, fold_nb, nch == 0);
		}
		if (device_filter) {
			// reverseEh + sort + filter_overlaps_max on the device (filter_kernels.hip).  The lists stay
			// there in the 16-byte device form, the projection (phase B's first kernel, for the whole
			// reference = part 0 of 1) follows at once when this call covers all genomes, and the host
			// reads a list back only when somebody asks for it.  A query whose list has two entries
			// with the same projected start, or more entries than the kernel holds, is flagged: then
			// everything below runs on the host as it always did.
			{
				KernelSpan s(c, "anchor_filter", sg);
				launch_sort_filter(c->a_raw.p, c->a_out_base.p, c->a_out_cnt.p, j0, j1, c->L, c->threshold, ref_local, c->b_homs.p,
								   c->b_hom_rng.p, c->a_flt.p, c->a_flt.p + 1, sg, c->opt_filter_kernel, long_lists ? 1 : 0);
				if (long_lists)
					launch_sort_filter_long(c->a_raw.p, c->a_out_base.p, c->a_out_cnt.p, j0, j1, c->L, c->b_homs.p, c->b_hom_rng.p,
											c->a_flt.p, c->a_flt.p + 1, c->a_long.p, c->a_misc.p + 7, sg);
			}
			if (tail_eager && j1 > j0) {
				launch_tile_index(TP, query_src(c), c->b_homs.p, c->b_hom_rng.p, c->b_first.p, j0, j1, sg);
				KernelSpan s(c, c->eager_five ? "pileup_project5" : "pileup_project", sg);
				launch_project(TP, c->eager_five, query_src(c), c->b_homs.p, c->b_hom_rng.p, c->b_first.p, c->b_flag.p,
							   0, TP.Npad / tsz_q, sg, c->b_bang.p, c->bang_cap, c->proj_resident);
			}
		}
	}
	dbg_sync("anchor tail (bridge, fold, filter, projection)");
#ifdef PHY_LEAN_TIMING
	{
		std::vector<unsigned long long> hv(dbg_words);
		unsigned long long *h = hv.data();
		(void)hipStreamSynchronize(st);
		(void)hipMemcpy(h, X.dbg, dbg_words * 8, hipMemcpyDeviceToHost);
		for (int m = 0; m < 2; m++) {
			const unsigned long long *pt = h + 16 + 4 * 8192 + m * 16, trips = h[16 + 4 * 8192 + 32 + m];
			fprintf(stderr, "[lean timing] mode %d trips %llu; share of trips with a lane in STEP %.3f SEARCH %.3f SCAN %.3f EXT %.3f REFILL %.3f SLOW %.3f SLOWEXT %.3f, lucky STEP %.3f; lanes per trip: STEP %.1f SEARCH %.2f SCAN %.2f EXT %.1f REFILL %.2f SLOW %.3f\n",
					m, trips, (double)pt[0] / trips, (double)pt[1] / trips, (double)pt[2] / trips, (double)pt[3] / trips, (double)pt[4] / trips,
					(double)pt[5] / trips, (double)pt[6] / trips, (double)pt[7] / trips, (double)pt[8] / trips, (double)pt[9] / trips,
					(double)pt[10] / trips, (double)pt[11] / trips, (double)pt[12] / trips, (double)pt[13] / trips);
		}
		for (int m = 0; m < 2; m++) {
			const unsigned long long *e = h + 16 + 4 * 8192 + 34 + m * 6;
			fprintf(stderr, "[lean timing] mode %d EXT lane-trips: first of a lucky check %llu, first of a candidate %llu, later %llu; of the first ones: match < 32 bases %llu, < 48 bases %llu, on to SEARCH/SLOW %llu\n",
					m, e[0], e[1], e[2], e[3], e[4], e[5]);
		}
		{
			const unsigned long long *b = h + 16 + 4 * 8192 + 64;
			fprintf(stderr, "[lean timing] bridges walked, by steps: 1: %llu  2: %llu  3-4: %llu  5-8: %llu  9-16: %llu  17-32: %llu  33-64: %llu  65+: %llu; the longest %llu\n",
					b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9]);
			fprintf(stderr, "[lean timing] look-ahead: trips with a leader %llu (helpers %.1f, of them plain %.1f per trip), trips in which the leader's own step was plain %llu, hops gained %llu\n",
					b[12], b[12] ? (double)b[13] / b[12] : 0.0, b[12] ? (double)b[14] / b[12] : 0.0, b[10], b[11]);
		}
		for (int m = 0; m < 2; m++) { // the wavefronts' lifetimes (100 MHz counter) and trips
			std::vector<double> dur, ends;
			std::vector<unsigned long long> tr;
			unsigned long long t0 = ~0ull;
			for (size_t w = 0; w < 4096; w++) {
				const unsigned long long *r = h + 16 + 4 * (m * 4096 + w);
				if (r[1]) t0 = std::min(t0, r[0]);
			}
			for (size_t w = 0; w < 4096; w++) {
				const unsigned long long *r = h + 16 + 4 * (m * 4096 + w);
				if (!r[1]) continue;
				dur.push_back((double)(r[1] - r[0]) / 100.0);
				ends.push_back((double)(r[1] - t0) / 100.0);
				tr.push_back(r[2]);
			}
			if (dur.empty()) continue;
			std::sort(dur.begin(), dur.end());
			std::sort(ends.begin(), ends.end());
			std::sort(tr.begin(), tr.end());
			const size_t n = dur.size();
			fprintf(stderr, "[lean timing] mode %d: %zu wavefronts; lifetime us min %.0f median %.0f p90 %.0f max %.0f; end (from the first start) us median %.0f p90 %.0f p99 %.0f max %.0f; trips min %llu median %llu p90 %llu max %llu\n",
					m, n, dur[0], dur[n / 2], dur[n * 9 / 10], dur[n - 1], ends[n / 2], ends[n * 9 / 10], ends[n * 99 / 100], ends[n - 1], tr[0], tr[n / 2], tr[n * 9 / 10], tr[n - 1]);
		}
		if (const char *wf = getenv("PHY_LEAN_WAVES_OUT")) {
			if (FILE *f = fopen(wf, "w")) { // the last call's speculative wavefronts: start, end (10 ns), trips, query
				for (size_t w = 0; w < 8192; w++)
					if (h[16 + 4 * w + 1]) fprintf(f, "%llu %llu %llu %llu\n", h[16 + 4 * w], h[16 + 4 * w + 1], h[16 + 4 * w + 2], h[16 + 4 * w + 3]);
				fclose(f);
			}
		}
		for (int m = 0; m < 2; m++)
			fprintf(stderr, "[lean timing] mode %d waves %llu  Mcycles: bookkeeping %.1f  phase+address %.1f  loads %.1f  digest %.1f  slow %.1f; resolves %llu, long compares %llu\n",
					m, h[m * 8 + 6], h[m * 8 + 0] / 1e6, h[m * 8 + 1] / 1e6, h[m * 8 + 2] / 1e6, h[m * 8 + 3] / 1e6, h[m * 8 + 4] / 1e6,
					h[m * 8 + 5] & 0xffffffffull, h[m * 8 + 5] >> 32);
	}
#endif
	HIPOK(c, hipGetLastError());
	c->homs_staged = false;
	c->eager_valid = false;
	c->att_homs = nullptr; // an attached buffer is only borrowed until the next phase A
	c->att_rng_on_device = false;
	c->host_stale.clear();
	if (device_filter) {
		uint32_t *hr = c->h_rng.p; // [0, 2nq) ranges, [2nq] the lists' total, [2nq + 1, 3nq + 1) flags (a_flt as it lies), then the misc words
		if (defer != 2) { // (defer 2: the kernel that writes the exchange block writes these words as well)
			uint32_t *hr_dev = nullptr;
			HIPOK(c, hipHostGetDevicePointer((void **)&hr_dev, hr, 0));
			hipLaunchKernelGGL(phase_a_report_kernel, dim3((uint32_t)((3 * nq + 9 + 255) / 256)), dim3(256), 0, st, (const uint32_t *)c->b_hom_rng.p,
							   (const uint32_t *)c->a_flt.p, (const uint32_t *)c->a_misc.p, (uint32_t)nq, hr_dev);
		}
		HIPOK(c, hipGetLastError());
		c->pend_t0 = t0, c->pend_t1 = t1, c->pend_total = (double)total, c->pend_nch = nch, c->pend_C = P.C;
		if (defer == 1 && tail_eager) { // (tail_eager: all genomes, device filter, projection queued)
			c->pend_t2 = now_ms();
			c->att_homs = c->b_homs.p;
			c->homs_staged = true;
			c->eager_valid = true;
			c->anchor_pending = true;
			return 0;
		}
		if (defer == 2) {
			// the block behind the filter; what the filter's flags and the chains' counters say (a list that needs the host's
			// std::sort, scratch that overflowed) rides in its header to every rank (block_export_kernel)
			if (queue_block_export(c, nq, c->xb_block, c->xb_maxq, c->xb_cap, c->a_flt.p + 1, c->a_misc.p, hr)) return 1;
			c->pend_t2 = now_ms();
			c->att_homs = c->b_homs.p;
			c->pend_range = true;
			c->pend_stats_only = false;
			c->pend_qb = q_begin;
			c->pend_qe = q_end;
			c->anchor_pending = true;
			return 0;
		}
		if (sync_stream(c)) return 1;
		double t2d = now_ms();
		const uint32_t *dmisc = hr + 3 * nq + 1;
		if (dmisc[3]) return c->fail("phase A scratch overflow (code %u: 1 chunk log, 2 bridge pool, 3 homology buffer)", dmisc[3]);
		size_t flagged = 0;
		for (size_t j = 0; j < nq; j++) flagged += hr[2 * nq + 1 + j] != 0;
		if (!flagged) {
			c->host_stale.assign(c->n, 0);
			adopt_device_lists(c, q_begin, q_end, tail_eager);
			c->stats["ms:anchor_setup"] += t1 - t0;
			c->stats["ms:anchor_gpu"] += t2d - t1;
			c->stats["ms:anchor_total"] += now_ms() - t0;
			return 0;
		}
		c->stats["count:queries_left_to_the_host"] += (double)flagged;
	}
	HIPOK(c, c->h_cnt.ensure(nq + 8));
	uint32_t *cnt = c->h_cnt.p, *misc = c->h_cnt.p + nq;
	HIPOK(c, hipMemcpyAsync(cnt, c->a_out_cnt.p, nq * 4, hipMemcpyDeviceToHost, st));
	HIPOK(c, hipMemcpyAsync(misc, c->a_misc.p, 32, hipMemcpyDeviceToHost, st));
	if (sync_stream(c)) return 1;
	double t2 = now_ms();
	if (misc[3]) return c->fail("phase A scratch overflow (code %u: 1 chunk log, 2 bridge pool, 3 homology buffer)", misc[3]);

	std::vector<uint64_t> cbase(nq + 1);
	uint64_t ctot = 0;
	for (size_t j = 0; j < nq; j++) {
		cbase[j] = ctot;
		ctot += cnt[j];
	}
	cbase[nq] = ctot;
	HIPOK(c, c->h_raw.ensure(ctot + 1));
	// (the workers read and, for lists that arrive out of query order, reorder c->h_raw in place)
	if (ctot) {
		HIPOK(c, c->a_raw_compact.ensure(ctot));
		HIPOK(c, hipMemcpyAsync(c->a_cmp_base.p, cbase.data(), (nq + 1) * 8, hipMemcpyHostToDevice, st));
		{
			KernelSpan s(c, "anchor_compact");
			hipLaunchKernelGGL(compact_raw_kernel, dim3((uint32_t)nq), dim3(256), 0, st, c->a_raw.p, c->a_out_base.p,
							   c->a_out_cnt.p, c->a_cmp_base.p, c->a_raw_compact.p);
		}
		HIPOK(c, hipMemcpyAsync(c->h_raw.p, c->a_raw_compact.p, ctot * sizeof(RawHom), hipMemcpyDeviceToHost, st));
		if (sync_stream(c)) return 1;
	}
	double t3 = now_ms();
	// reverseEh + std::sort + filter_overlaps_max on the host cores (process.cxx:438-443)
	uint64_t border = c->L;
	std::atomic<uint32_t> tie_lists{0};
	// When this call makes every genome's list, phase B's device copy of them is staged
	// here as well: a worker writes its list in the 16-byte device form into pinned
	// memory (slot cbase[j] + j: the raw count bounds the filtered one, and the self
	// query keeps one entry of zero raw ones) and counts its group down; the calling
	// thread sends every finished group off — records and ranges go up through the
	// copy stream, and the projection of those genomes (phase B's first kernel, for
	// the whole reference = part 0 of 1) starts behind them — so upload and projection
	// run while the other lists are still being sorted.
	const bool stage = q_begin == 0 && q_end == c->n && nq > 0;
	c->homs_staged = false;
	c->att_homs = nullptr; // an attached buffer is only borrowed until the next phase A
	c->att_rng_on_device = false;
	c->host_stale.clear();
	// genomes per group: a whole number of projection tiles — three, or an eighth of all of them
	// (measured on C3 and C4: every group pays ~30 us of hand-over between the copy engine and
	// the compute queue, one big group overlaps nothing)
	const size_t tsz = project_genomes_per_tile();
	const size_t gsz = tsz * std::max<size_t>(3, ((nq + tsz - 1) / tsz) / 8);
	const size_t ngroups = stage ? (nq + gsz - 1) / gsz : 0;
	std::vector<std::atomic<uint32_t>> group_left(ngroups);
	std::mutex group_m; // the calling thread sleeps until a group is complete (it used to spin: under the boxes' CPU-time
	std::condition_variable group_cv; // quota a spinning thread takes time from the workers it is waiting for)
	const bool eager = stage && c->backend == 0;
	c->eager_valid = false;
	Pileup EP;
	std::atomic<int> stage_err{0};
	uint32_t *rng = nullptr;
	DevHom *dh = nullptr;
	if (stage) {
		HIPOK(c, c->h_devhom.ensure(ctot + nq + 1));
		HIPOK(c, c->h_rng.ensure(2 * nq));
		HIPOK(c, c->b_homs.ensure(ctot + nq + 1));
		HIPOK(c, c->b_hom_rng.ensure(2 * nq));
		rng = c->h_rng.p;
		dh = c->h_devhom.p;
		for (size_t g = 0; g < ngroups; g++) group_left[g] = (uint32_t)(std::min(nq, (g + 1) * gsz) - g * gsz);
		while (c->copy_events.size() < ngroups) {
			hipEvent_t e;
			HIPOK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
			c->copy_events.push_back(e);
		}
		if (eager) {
			if (make_pileup(c, 0, 1, &EP)) return 1;
			HIPOK(c, c->b_flag.ensure(8));
			HIPOK(c, c->b_first.ensure(project_index_entries(EP) + 1));
			HIPOK(c, hipMemsetAsync(c->b_flag.p, 0, 16, st));
			c->eager_five = c->pileup_five && c->opt_pairs_kernel != 0; // (the matrix-core path lists the '!' instead: compare_pileup)
			HIPOK(c, c->b_bang.ensure(2 * (size_t)c->bang_cap + 2));
		}
	}
	auto stage_list = [&](size_t j, const std::vector<phylo_homology> &list) {
		const size_t o = cbase[j] + j;
		for (size_t t = 0; t < list.size(); t++)
			dh[o + t] = DevHom{(uint32_t)list[t].index_reference_projected, (uint32_t)list[t].index_query,
							   (uint32_t)list[t].length, (uint32_t)list[t].direction};
		rng[2 * j] = (uint32_t)o;
		rng[2 * j + 1] = (uint32_t)(o + list.size());
		if (group_left[j / gsz].fetch_sub(1, std::memory_order_acq_rel) == 1) {
			std::lock_guard<std::mutex> lk(group_m);
			group_cv.notify_all();
		}
	};
	double t_send_done = 0;
	auto send_groups = [&]() {
		for (size_t g = 0; g < ngroups; g++) {
			if (group_left[g].load(std::memory_order_acquire) != 0) {
				std::unique_lock<std::mutex> lk(group_m);
				group_cv.wait(lk, [&] { return group_left[g].load(std::memory_order_acquire) == 0; });
			}
			const size_t j0 = g * gsz, j1 = std::min(nq, (g + 1) * gsz);
			const size_t o0 = cbase[j0] + j0, o1 = cbase[j1] + j1;
			// the upload goes through the copy stream (DMA engine), so group g+1 travels while
			// group g is being projected; an event orders the group's kernels after its upload
			if (hipMemcpyAsync(c->b_homs.p + o0, dh + o0, (o1 - o0) * sizeof(DevHom), hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
				hipMemcpyAsync(c->b_hom_rng.p + 2 * j0, rng + 2 * j0, (j1 - j0) * 8, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
				hipEventRecord(c->copy_events[g], c->copy_stream) != hipSuccess ||
				hipStreamWaitEvent(st, c->copy_events[g], 0) != hipSuccess) {
				stage_err = 1;
			} else if (eager) {
				launch_tile_index(EP, query_src(c), c->b_homs.p, c->b_hom_rng.p, c->b_first.p, (uint32_t)j0, (uint32_t)j1, st);
				KernelSpan s(c, c->eager_five ? "pileup_project5" : "pileup_project");
				launch_project(EP, c->eager_five, query_src(c), c->b_homs.p, c->b_hom_rng.p, c->b_first.p,
							   c->b_flag.p, (uint32_t)(j0 / tsz), g + 1 == ngroups ? EP.Npad / (uint32_t)tsz : (uint32_t)(j1 / tsz), st, c->b_bang.p, c->bang_cap, c->proj_resident);
			}
		}
		t_send_done = now_ms();
	};
	std::unique_ptr<KernelSpan> stage_span; // GPU-side time from here until the last group's projection is done
	if (stage) stage_span.reset(new KernelSpan(c, "stage_all"));
	workers(c).run(nq, [&](size_t j) {
		std::vector<phylo_homology> &dst = c->homs[q_begin + j];
		if (q_begin + j == c->ref_idx) {
			// anchor_homologies(ref, threshold, subject): homology(0, 0, L), pushed iff
			// last_length / 2 >= threshold (process.cxx:285-292)
			dst.clear();
			if (border / 2 >= c->threshold) dst.push_back(project_homology(RawHom{0, 0, (uint32_t)border}, border));
			if (stage) stage_list(j, dst);
			return;
		}
		static thread_local SortFilterScratch scratch;
		static thread_local std::vector<uint32_t> kept;
		RawHom *r = c->h_raw.p + cbase[j];
		const size_t m = cnt[j];
		// several fold blocks per query hand a list over in the order they got their slots: back to query order,
		// which is what the reference's std::sort is given (process.cxx:438; query positions ascend along a chain)
		if (!std::is_sorted(r, r + m, [](const RawHom &a, const RawHom &b) { return a.iq < b.iq; }))
			std::sort(r, r + m, [](const RawHom &a, const RawHom &b) { return a.iq < b.iq; });
		auto get = [&](size_t i, uint64_t *start, uint64_t *len) {
			*len = r[i].len;
			*start = r[i].iref >= border ? 2 * border + 1 - r[i].len - r[i].iref : r[i].iref;
		};
		if (sort_filter_order(m, get, scratch, kept)) {
			dst.resize(kept.size());
			for (size_t t = 0; t < kept.size(); t++) dst[t] = project_homology(r[kept[t]], border);
		} else { // equal starts: only std::sort on the structs reproduces the reference's order
			std::vector<phylo_homology> hv(m);
			for (size_t t = 0; t < m; t++) hv[t] = project_homology(r[t], border);
			sort_and_filter(hv);
			dst = std::move(hv);
			tie_lists.fetch_add(1, std::memory_order_relaxed);
		}
		if (stage) stage_list(j, dst);
	}, stage ? std::function<void()>(send_groups) : std::function<void()>());
	stage_span.reset();
	if (stage) {
		if (stage_err) return c->fail("staging the homology lists on the device failed");
		HIPOK(c, hipGetLastError());
		c->stats["ms:stage_send_done"] += t_send_done - t3;
		c->homs_staged = true;
		c->eager_valid = eager;
	}
	double t4 = now_ms();
	c->stats["ms:anchor_setup"] += t1 - t0;
	c->stats["ms:anchor_gpu"] += t2 - t1;
	c->stats["ms:anchor_copyback"] += t3 - t2;
	c->stats["ms:host_sort_filter"] += t4 - t3;
	c->stats["ms:anchor_total"] += t4 - t0;
	c->stats["n:anchor_calls"] += 1;
	c->stats["count:query_bases"] += (double)total;
	c->stats["count:chunks"] += nch;
	c->stats["count:raw_homologies"] += (double)ctot;
	c->stats["count:lists_with_equal_starts"] += (double)tie_lists.load();
	c->stats["count:pool_blocks_used"] += misc[2];
	c->stats["count:overrun_runs"] += misc[5];
	c->stats["count:overrun_bytes_compared"] += misc[6];
	(void)pool_blocks;
	c->stats["anchor:chunk"] = P.C;
	return 0;
}

int phylo_anchor(phylo_ctx *c, size_t q_begin, size_t q_end) { return anchor_impl(c, q_begin, q_end, 0); }

// Phase A of a rank's block of queries with its exchange block written behind it, nothing waited for: the caller's
// all-gather goes straight behind it on the stream.  (With the host's sort + filter asked for, option "filter" = 1,
// the lists pass through the host anyway: then this is phylo_anchor + phylo_export_block_device.)
int phylo_anchor_block_device(phylo_ctx *c, size_t q_begin, size_t q_end, void *dev_block, size_t max_queries, size_t cap_records)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !dev_block) return c->fail("phylo_anchor_block_device: bad arguments");
	const size_t nq = q_end - q_begin;
	if (max_queries < nq || max_queries % 4 || max_queries == 0) return c->fail("phylo_anchor_block_device: max_queries must be a multiple of 4 and hold the block's queries");
	if (4 + max_queries + 4 * cap_records >= 0xffffffffull) return c->fail("phylo_anchor_block_device: block too large");
	if (c->filter_mode == 1 || nq == 0) {
		if (anchor_impl(c, q_begin, q_end, 0)) return 1;
		return phylo_export_block_device(c, q_begin, q_end, dev_block, max_queries, cap_records);
	}
	c->xb_block = dev_block;
	c->xb_maxq = max_queries;
	c->xb_cap = cap_records;
	return anchor_impl(c, q_begin, q_end, 2);
}

} // extern "C"

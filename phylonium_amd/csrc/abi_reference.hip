// abi_reference.hip — `esa ref(subject)` + threshold (/root/reference/src/process.cxx:413-417, src/esa.cxx:69-81):
// S = subject + '#' + reverse complement, its suffix array (the caller's, the device's or the host cores'), LCP, the
// k-mer table T, the SAX records and the k-mer slots the chain kernels gather from, the packed view of S, and the
// check for the reference's 6-mer-cache quirk (esa.cxx:174-199).  Once per reference, outside the metric.
#include "abi_ctx.hpp"

using namespace phy;
using namespace phyabi;

extern "C" {

// One 16-byte slot per k-mer (anchor_core.h: slot_make), from T and the SAX records.  One thread per slot.
__global__ __launch_bounds__(256) void build_slots_kernel(const uint32_t *__restrict__ T, const U4 *__restrict__ sax,
														   uint32_t n, uint32_t k, uint64_t codes, U4 *__restrict__ slot)
{
	const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= codes) return;
	slot[c] = slot_make(c, k, T[c], T[c + 1], n, [&](uint32_t r) { return sax[r]; });
}

int phylo_set_reference(phylo_ctx *c, size_t ref_idx, const int64_t *sa, size_t threshold)
{
	if (!c) return 1;
	drop_pending_anchor(c);
	if (ref_idx >= c->n) return c->fail("reference index %zu out of range (n=%zu)", ref_idx, c->n);
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	uint64_t L = c->glen[ref_idx];
	if (2 * L + 1 >= 0x7fffffffull) return c->fail("reference too long: 2L+1 must be < 2^31 (src/esa.cxx:374-375)");
	if (L == 0) return c->fail("reference genome is empty");
	uint32_t ns = (uint32_t)(2 * L + 1);
	// S = subject + '#' + reverse complement is made where the subject is — on the device — together with the GC
	// count the threshold needs; the host gets a copy only for the steps that walk it there (its own suffix sorters,
	// Kasai's LCP for repeats beyond the clip, the exact 6-mer-cache check)
	std::vector<uint8_t> S;
	std::vector<uint32_t> SA; // +4: tables are read 16 bytes at a time
	hipStream_t st = c->stream;
	HIPOK(c, c->d_S.ensure((size_t)ns + 64));
	HIPOK(c, c->d_SA.ensure((size_t)ns + 4));
	HIPOK(c, c->a_misc.ensure(16));
	DevBuf<uint32_t> d_next;
	HIPOK(c, d_next.ensure(next_bytes_entries() + 4));
	HIPOK(c, hipMemsetAsync(c->a_misc.p, 0, 64, st));
	HIPOK(c, hipMemsetAsync(d_next.p, 0, (next_bytes_entries() + 4) * 4, st));
	launch_build_subject(c->d_genomes + c->goff[ref_idx], (uint32_t)L, c->d_S.p, (unsigned long long *)(c->a_misc.p + 2), st);
	launch_next_bytes(c->d_S.p, ns, d_next.p, st);
	unsigned long long gc_count = 0;
	std::vector<uint32_t> next_masks(next_bytes_entries());
	HIPOK(c, hipMemcpyAsync(&gc_count, c->a_misc.p + 2, 8, hipMemcpyDeviceToHost, st));
	HIPOK(c, hipMemcpyAsync(next_masks.data(), d_next.p, next_masks.size() * 4, hipMemcpyDeviceToHost, st));
	HIPOK(c, hipStreamSynchronize(st));
	d_next.release();
	auto host_S = [&]() -> int {
		if (!S.empty()) return 0;
		S.assign((size_t)ns + 64, 0);
		HIPOK(c, hipMemcpy(S.data(), c->d_S.p, (size_t)ns, hipMemcpyDeviceToHost));
		return 0;
	};
	double t1 = now_ms();
	bool sa_on_device = false; // the array exists in d_SA only; `SA` is fetched if a host step asks for it
	uint32_t sa_rounds = 0;
	if (!sa && c->opt_sa_builder == 1) {
		// esa.cxx:74's divsufsort64, on the device (sa_kernels.hip)
		DevBuf<uint8_t> scratch;
		if (scratch.ensure(suffix_array_scratch_bytes(ns)) == hipSuccess) { // ~41 bytes per suffix
			HIPOK(c, hipMemsetAsync(c->d_SA.p + ns, 0, 16, st));
			const int rc = device_suffix_array(c->d_S.p, ns, c->d_SA.p, scratch.p, &sa_rounds, st);
			HIPOK(c, hipStreamSynchronize(st));
			scratch.release();
			if (rc == 2) { // a library primitive refused (e.g. its scratch requirement): the host builders take over
				c->stats["ref:sa_device_error"] = (double)hipGetLastError();
				(void)hipStreamSynchronize(st);
			}
			sa_on_device = rc == 0; // rc == 1: a byte outside ! # A C G T — the host builders order any bytes
#ifdef PHY_DEV_HOOKS
			if (sa_on_device && ns > 8 && getenv("PHYLONIUM_AMD_TEST_CORRUPT_SA")) { // tests: the device's array damaged — the check must catch it
				const uint32_t wrong[2] = {0xfffffff0u, 3u};
				HIPOK(c, hipMemcpy(c->d_SA.p + ns / 2, wrong, 8, hipMemcpyHostToDevice));
			}
#endif
		} else {
			(void)hipGetLastError(); // no room for the working set next to the genomes: the host cores sort
		}
	}
	auto host_sa = [&]() -> int { // the array on the host, for the steps that walk it there
		if (!SA.empty()) return 0;
		SA.assign((size_t)ns + 4, 0);
		HIPOK(c, hipMemcpy(SA.data(), c->d_SA.p, (size_t)ns * 4, hipMemcpyDeviceToHost));
		return 0;
	};
	if (!sa_on_device) {
		SA.assign((size_t)ns + 4, 0);
		if (sa) {
			for (uint32_t i = 0; i < ns; i++) {
				if (sa[i] < 0 || sa[i] >= (int64_t)ns) return c->fail("suffix array entry %u out of range", i);
				SA[i] = (uint32_t)sa[i];
			}
		} else {
			// host cores (north star); everything below is on the device
			if (host_S()) return 1;
			WorkerPool &pool = workers(c);
			auto par = [&](size_t nt, const std::function<void(size_t)> &f) { pool.run(nt, f); };
			suffix_array_u32_par(S.data(), ns, SA.data(), par, std::max<size_t>(1, pool.size()));
		}
		HIPOK(c, hipMemcpyAsync(c->d_SA.p, SA.data(), SA.size() * 4, hipMemcpyHostToDevice, st));
	}
	double t2 = now_ms();
	uint32_t k = c->opt_kmer ? c->opt_kmer : choose_k(ns);
	if (threshold == 0) threshold = min_anchor_length(0.025, (double)gc_count / (double)L, ns); // gc_content, sequence.cxx:152-165
	const uint64_t codes = (uint64_t)1 << (2 * k);
	{
		const double ta = now_ms();
		HIPOK(c, c->d_SAX.ensure((size_t)ns + 4));
		HIPOK(c, c->d_LCP.ensure((size_t)ns + 1 + 4));
		HIPOK(c, c->d_T.ensure(codes + 1 + 4 + kmer_table_scratch(k)));
		size_t slot_pad = 0;
#ifdef PHY_DEV_HOOKS
		if (const char *e = getenv("PHY_SLOT_ALIGN_GB")) slot_pad = ((size_t)std::max(0, atoi(e)) << 30) / sizeof(U4); // experiments: the table on that boundary
#endif
		HIPOK(c, c->d_SLOT.ensure(codes + 8 + slot_pad)); // (+8: the chain kernels' load batch reads up to 64 bytes from a slot's address)
		c->slot_at = c->d_SLOT.p;
		if (slot_pad) c->slot_at = (U4 *)(((uintptr_t)c->d_SLOT.p + slot_pad * sizeof(U4) - 1) / (slot_pad * sizeof(U4)) * (slot_pad * sizeof(U4)));
		c->stats["ms:ref_alloc"] += now_ms() - ta; // hipMalloc of tens of GB stalls when other processes have just released as much (round 2: one C5 run waited 2.4 s there)
	}
	HIPOK(c, hipMemsetAsync(c->a_misc.p, 0, 64, st));
	HIPOK(c, hipMemsetAsync(c->d_LCP.p, 0, ((size_t)ns + 1 + 4) * 4, st));
	// LCP by direct comparison of neighbouring suffixes, capped at the 16-bit clip of the
	// SAX records; only a repeat of >= 64 kbp needs the exact values (host, Kasai)
	launch_lcp(c->d_S.p, c->d_SA.p, ns, 0xffffu, c->d_LCP.p, c->a_misc.p, st);
	uint32_t lcp_words[2] = {0, 0}; // {ranks that reached the clip, the array is not the suffix array of S}
	HIPOK(c, hipMemcpyAsync(lcp_words, c->a_misc.p, 8, hipMemcpyDeviceToHost, st));
	HIPOK(c, hipStreamSynchronize(st));
	if (lcp_words[1]) {
		// The LCP kernel compares every suffix with its successor and so proves the array — or not.  A caller's array that
		// fails is the caller's error; the device builder's (seen once in thousands of runs with eight or more processes
		// starting on one GPU at the same moment, profiles/EXPERIMENTS.md round 6; never otherwise) is built again on the host cores.
		if (!sa_on_device) return c->fail("phylo_set_reference: the suffix array is not the suffix array of S = subject + '#' + reverse complement");
		c->stats["ref:sa_device_rejected"] += 1;
		sa_on_device = false;
		if (host_S()) return 1;
		SA.assign((size_t)ns + 4, 0);
		{
			WorkerPool &pool = workers(c);
			auto par = [&](size_t nt, const std::function<void(size_t)> &f) { pool.run(nt, f); };
			suffix_array_u32_par(S.data(), ns, SA.data(), par, std::max<size_t>(1, pool.size()));
		}
		HIPOK(c, hipMemcpyAsync(c->d_SA.p, SA.data(), SA.size() * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, hipMemsetAsync(c->a_misc.p, 0, 64, st));
		HIPOK(c, hipMemsetAsync(c->d_LCP.p, 0, ((size_t)ns + 1 + 4) * 4, st));
		launch_lcp(c->d_S.p, c->d_SA.p, ns, 0xffffu, c->d_LCP.p, c->a_misc.p, st);
		HIPOK(c, hipMemcpyAsync(lcp_words, c->a_misc.p, 8, hipMemcpyDeviceToHost, st));
		HIPOK(c, hipStreamSynchronize(st));
		if (lcp_words[1]) return c->fail("phylo_set_reference: the host's suffix array failed the device's check");
	}
	const uint32_t capped = lcp_words[0];
	if (capped) {
		std::vector<uint32_t> LCP((size_t)ns + 1 + 4, 0);
		if (host_sa() || host_S()) return 1;
		lcp_kasai(S.data(), ns, SA.data(), LCP.data());
		HIPOK(c, hipMemcpyAsync(c->d_LCP.p, LCP.data(), LCP.size() * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, hipStreamSynchronize(st));
		c->stats["ref:lcp_from_host"] = 1;
	}
	double t2a = now_ms();
	launch_kmer_table(c->d_S.p, ns, k, c->d_T.p, c->d_T.p + codes + 1 + 4, st);
	launch_sax(c->d_S.p, c->d_SA.p, c->d_LCP.p, ns, c->d_SAX.p, st);
	{
		hipLaunchKernelGGL(build_slots_kernel, dim3((uint32_t)((codes + 255) / 256)), dim3(256), 0, st, c->d_T.p,
						   c->d_SAX.p, ns, k, codes, c->slot_at);
	}
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipStreamSynchronize(st));
	double t2b = now_ms();
	{ // the lean chain's view of S: 2-bit codes and the sorted non-ACGT positions, then n
		const size_t swords = ((size_t)ns + 64) / 16; // S's buffer is ns + 64 bytes
		HIPOK(c, c->d_S2.ensure(swords + 64));
		HIPOK(c, hipMemsetAsync(c->d_S2.p, 0, (swords + 64) * 4, st));
		launch_pack2(c->d_S.p, (uint64_t)swords * 16, c->d_S2.p, st);
		HIPOK(c, c->d_badoff.ensure(2));
		HIPOK(c, c->d_badscr.ensure(8));
		const uint64_t zero = 0;
		HIPOK(c, hipMemcpyAsync(c->d_badoff.p, &zero, 8, hipMemcpyHostToDevice, st));
		DevBuf<uint32_t> lenbuf;
		HIPOK(c, lenbuf.ensure(2));
		HIPOK(c, hipMemcpyAsync(lenbuf.p, &ns, 4, hipMemcpyHostToDevice, st));
		std::vector<uint32_t> off;
		int rc = bad_lists(c, c->d_S.p, c->d_badoff.p, lenbuf.p, std::vector<uint64_t>{ns}, c->d_SBAD, off, 1);
		lenbuf.release();
		if (rc) return 1;
		HIPOK(c, hipMemcpy(c->d_SBAD.p + off[1], &ns, 4, hipMemcpyHostToDevice)); // the end of S closes the list
		c->nsb = off[1] + 1;
		HIPOK(c, hipMemcpy(&c->sb_first, c->d_SBAD.p, 4, hipMemcpyDeviceToHost));
		// The reference's 6-mer cache bug (esa.cxx:174-199) needs two contig joins behind the same few
		// nucleotides: looked for only when S holds a '!' at all (nsb counts '#', the end and the '!'s).
		c->cache_quirk = false;
		c->stats["ms:ref_packed_view"] += now_ms() - t2b;
		// ... and a nucleotide string (up to 5 letters) all of whose occurrences go on with the same byte: when every
		// string that occurs is followed by two different bytes or more, the cache's walk never fast-forwards
		// (esa.cxx:150-155 is the only branch taken) and the exact check over the suffix array is not needed
		bool suspect = false;
		for (uint32_t m : next_masks) suspect = suspect || (m && !(m & (m - 1)));
		c->stats["ref:cache_quirk_exact_check"] = (c->nsb > 2 && suspect) ? 1 : 0;
		c->nquirk = 0;
		if (c->nsb > 2 && suspect) {
			if (host_sa() || host_S()) return 1;
			const std::vector<CacheQuirk> qs = esa_cache_quirks(S.data(), ns, SA.data());
			c->cache_quirk = !qs.empty();
			if (!qs.empty()) {
				std::vector<U4> tab;
				for (const CacheQuirk &e : qs) tab.push_back(U4{e.prefix, e.k | (e.depth << 8), e.lo, e.hi});
				HIPOK(c, c->d_quirk.ensure(tab.size()));
				HIPOK(c, hipMemcpy(c->d_quirk.p, tab.data(), tab.size() * sizeof(U4), hipMemcpyHostToDevice));
				c->nquirk = (uint32_t)tab.size();
			}
		}
		c->stats["ref:cache_quirk_entries"] = c->nquirk;
		c->stats["ref:cache_quirk"] = c->cache_quirk ? 1 : 0;
	}
	double t3 = now_ms();
	c->ref_idx = ref_idx;
	c->L = (uint32_t)L;
	c->ns = ns;
	c->k = k;
	c->threshold = (uint32_t)threshold;
	c->have_ref = true;
	c->plan_valid = false;
	c->homs_staged = false;
	c->att_homs = nullptr; // lists that only lived in an attached buffer are gone
	c->att_rng_on_device = false;
	c->host_stale.clear();
	c->stats["ms:ref_fetch"] += t1 - t0;
	c->stats["ms:ref_suffix_array"] += t2 - t1;
	c->stats["ms:ref_lcp_table"] += t3 - t2;
	c->stats["ms:ref_lcp"] += t2a - t2;
	c->stats["ms:ref_slots"] += t2b - t2a;
	c->stats["ms:ref_total"] += now_ms() - t0;
	c->stats["ref:sa_on_device"] = sa_on_device ? 1 : 0;
	c->stats["ref:sa_rounds"] = sa_rounds;
	c->stats["ref:k"] = k;
	c->stats["ref:threshold"] = (double)threshold;
	c->stats["ref:size"] = ns;
	return 0;
}

size_t phylo_threshold(const phylo_ctx *c) { return c ? c->threshold : 0; }

int phylo_reference_suffix_array(phylo_ctx *c, int64_t *sa)
{
	if (!c) return 1;
	if (!c->have_ref) return c->fail("phylo_reference_suffix_array: no reference set");
	if (!sa) return c->fail("null buffer");
	HIPOK(c, hipSetDevice(c->device));
	std::vector<uint32_t> tmp(c->ns);
	HIPOK(c, hipMemcpy(tmp.data(), c->d_SA.p, (size_t)c->ns * 4, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < c->ns; i++) sa[i] = tmp[i];
	return 0;
}

int phylo_reference_cache_quirk(const phylo_ctx *c) { return c && c->have_ref && c->cache_quirk ? 1 : 0; }

} // extern "C"

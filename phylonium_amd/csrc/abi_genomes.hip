// abi_genomes.hip — `queries` of process() (/root/reference/src/process.cxx:408-409) on the device: the genome
// arena and its layout, uploads as bytes or as 2-bit codes (phylo_set_genomes*), the packed companions the chain
// kernels read (Q2, QBAD; lean_core.h) and the byte arena the other kernels read.
#include "abi_ctx.hpp"

using namespace phy;
using namespace phyabi;

extern "C" {

// Sorted non-ACGT positions of `nseq` sequences (base + off[j], len[j] bytes; device arrays) into
// `out` (grown as needed), list j at [list_off[j], list_off[j+1]); `extra` more slots are left
// after the last list.  Two passes of bad_positions_kernel over segments of 1 MiB.
int phyabi::bad_lists(phylo_ctx *c, const uint8_t *base, const uint64_t *d_off, const uint32_t *d_len,
						const std::vector<uint64_t> &len, DevBuf<uint32_t> &out, std::vector<uint32_t> &list_off, size_t extra)
{
	const size_t nseq = len.size();
	const uint64_t SEG = bad_segment_bytes();
	std::vector<uint32_t> seg_seq, seg_first(nseq + 1);
	for (size_t j = 0; j < nseq; j++) {
		seg_first[j] = (uint32_t)seg_seq.size();
		const uint64_t ns = std::max<uint64_t>(1, (len[j] + SEG - 1) / SEG);
		for (uint64_t t = 0; t < ns; t++) seg_seq.push_back((uint32_t)j);
	}
	seg_first[nseq] = (uint32_t)seg_seq.size();
	const size_t nseg = seg_seq.size();
	list_off.assign(nseq + 1, 0);
	if (!nseg) return c->d_badscr.ensure(4) == hipSuccess && out.ensure(extra + 1) == hipSuccess ? 0 : c->fail("out of device memory");
	// scratch: seg_seq | seg_first | seg_cnt | seg_off
	HIPOK(c, c->d_badscr.ensure(3 * nseg + nseq + 1));
	uint32_t *d_seq = c->d_badscr.p, *d_first = d_seq + nseg, *d_cnt = d_first + nseq + 1, *d_soff = d_cnt + nseg;
	hipStream_t st = c->stream;
	HIPOK(c, hipMemcpyAsync(d_seq, seg_seq.data(), nseg * 4, hipMemcpyHostToDevice, st));
	HIPOK(c, hipMemcpyAsync(d_first, seg_first.data(), (nseq + 1) * 4, hipMemcpyHostToDevice, st));
	launch_bad_positions(base, d_off, d_len, d_seq, d_first, (uint32_t)nseg, d_cnt, nullptr, nullptr, st);
	std::vector<uint32_t> cnt(nseg), soff(nseg + 1, 0);
	HIPOK(c, hipMemcpyAsync(cnt.data(), d_cnt, nseg * 4, hipMemcpyDeviceToHost, st));
	HIPOK(c, hipStreamSynchronize(st));
	uint64_t tot = 0;
	for (size_t g = 0; g < nseg; g++) {
		soff[g] = (uint32_t)tot;
		tot += cnt[g];
	}
	if (tot + extra >= 0xffffffffull) return c->fail("more than 2^32 non-ACGT positions");
	soff[nseg] = (uint32_t)tot;
	for (size_t j = 0; j <= nseq; j++) list_off[j] = soff[seg_first[j]];
	HIPOK(c, out.ensure(tot + extra + 1));
	if (tot) {
		HIPOK(c, hipMemcpyAsync(d_soff, soff.data(), nseg * 4, hipMemcpyHostToDevice, st));
		launch_bad_positions(base, d_off, d_len, d_seq, d_first, (uint32_t)nseg, d_cnt, d_soff, out.p, st);
	}
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipStreamSynchronize(st));
	return 0;
}

// Q2 + QBAD of the installed genomes (d_goff / d_glen are in place)
static int pack_genomes(phylo_ctx *c)
{
	const size_t n = c->n;
	double t0 = now_ms();
	uint64_t extent = 64;
	for (size_t j = 0; j < n; j++) extent = std::max<uint64_t>(extent, c->goff[j] + (c->glen[j] + 63) / 64 * 64 + 64);
	if (extent / 16 + 64 >= 0xffffffffull) return c->fail("genome buffer too large for 32-bit word offsets");
	const size_t words = (size_t)(extent / 16);
	HIPOK(c, c->d_Q2.ensure(words + 64));
	HIPOK(c, hipMemsetAsync(c->d_Q2.p, 0, (words + 64) * 4, c->stream));
	if (n) launch_pack2(c->d_genomes, (uint64_t)words * 16, c->d_Q2.p, c->stream);
	std::vector<uint32_t> off;
	if (bad_lists(c, c->d_genomes, c->d_goff.p, c->d_glen.p, c->glen, c->d_QBAD, off, 1)) return 1;
	HIPOK(c, c->d_qbad_off.ensure(n + 2));
	HIPOK(c, hipMemcpy(c->d_qbad_off.p, off.data(), (n + 1) * 4, hipMemcpyHostToDevice));
	c->stats["ms:pack_genomes"] += now_ms() - t0;
	c->stats["count:genome_non_acgt"] = off[n];
	c->pileup_five = off[n] > 0;
	c->bang_cap = off[n];
	return 0;
}

static int install_layout(phylo_ctx *c, bool pack = true)
{
	size_t n = c->n;
	std::vector<uint32_t> l32(n);
	for (size_t j = 0; j < n; j++) {
		if (c->glen[j] >= 0xfff00000ull) return c->fail("genome %zu is too long (%llu >= 2^32-2^20)", j, (unsigned long long)c->glen[j]);
		l32[j] = (uint32_t)c->glen[j];
	}
	HIPOK(c, c->d_goff.ensure(n + 1));
	HIPOK(c, c->d_glen.ensure(n + 1));
	HIPOK(c, hipMemcpyAsync(c->d_goff.p, c->goff.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipMemcpyAsync(c->d_glen.p, l32.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	c->homs.assign(n, {});
	c->have_ref = false;
	c->plan_valid = false;
	c->homs_staged = false;
	c->att_homs = nullptr; // lists that only lived in an attached buffer are gone
	c->att_rng_on_device = false;
	c->host_stale.clear();
	// Phase B goes ahead on a guess of whether a projected position will hold '!' (compare_pileup).  Genomes without any
	// separator cannot project one; genomes in several contigs nearly always do (a homology that ends at a contig join
	// carries it).  The guess is re-seeded from the new genomes' '!' count where that count becomes known, so the first
	// call after an install starts with the right kernels instead of repeating the work.
	c->pileup_five = false;
	return pack ? pack_genomes(c) : 0;
}

// goff / glen / arena size for n genomes of the given lengths: 64 bytes in front, every genome padded to a
// multiple of 64 and followed by 64 zero bytes, 256 behind the last (kernels prefetch whole 128-byte windows)
static uint64_t layout_genomes(phylo_ctx *c, size_t n, const size_t *len)
{
	c->n = n;
	c->goff.assign(n, 0);
	c->glen.assign(n, 0);
	uint64_t tot = 64;
	for (size_t j = 0; j < n; j++) {
		c->goff[j] = tot;
		c->glen[j] = len[j];
		tot += ((len[j] + 63) / 64) * 64 + 64;
	}
	return tot + 256;
}

int phylo_set_genomes(phylo_ctx *c, size_t n, const char *const *seq, const size_t *len)
{
	if (!c) return 1;
	drop_pending_anchor(c);
	if (n && (!seq || !len)) return c->fail("null genome arrays");
	HIPOK(c, hipSetDevice(c->device));
	const uint64_t tot = layout_genomes(c, n, len);
	double t0 = now_ms();
	HIPOK(c, c->genomes_store.ensure(tot));
	HIPOK(c, hipMemsetAsync(c->genomes_store.p, 0, tot, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	double t1 = now_ms();
	// one thread, one stream: copies from pageable memory issued from several threads at once run at a
	// fraction of this rate (measured: 38 GB/s against 6-20 GB/s from 4-16 threads)
	for (size_t j = 0; j < n; j++)
		if (len[j])
			HIPOK(c, hipMemcpyAsync(c->genomes_store.p + c->goff[j], seq[j], len[j], hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	c->d_genomes = c->genomes_store.p;
	c->own_genomes = true;
	double t2 = now_ms();
	int rc = install_layout(c);
	c->stats["ms:genomes_alloc"] += t1 - t0;
	c->stats["ms:genomes_copy"] += t2 - t1;
	c->stats["ms:genomes_install"] += now_ms() - t2;
	return rc;
}

// Genomes that arrive as 2-bit codes + separator positions (what phylo_host_read_fasta_packed makes): a quarter
// of the bytes cross PCIe, Q2 is copied straight into place and the byte arena is written by the device.
int phylo_set_genomes_packed(phylo_ctx *c, size_t n, const uint32_t *const *q2, const size_t *len, const uint32_t *const *bad,
							 const size_t *nbad)
{
	if (!c) return 1;
	drop_pending_anchor(c);
	if (n && (!q2 || !len || !bad || !nbad)) return c->fail("null genome arrays");
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	std::vector<uint32_t> boff(n + 1, 0), blist;
	{
		uint64_t tb = 0;
		for (size_t j = 0; j < n; j++) {
			if (len[j] && !q2[j]) return c->fail("genome %zu: null code array", j);
			if (nbad[j] && !bad[j]) return c->fail("genome %zu: null position list", j);
			for (size_t k = 0; k < nbad[j]; k++)
				if (bad[j][k] >= len[j] || (k && bad[j][k] <= bad[j][k - 1]))
					return c->fail("genome %zu: separator positions must ascend and lie inside the genome", j);
			tb += nbad[j];
			if (tb + 1 >= 0xffffffffull) return c->fail("more than 2^32 non-ACGT positions");
			boff[j + 1] = (uint32_t)tb;
		}
		blist.reserve(tb);
		for (size_t j = 0; j < n; j++) blist.insert(blist.end(), bad[j], bad[j] + nbad[j]);
	}
	const uint64_t tot = layout_genomes(c, n, len);
	if (tot / 16 + 64 >= 0xffffffffull) return c->fail("genome buffer too large for 32-bit word offsets");
	for (size_t j = 0; j < n; j++)
		if (len[j] >= 0xfff00000ull) return c->fail("genome %zu is too long (%llu >= 2^32-2^20)", j, (unsigned long long)len[j]);
	const size_t words = (size_t)(tot / 16);
	HIPOK(c, c->genomes_store.ensure(tot));
	HIPOK(c, c->d_Q2.ensure(words + 64));
	HIPOK(c, hipMemsetAsync(c->d_Q2.p, 0, (words + 64) * 4, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	double t1 = now_ms();
	// (the codes come out of pageable memory — the reader's arena — and the runtime stages them through page-locked buffers
	// of its own at ~12 GB/s; several host threads with a stream each staging side by side were measured and are slower:
	// C3 0.075 s against 0.026, C4 0.08-0.14 against 0.09-0.10)
	for (size_t j = 0; j < n; j++)
		if (len[j])
			HIPOK(c, hipMemcpyAsync(c->d_Q2.p + c->goff[j] / 16, q2[j], (len[j] + 15) / 16 * 4, hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	double t2 = now_ms();
	c->d_genomes = c->genomes_store.p;
	c->own_genomes = true;
	if (install_layout(c, false)) return 1; // d_goff / d_glen in place
	HIPOK(c, c->d_QBAD.ensure(blist.size() + 2));
	HIPOK(c, c->d_qbad_off.ensure(n + 2));
	if (!blist.empty()) HIPOK(c, hipMemcpyAsync(c->d_QBAD.p, blist.data(), blist.size() * 4, hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipMemcpyAsync(c->d_qbad_off.p, boff.data(), (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
	launch_unpack2(c->d_Q2.p, c->d_goff.p, c->d_glen.p, (uint32_t)n, tot, c->genomes_store.p, c->d_QBAD.p, c->d_qbad_off.p,
				   (uint32_t)blist.size(), c->stream);
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipStreamSynchronize(c->stream));
	c->stats["ms:genomes_alloc"] += t1 - t0;
	c->stats["ms:genomes_copy"] += t2 - t1;
	c->stats["ms:genomes_install"] += now_ms() - t2;
	c->stats["count:genome_non_acgt"] = (double)blist.size();
	c->pileup_five = !blist.empty();
	c->bang_cap = (uint32_t)blist.size();
	return 0;
}

// The packed genomes already in device memory, laid out as the arena's Q2: word w of dev_q2 holds the codes of
// arena bytes [16w, 16w + 16), genome j at byte offset offsets[j] (the rules of phylo_set_genomes_device), codes
// outside the genomes and at the separator positions 0.  A rank of a multi-GPU run gathers exactly this.
int phylo_set_genomes_packed_device(phylo_ctx *c, size_t n, const void *dev_q2, const uint64_t *offsets, const uint64_t *lens,
									const uint32_t *const *bad, const size_t *nbad)
{
	if (!c) return 1;
	drop_pending_anchor(c);
	if (n && (!dev_q2 || !offsets || !lens || !bad || !nbad)) return c->fail("null genome arrays");
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	std::vector<uint32_t> boff(n + 1, 0), blist;
	uint64_t tot = 64;
	for (size_t j = 0; j < n; j++) {
		if (offsets[j] % 64 || offsets[j] < 64)
			return c->fail("genome %zu: device offset must be a multiple of 64 and >= 64", j);
		if (j && offsets[j] < offsets[j - 1] + (lens[j - 1] + 63) / 64 * 64 + 64)
			return c->fail("genome %zu: device offsets must ascend, each genome followed by at least 64 bytes of padding", j);
		if (lens[j] >= 0xfff00000ull) return c->fail("genome %zu is too long (%llu >= 2^32-2^20)", j, (unsigned long long)lens[j]);
		if (nbad[j] && !bad[j]) return c->fail("genome %zu: null position list", j);
		for (size_t k = 0; k < nbad[j]; k++)
			if (bad[j][k] >= lens[j] || (k && bad[j][k] <= bad[j][k - 1]))
				return c->fail("genome %zu: separator positions must ascend and lie inside the genome", j);
		if (blist.size() + nbad[j] + 1 >= 0xffffffffull) return c->fail("more than 2^32 non-ACGT positions");
		blist.insert(blist.end(), bad[j], bad[j] + nbad[j]);
		boff[j + 1] = (uint32_t)blist.size();
		tot = offsets[j] + (lens[j] + 63) / 64 * 64 + 64;
	}
	// the caller's buffer reaches 16 bytes' worth of words past the last genome's padding (the header's promise): that
	// much is copied, the rest of this context's Q2 — the 256 bytes the kernels may prefetch behind it — is cleared here
	const size_t src_words = (size_t)(tot / 16) + 1;
	tot += 256;
	if (tot / 16 + 64 >= 0xffffffffull) return c->fail("genome buffer too large for 32-bit word offsets");
	const size_t words = (size_t)(tot / 16);
	c->n = n;
	c->goff.assign(offsets, offsets + n);
	c->glen.assign(lens, lens + n);
	HIPOK(c, c->genomes_store.ensure(tot));
	HIPOK(c, c->d_Q2.ensure(words + 64));
	HIPOK(c, hipMemsetAsync(c->d_Q2.p + src_words, 0, (words + 64 - src_words) * 4, c->stream));
	HIPOK(c, hipMemcpyAsync(c->d_Q2.p, dev_q2, src_words * 4, hipMemcpyDeviceToDevice, c->stream));
	c->d_genomes = c->genomes_store.p;
	c->own_genomes = true;
	if (install_layout(c, false)) return 1;
	HIPOK(c, c->d_QBAD.ensure(blist.size() + 2));
	HIPOK(c, c->d_qbad_off.ensure(n + 2));
	if (!blist.empty()) HIPOK(c, hipMemcpyAsync(c->d_QBAD.p, blist.data(), blist.size() * 4, hipMemcpyHostToDevice, c->stream));
	HIPOK(c, hipMemcpyAsync(c->d_qbad_off.p, boff.data(), (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
	launch_unpack2(c->d_Q2.p, c->d_goff.p, c->d_glen.p, (uint32_t)n, tot, c->genomes_store.p, c->d_QBAD.p, c->d_qbad_off.p,
				   (uint32_t)blist.size(), c->stream);
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipStreamSynchronize(c->stream));
	c->stats["ms:genomes_install"] += now_ms() - t0;
	c->stats["count:genome_non_acgt"] = (double)blist.size();
	c->pileup_five = !blist.empty();
	c->bang_cap = (uint32_t)blist.size();
	return 0;
}

int phylo_get_genome(phylo_ctx *c, size_t i, char *buf)
{
	if (!c) return 1;
	if (i >= c->n) return c->fail("genome index %zu out of range (n=%zu)", i, c->n);
	if (!buf && c->glen[i]) return c->fail("null buffer");
	HIPOK(c, hipSetDevice(c->device));
	if (c->glen[i]) HIPOK(c, hipMemcpy(buf, c->d_genomes + c->goff[i], c->glen[i], hipMemcpyDeviceToHost));
	return 0;
}

int phylo_set_genomes_device(phylo_ctx *c, size_t n, const void *dev_base, const uint64_t *offsets,
							 const uint64_t *lens)
{
	if (!c) return 1;
	drop_pending_anchor(c);
	if (n && (!dev_base || !offsets || !lens)) return c->fail("null genome arrays");
	HIPOK(c, hipSetDevice(c->device));
	for (size_t j = 0; j < n; j++) {
		if (offsets[j] % 64 || offsets[j] < 64)
			return c->fail("genome %zu: device offset must be a multiple of 64 and >= 64 (kernels read up to 32 bytes before a genome)", j);
		// ascending and apart: phase A clears and addresses its visited bitmap by buffer offset, genome after genome
		if (j && offsets[j] < offsets[j - 1] + (lens[j - 1] + 63) / 64 * 64 + 64)
			return c->fail("genome %zu: device offsets must ascend, each genome followed by at least 64 bytes of zero padding "
						   "(rounded up to a multiple of 64) before the next one starts", j);
	}
	c->n = n;
	c->goff.assign(offsets, offsets + n);
	c->glen.assign(lens, lens + n);
	c->d_genomes = (uint8_t *)dev_base;
	c->own_genomes = false;
	return install_layout(c);
}

} // extern "C"

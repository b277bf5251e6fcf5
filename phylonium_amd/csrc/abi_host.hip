// abi_host.hip — seam B0 (seqcmp / revseqcmp under the reference's signatures, /root/reference/libs/seqcmp.h:14,
// libs/revseqcmp.h:25) and the host-side helpers of the C ABI: FASTA ingest, host suffix array, threshold math,
// distances and the PHYLIP writer (src/io.cxx:141-233, src/evo_model.cxx:100-131).
#include "abi_ctx.hpp"

using namespace phy;
using namespace phyabi;

extern "C" {

// ───────────────────────── B0 ─────────────────────────

int phylo_seqcmp_batch(phylo_ctx *c, size_t n, const uint32_t *ga, const uint64_t *offa, const uint32_t *gb,
					   const uint64_t *offb, const uint64_t *len, const uint8_t *rev, uint64_t *out)
{
	if (!c) return 1;
	if (n == 0) return 0;
	if (!ga || !offa || !gb || !offb || !len || !out) return c->fail("null segment arrays");
	HIPOK(c, hipSetDevice(c->device));
	std::vector<Segment> segs(n);
	for (size_t s = 0; s < n; s++) {
		if (ga[s] >= c->n || gb[s] >= c->n) return c->fail("segment %zu: genome index out of range", s);
		if (offa[s] + len[s] > c->glen[ga[s]] || offb[s] + len[s] > c->glen[gb[s]])
			return c->fail("segment %zu reaches beyond its genome", s);
		if (len[s] > 0xffffffffull) return c->fail("segment %zu too long", s);
		segs[s] = Segment{c->goff[ga[s]] + offa[s], c->goff[gb[s]] + offb[s], (uint32_t)len[s], rev && rev[s] ? 1u : 0u};
	}
	return run_segments(c, c->d_genomes, segs.data(), n, out, "seqcmp_batch");
}

// A batch of segments through the byte kernels (seqcmp_kernels.hip): a single segment — one seqcmp() of megabytes — comes
// with the kernel's arguments; any other batch is cut into rounds of 63 chunks on the device, from the prefix sums made here.
int phyabi::run_segments(phylo_ctx *c, const uint8_t *base, const Segment *segs, size_t n, uint64_t *out, const char *span)
{
	if (!n) return 0;
	if (n >= 0xffffffffull) return c->fail("more than 2^32 segments in one batch");
	HIPOK(c, c->s_segs.ensure(n));
	HIPOK(c, c->s_out.ensure(n));
	hipStream_t st = c->stream;
	uint64_t nrounds = 0;
	if (n > 1) {
		HIPOK(c, hipMemcpyAsync(c->s_segs.p, segs, n * sizeof(Segment), hipMemcpyHostToDevice, st));
		// the rounds' prefix sums, and for every pass of four rounds the segment its first round lies in
		const uint32_t rpp = seqcmp_rounds_per_pass();
		uint64_t total = 0;
		for (size_t s = 0; s < n; s++) total += (segs[s].len + SEQCMP_ROUND - 1) / SEQCMP_ROUND;
		if (total >= 0xfffffff0ull) return c->fail("more than 2^32 KiB in one batch of segments");
		const size_t npasses = (size_t)((total + rpp - 1) / rpp);
		HIPOK(c, c->h_piece0.ensure(n + 1 + npasses));
		uint32_t *r0 = c->h_piece0.p, *hint = c->h_piece0.p + n + 1;
		size_t pass = 0;
		for (size_t s = 0; s < n; s++) {
			r0[s] = (uint32_t)nrounds;
			nrounds += (segs[s].len + SEQCMP_ROUND - 1) / SEQCMP_ROUND;
			for (; pass < npasses && pass * rpp < nrounds; pass++) hint[pass] = (uint32_t)s;
		}
		r0[n] = (uint32_t)nrounds;
		HIPOK(c, c->s_piece0.ensure(n + 1 + npasses));
		HIPOK(c, hipMemcpyAsync(c->s_piece0.p, c->h_piece0.p, (n + 1 + npasses) * 4, hipMemcpyHostToDevice, st));
		HIPOK(c, c->s_rounds.ensure(seqcmp_rounds_bytes(nrounds)));
	}
	HIPOK(c, hipMemsetAsync(c->s_out.p, 0, n * 8, st));
	{
		KernelSpan s(c, span);
		launch_seqcmp_batch(base, c->s_segs.p, (uint32_t)n, c->s_piece0.p, c->s_piece0.p + n + 1, (uint32_t)nrounds, c->s_rounds.p, c->s_out.p, c->n_cu, st, n == 1 ? segs : nullptr);
	}
	HIPOK(c, hipGetLastError());
	HIPOK(c, hipMemcpyAsync(out, c->s_out.p, n * 8, hipMemcpyDeviceToHost, st));
	return sync_stream(c);
}

// seqcmp / revseqcmp with the reference's signature and calling convention (libs/seqcmp.h:14-25,
// libs/revseqcmp.h:25-33): pure, borrowing both host buffers, callable from many threads at once
// (evo_model::account* runs inside an OpenMP team, src/evo_model.cxx:53-75).  Every calling thread
// gets its own context — stream, scratch, nothing shared — created on its first call and released
// when the thread ends.  The signature has no error channel and the library never aborts or falls
// back to the CPU: a failed call returns SIZE_MAX, the message is in phylo_last_error(NULL) of that
// thread and is printed to stderr once per thread.
namespace {
struct B0Thread {
	phylo_ctx *ctx = nullptr;
	DevBuf<uint8_t> buf;
	bool complained = false;
	~B0Thread()
	{
		if (ctx) {
			(void)hipSetDevice(ctx->device);
			buf.release();
			phylo_ctx_destroy(ctx);
		}
	}
};
thread_local B0Thread g_b0;

size_t b0_fail(const char *what)
{
	if (what) g_phylo_last_error = what;
	if (!g_b0.complained) {
		fprintf(stderr, "phylonium_amd: seqcmp/revseqcmp: %s\n", g_phylo_last_error.c_str());
		g_b0.complained = true;
	}
	return (size_t)-1;
}

size_t b0_call(const char *a, const char *b, size_t length, int rev)
{
	if (length == 0) return 0;
	if (!a || !b) return b0_fail("null buffer");
	B0Thread &t = g_b0;
	if (!t.ctx && phylo_ctx_create(&t.ctx, 0)) return b0_fail(nullptr);
	phylo_ctx *c = t.ctx;
	if (hipSetDevice(c->device) != hipSuccess) return b0_fail("hipSetDevice failed");
	// both strings into this thread's scratch, 64-byte aligned starts; pieces of < 2^32 bytes
	const uint64_t stride = (length + 63) / 64 * 64 + 64;
	if (t.buf.ensure(2 * stride + 64) != hipSuccess) return b0_fail("out of device memory");
	const uint64_t piece = 1ull << 30;
	std::vector<Segment> segs;
	for (uint64_t o = 0; o < length; o += piece) {
		const uint64_t m = std::min<uint64_t>(piece, length - o);
		segs.push_back(Segment{o, stride + (rev ? length - o - m : o), (uint32_t)m, rev ? 1u : 0u});
	}
	std::vector<uint64_t> out(segs.size());
	hipStream_t st = c->stream;
	if (hipMemcpyAsync(t.buf.p, a, length, hipMemcpyHostToDevice, st) != hipSuccess ||
		hipMemcpyAsync(t.buf.p + stride, b, length, hipMemcpyHostToDevice, st) != hipSuccess)
		return b0_fail("upload failed");
	// (one call = one segment per GiB: its pieces are dealt out over all wavefronts of the launch)
	if (run_segments(c, t.buf.p, segs.data(), segs.size(), out.data(), "seqcmp_b0")) return b0_fail(nullptr);
	uint64_t total = 0;
	for (uint64_t v : out) total += v;
	return (size_t)total;
}
} // namespace

size_t phylo_seqcmp(const char *begin, const char *other, size_t length) { return b0_call(begin, other, length, 0); }
size_t phylo_revseqcmp(const char *begin, const char *other, size_t length) { return b0_call(begin, other, length, 1); }

// ───────────────────────── host-side helpers ─────────────────────────

// `s` is followed by 16 zero bytes
static int host_suffix_array_padded(const uint8_t *s, size_t n, int64_t *sa)
{
	std::vector<uint32_t> tmp(n);
	unsigned h = std::thread::hardware_concurrency();
	ThreadFan fan{std::min<size_t>(h ? h : 1, 16)};
	suffix_array_u32_par(s, (uint32_t)n, tmp.data(), fan, fan.nthreads);
	fan(64, [&](size_t t) {
		for (size_t i = n * t / 64, e = n * (t + 1) / 64; i < e; i++) sa[i] = tmp[i];
	});
	return 0;
}

int phylo_host_suffix_array(const char *s, size_t n, int64_t *sa)
{
	if (!s || !sa || n >= 0x7fffffffull) return 1;
	std::vector<uint8_t> padded(n + 16, 0);
	memcpy(padded.data(), s, n);
	return host_suffix_array_padded(padded.data(), n, sa);
}

int phylo_host_reference_suffix_array(const char *ref, size_t len, int64_t *sa)
{
	if (!ref || !sa || 2 * len + 1 >= 0x7fffffffull) return 1;
	std::vector<uint8_t> S(2 * len + 1 + 64, 0);
	memcpy(S.data(), ref, len);
	S[len] = '#';
	revcomp((const uint8_t *)ref, len, S.data() + len + 1);
	return host_suffix_array_padded(S.data(), 2 * len + 1, sa);
}

size_t phylo_host_min_anchor_length(double p, double gc, size_t l) { return min_anchor_length(p, gc, l); }

int phylo_host_device_count(int *count)
{
	if (!count) return 1;
	*count = 0;
	return hipGetDeviceCount(count) == hipSuccess ? 0 : 1;
}

int phylo_host_read_fasta(size_t n, const char *const *paths, size_t threads, char **out, size_t *len)
{
	if (!paths || !out || !len) return 1;
	std::vector<std::string> files(paths, paths + n);
	std::vector<phyfasta::ReadResult> res(n);
	ThreadFan fan{std::max<size_t>(1, threads)};
	std::vector<char *> bufs(n, nullptr);
	std::vector<size_t> sizes(n, 0);
	std::vector<std::string> errors(n);
	fan(n, [&](size_t i) { // read, filter and hand over in the same task: the copies run on all threads
		phyfasta::ReadResult r = phyfasta::read_genome(files[i]);
		if (!r.error.empty()) {
			errors[i] = r.error;
			return;
		}
		const std::string &g = r.g.nucl;
		bufs[i] = (char *)malloc(g.size() + 1);
		if (!bufs[i]) {
			errors[i] = files[i] + ": out of memory";
			return;
		}
		memcpy(bufs[i], g.data(), g.size());
		bufs[i][g.size()] = 0;
		sizes[i] = g.size();
	});
	for (size_t i = 0; i < n; i++)
		if (!errors[i].empty()) {
			g_phylo_last_error = errors[i];
			for (size_t k = 0; k < n; k++) free(bufs[k]);
			return (int)(i + 1);
		}
	for (size_t i = 0; i < n; i++) {
		out[i] = bufs[i];
		len[i] = sizes[i];
	}
	return 0;
}

int phylo_host_read_fasta_packed(size_t n, const char *const *paths, size_t threads, uint32_t **q2, size_t *len, uint32_t **bad,
								 size_t *nbad, void **arena)
{
	if (!paths || !q2 || !len || !bad || !nbad || !arena) return 1;
	std::vector<std::string> files(paths, paths + n);
	std::string error;
	uint32_t *words = nullptr;
	size_t bad_idx = 0; // the first file, in the order given, that failed
	std::vector<phyfasta::PackedGenome> g = phyfasta::read_genomes_packed(files, std::max<size_t>(1, threads), &error, &words, &bad_idx);
	// one more allocation holds the separator lists and whatever did not fit its place in the first (a file that was
	// not a regular file): both leave with the arena
	size_t extra = 0;
	for (auto &x : g) extra += x.bad.size() + 1 + (x.own ? (size_t)((x.len + 15) / 16) + 16 : 0);
	uint32_t *second = error.empty() ? (uint32_t *)malloc((extra + 16) * sizeof(uint32_t)) : nullptr;
	if (error.empty() && (!words || !second)) error = "out of memory";
	if (!error.empty()) {
		g_phylo_last_error = error;
		for (size_t i = 0; i < n; i++)
			if (g[i].own) free(g[i].q2);
		free(words);
		free(second);
		return (int)(bad_idx + 1);
	}
	// header of the handle: the two allocations
	void **handle = (void **)malloc(2 * sizeof(void *));
	if (!handle) {
		for (auto &x : g)
			if (x.own) free(x.q2);
		free(words);
		free(second);
		g_phylo_last_error = "out of memory";
		return 1;
	}
	handle[0] = words;
	handle[1] = second;
	size_t w = 0;
	for (size_t i = 0; i < n; i++) {
		if (g[i].own) {
			const size_t nw = (size_t)((g[i].len + 15) / 16);
			w = (w + 15) / 16 * 16;
			if (nw) memcpy(second + w, g[i].q2, nw * sizeof(uint32_t));
			free(g[i].q2);
			q2[i] = second + w;
			w += nw;
		} else {
			q2[i] = g[i].q2;
		}
		len[i] = (size_t)g[i].len;
	}
	for (size_t i = 0; i < n; i++) {
		bad[i] = second + w;
		nbad[i] = g[i].bad.size();
		if (nbad[i]) memcpy(second + w, g[i].bad.data(), nbad[i] * sizeof(uint32_t));
		w += nbad[i];
	}
	*arena = handle;
	return 0;
}

void phylo_host_free_packed(void *arena)
{
	if (!arena) return;
	void **handle = (void **)arena;
	free(handle[0]);
	free(handle[1]);
	free(handle);
}

void phylo_host_free(void *p) { free(p); }

size_t phylo_host_median_length_index(size_t n, const size_t *len)
{
	if (!n || !len) return 0;
	std::vector<size_t> idx(n);
	for (size_t i = 0; i < n; i++) idx[i] = i;
	std::nth_element(idx.begin(), idx.begin() + n / 2, idx.end(), [&](size_t a, size_t b) { return len[a] < len[b]; });
	return idx[n / 2];
}

size_t phylo_host_sort_filter(phylo_homology *h, size_t n, int do_sort)
{
	std::vector<phylo_homology> v(h, h + n);
	if (do_sort) {
		// the path phase A takes: packed keys, structs only when two entries share a start
		SortFilterScratch scratch;
		std::vector<uint32_t> kept;
		auto get = [&](size_t i, uint64_t *start, uint64_t *len) {
			*start = v[i].index_reference_projected;
			*len = v[i].length;
		};
		if (sort_filter_order(n, get, scratch, kept)) {
			for (size_t t = 0; t < kept.size(); t++) h[t] = v[kept[t]];
			return kept.size();
		}
		sort_and_filter(v);
	} else {
		filter_overlaps_max(v);
	}
	std::copy(v.begin(), v.end(), h);
	return v.size();
}

double phylo_estimate(int kind, uint64_t subst, uint64_t homologs, int zero_on_error)
{
	switch (kind) {
		case 0: return estimate_jc(subst, homologs, zero_on_error != 0);
		case 1: return estimate_raw(subst, homologs, zero_on_error != 0);
		default: return estimate_ani(subst, homologs, zero_on_error != 0);
	}
}

// src/io.cxx:141-163
size_t phylo_format_phylip(size_t n, const char *const *names, const uint64_t *subst, const uint64_t *homologs,
						   int kind, char *out, size_t cap)
{
	// just_print, io.cxx:141-163: precision 4 with std::scientific ("%.4e"), or the default float format for ANI
	// (std::dec does not touch it: "%.4g").  Row blocks are formatted on the host threads and joined in order.
	unsigned hw = std::thread::hardware_concurrency();
	const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 1, 16), n / 16 + 1));
	std::vector<std::string> part(nt);
	ThreadFan fan{nt};
	fan(nt, [&](size_t t) {
		std::string &o = part[t];
		const size_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
		o.reserve((i1 - i0) * (n * 12 + 32));
		char buf[64];
		for (size_t i = i0; i < i1; i++) {
			o += names[i];
			for (size_t j = 0; j < n; j++) {
				const double d = (i == j) ? 0.0 : phylo_estimate(kind, subst[i * n + j], homologs[i * n + j], 0);
				buf[0] = buf[1] = ' ';
				o.append(buf, 2 + (kind == 2 ? (size_t)snprintf(buf + 2, sizeof buf - 2, "%.4g", d) : phyfmt::e4(buf + 2, d)));
			}
			o += '\n';
		}
	});
	std::string head = std::to_string(n) + "\n";
	size_t need = head.size() + 1;
	for (auto &p : part) need += p.size();
	if (out && cap >= need) {
		char *w = out;
		memcpy(w, head.data(), head.size());
		w += head.size();
		for (auto &p : part) {
			memcpy(w, p.data(), p.size());
			w += p.size();
		}
		*w = 0;
	}
	return need;
}

} // extern "C"

// abi_context.hip — the C ABI's context: create / destroy, the caller's stream, options, statistics
// (include/phylonium_amd.h; struct phylo_ctx is abi_ctx.hpp).
#include "abi_ctx.hpp"

using namespace phy;
using namespace phyabi;

thread_local std::string g_phylo_last_error;

extern "C" {

const char *phylo_version(void) { return "phylonium_amd 0.1 (gfx950)"; }

const char *phylo_last_error(const phylo_ctx *ctx) { return ctx ? ctx->err.c_str() : g_phylo_last_error.c_str(); }

int phylo_ctx_create(phylo_ctx **out, int device)
{
	if (!out) return 1;
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		g_phylo_last_error = std::string("no usable HIP device: ") + hipGetErrorString(e);
		return 2;
	}
	if (device < 0 || device >= count) {
		g_phylo_last_error = "device ordinal out of range";
		return 3;
	}
	phylo_ctx *c = new phylo_ctx();
	c->device = device;
	if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&c->stream)) != hipSuccess ||
		(e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) != hipSuccess) {
		g_phylo_last_error = std::string("cannot initialise device: ") + hipGetErrorString(e);
		delete c;
		return 4;
	}
	c->own_stream = c->stream;
#ifdef PHY_DEV_HOOKS
	if (const char *e = getenv("PHYLONIUM_AMD_ZERO_COPY")) c->opt_result_zero_copy = atoi(e) != 0; // experiments
#endif
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
	project_resident_blocks(c->proj_resident);
	*out = c;
	return 0;
}

int phylo_ctx_set_stream(phylo_ctx *c, void *stream)
{
	if (!c) return 1;
	HIPOK(c, hipSetDevice(c->device));
	HIPOK(c, hipStreamSynchronize(c->stream));
	harvest_spans(c);
	c->stream = stream ? (hipStream_t)stream : c->own_stream;
	return 0;
}

int phylo_ctx_device(const phylo_ctx *c) { return c ? c->device : -1; }

void phylo_ctx_destroy(phylo_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	c->pool.reset();
	phylo_result_close(c);
	c->h_cnt.release();
	c->h_rng.release();
	c->h_raw.release();
	c->h_devhom.release();
	c->h_mat.release();
	for (auto &r : c->host_regs)
		if (r.dev) (void)hipHostUnregister(r.ptr);
	(void)hipGetLastError();
	c->genomes_store.release();
	c->d_goff.release();
	c->d_glen.release();
	c->d_S.release();
	c->d_SAX.release();
	c->d_SLOT.release();
	c->d_SA.release();
	c->d_LCP.release();
	c->d_T.release();
	c->d_Q2.release();
	c->d_QBAD.release();
	c->d_qbad_off.release();
	c->d_S2.release();
	c->d_SBAD.release();
	c->d_badscr.release();
	c->d_badoff.release();
	c->d_quirk.release();
	c->a_flt.release();
	c->a_long.release();
	c->a_qoff.release();
	c->a_qlen.release();
	c->a_qchunk0.release();
	c->a_qanc0.release();
	c->a_items.release();
	c->a_work.release();
	c->a_qdesc.release();
	c->a_chunk_query.release();
	c->a_spec_cnt.release();
	c->a_visited.release();
	c->a_misc.release();
	c->a_spec_anchors.release();
	c->a_spec_exit.release();
	c->a_bridge.release();
	c->a_bridge_start.release();
	c->a_pool.release();
	c->a_raw.release();
	c->a_raw_compact.release();
	c->a_out_base.release();
	c->a_cmp_base.release();
	c->a_out_cap.release();
	c->a_out_cnt.release();
	c->b_planes.release();
	c->b_hom_rng.release();
	c->b_tiles.release();
	c->tiles_key = 0;
	c->tiles_at = nullptr;
	c->b_flag.release();
	c->b_first.release();
	c->b_homs.release();
	c->b_subst.release();
	c->b_sym32.release();
	c->b_bang.release();
	c->b_clk.release();
	c->s_segs.release();
	c->s_out.release();
	c->s_piece0.release();
	c->h_piece0.release();
	for (TimedSpan &s : c->spans) {
		(void)hipEventDestroy(s.a);
		(void)hipEventDestroy(s.b);
	}
	for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
	for (hipEvent_t e : c->copy_events) (void)hipEventDestroy(e);
	if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
	(void)hipStreamDestroy(c->own_stream);
	delete c;
}

int phylo_set_option(phylo_ctx *c, const char *key, long value)
{
	if (!c || !key) return 1;
	std::string k = key;
	if (k == "chunk") {
		if (value != 0 && (value < 64 || value % 64 || value > 65536)) return c->fail("chunk must be 0 or a multiple of 64 in 64..65536");
		c->opt_chunk = (uint32_t)value;
		c->plan_valid = false;
		c->homs_staged = false;
	} else if (k == "kmer") {
		if (value < 0 || value > 14) return c->fail("kmer must be in 0..14");
		c->opt_kmer = (uint32_t)value;
		c->have_ref = false;
	} else if (k == "filter_kernel") {
		if (value != 0 && value != 1) return c->fail("filter_kernel must be 0 (stretch-wise) or 1 (general)");
		c->opt_filter_kernel = (int)value;
	} else if (k == "sa_builder") {
		if (value != 0 && value != 1) return c->fail("sa_builder must be 1 (device) or 0 (host cores)");
		c->opt_sa_builder = (int)value;
	} else if (k == "pairs_wchunk") {
		if (value < 0 || value > (1 << 20)) return c->fail("pairs_wchunk must be in 0..2^20");
		c->opt_pairs_wchunk = (uint32_t)value;
	} else if (k == "cache_quirk") {
		if (value != 0 && value != 1) return c->fail("cache_quirk must be 1 (as the reference answers) or 0 (true longest matches)");
		c->opt_cache_quirk = (int)value;
		c->plan_valid = false;
		c->homs_staged = false;
	} else if (k == "spec_blocks") {
		if (value < 0 || value > 65536) return c->fail("spec_blocks must be in 0..65536");
		c->opt_spec_blocks = (uint32_t)value;
	} else if (k == "fold_blocks") {
		if (value < 0 || value > 64) return c->fail("fold_blocks must be in 0..64");
		c->opt_fold_blocks = (uint32_t)value;
	} else if (k == "pairs_kernel") {
		if (value != 0 && value != 1) return c->fail("pairs_kernel must be 0 (matrix cores unless '!' is projected) or 1 (vector ALUs)");
		c->opt_pairs_kernel = (int)value;
	} else if (k == "result_zero_copy") {
		c->opt_result_zero_copy = value != 0;
		if (!value) { // and let go of the caller's matrices (a caller about to free them says so this way)
			HIPOK(c, hipSetDevice(c->device));
			HIPOK(c, hipStreamSynchronize(c->stream));
			for (auto &r : c->host_regs)
				if (r.dev) (void)hipHostUnregister(r.ptr);
			(void)hipGetLastError();
			c->host_regs.clear();
		}
	} else if (k == "lean_force_slow") {
		c->lean_force_slow = value != 0;
	} else if (k == "profile") {
		if (value < 0 || value > 2) return c->fail("profile must be 0 (off), 1 (every kernel) or 2 (the chain kernel only)");
		c->profile = (int)value;
	} else if (k == "filter") {
		if (value < 0 || value > 2) return c->fail("filter must be 0 (auto), 1 (host) or 2 (device)");
		c->filter_mode = (int)value;
	} else if (k == "compare_backend") {
		if (value != 0 && value != 1) return c->fail("compare_backend must be 0 (pileup) or 1 (segments)");
		c->backend = (int)value;
	} else if (k == "host_threads") {
		c->host_threads = (int)value;
		c->pool.reset();
	} else {
		return c->fail("unknown option '%s'", key);
	}
	return 0;
}

int phylo_get_stat(phylo_ctx *c, const char *key, double *out)
{
	if (!c || !key || !out) return 1;
	if (c->b_clk.p && !strcmp(key, "clock:pairs_mfma_mhz")) { // the clock the chip held under the matrix-core pair kernel (profiled launches since the last reset)
		unsigned long long h[2] = {0, 0};
		HIPOK(c, hipSetDevice(c->device));
		HIPOK(c, hipStreamSynchronize(c->stream));
		HIPOK(c, hipMemcpy(h, c->b_clk.p, 16, hipMemcpyDeviceToHost));
		c->stats["clock:pairs_mfma_mhz"] = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
	}
	auto it = c->stats.find(key);
	if (it == c->stats.end()) return 1;
	*out = it->second;
	return 0;
}

int phylo_reset_stats(phylo_ctx *c)
{
	if (!c) return 1;
	c->stats.clear();
	if (c->b_clk.p) {
		HIPOK(c, hipSetDevice(c->device));
		HIPOK(c, hipMemsetAsync(c->b_clk.p, 0, 16, c->stream));
		c->stats["clock:pairs_mfma_mhz"] = 0;
	}
	return 0;
}

size_t phylo_stat_keys(phylo_ctx *c, char *buf, size_t cap)
{
	if (!c) return 0;
	size_t need = 1;
	for (auto &kv : c->stats) need += kv.first.size() + 1;
	if (buf && cap >= need) {
		char *w = buf;
		for (auto &kv : c->stats) {
			memcpy(w, kv.first.c_str(), kv.first.size() + 1);
			w += kv.first.size() + 1;
		}
		*w = '\0';
	}
	return need;
}

} // extern "C"

// abi_lists.hip — the homology lists between the phases (`homologies[j]` of /root/reference/src/process.cxx:430-469):
// read back on demand, installed by the caller, exported / imported (struct and 16-byte wire form), the exchange
// between ranks without the host in it (fixed-shape device blocks), and complete deletion (process.cxx:725-776).
#include "abi_ctx.hpp"

using namespace phy;
using namespace phyabi;

extern "C" {

static int unpack_lists(phylo_ctx *c, size_t q_begin, size_t q_end, const uint64_t *counts,
						const phylo_packed_homology *buf);

// Host lists of genomes [g0, g1) that only exist as attached device records: fetch them.
int phyabi::ensure_host_lists(phylo_ctx *c, size_t g0, size_t g1)
{
	if (settle_anchor(c)) return 1;
	if (c->host_stale.empty()) return 0;
	if (fetch_att_ranges(c)) return 1;
	for (size_t g = g0; g < g1; g++) {
		if (!c->host_stale[g]) continue;
		const size_t m = c->att_count[g];
		std::vector<phylo_packed_homology> tmp(m);
		if (m) {
			HIPOK(c, hipSetDevice(c->device));
			HIPOK(c, hipMemcpy(tmp.data(), c->att_homs + c->att_begin[g], m * sizeof(DevHom), hipMemcpyDeviceToHost));
		}
		uint64_t one = m;
		c->host_stale[g] = 0;
		if (unpack_lists(c, g, g + 1, &one, tmp.data())) return 1;
	}
	return 0;
}

int phylo_get_homologies(phylo_ctx *c, size_t j, const phylo_homology **h, size_t *n)
{
	if (!c || !h || !n) return 1;
	if (j >= c->n) return c->fail("genome index out of range");
	if (ensure_host_lists(c, j, j + 1)) return 1;
	*h = c->homs[j].data();
	*n = c->homs[j].size();
	return 0;
}

int phylo_set_homologies(phylo_ctx *c, size_t j, const phylo_homology *h, size_t n)
{
	if (!c) return 1;
	if (j >= c->n) return c->fail("genome index out of range");
	if (n && !h) return c->fail("null homology list");
	if (settle_anchor(c)) return 1; // (a phase A queued by phylo_anchor_block_device: its lists come in first, this one on top of them)
	c->homs[j].assign(h, h + n);
	c->homs_staged = false;
	if (!c->host_stale.empty()) {
		if (ensure_host_lists(c, 0, j) || ensure_host_lists(c, j + 1, c->n)) return 1;
		c->host_stale.clear();
	}
	return 0;
}

int phylo_export_homologies(phylo_ctx *c, size_t q_begin, size_t q_end, uint64_t *counts, phylo_homology *buf,
							size_t cap, size_t *total)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !counts || !total) return c->fail("phylo_export_homologies: bad arguments");
	if (ensure_host_lists(c, q_begin, q_end)) return 1;
	size_t tot = 0;
	for (size_t j = q_begin; j < q_end; j++) {
		counts[j - q_begin] = c->homs[j].size();
		tot += c->homs[j].size();
	}
	*total = tot;
	if (buf && cap >= tot) {
		size_t o = 0;
		for (size_t j = q_begin; j < q_end; j++) {
			std::copy(c->homs[j].begin(), c->homs[j].end(), buf + o);
			o += c->homs[j].size();
		}
	}
	return 0;
}

int phylo_import_homologies(phylo_ctx *c, size_t q_begin, size_t q_end, const uint64_t *counts,
							const phylo_homology *buf)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !counts) return c->fail("phylo_import_homologies: bad arguments");
	if (settle_anchor(c)) return 1; // (as phylo_set_homologies)
	if (!c->host_stale.empty()) {
		if (ensure_host_lists(c, 0, q_begin) || ensure_host_lists(c, q_end, c->n)) return 1;
		c->host_stale.clear();
	}
	c->homs_staged = false;
	size_t o = 0;
	for (size_t j = q_begin; j < q_end; j++) {
		if (counts[j - q_begin] && !buf) return c->fail("phylo_import_homologies: null buffer");
		c->homs[j].assign(buf + o, buf + o + counts[j - q_begin]);
		o += counts[j - q_begin];
	}
	return 0;
}

int phylo_export_packed(phylo_ctx *c, size_t q_begin, size_t q_end, uint64_t *counts, phylo_packed_homology *buf,
						size_t cap, size_t *total)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !counts || !total) return c->fail("phylo_export_packed: bad arguments");
	if (ensure_host_lists(c, q_begin, q_end)) return 1;
	size_t tot = 0;
	for (size_t j = q_begin; j < q_end; j++) {
		counts[j - q_begin] = c->homs[j].size();
		tot += c->homs[j].size();
	}
	*total = tot;
	if (buf && cap >= tot) {
		size_t o = 0;
		for (size_t j = q_begin; j < q_end; j++)
			for (const phylo_homology &h : c->homs[j])
				buf[o++] = phylo_packed_homology{(uint32_t)h.index_reference_projected, (uint32_t)h.index_query,
												 (uint32_t)h.length, (uint32_t)h.direction};
	}
	return 0;
}

// packed records → host lists of genomes [q_begin, q_end)
static int unpack_lists(phylo_ctx *c, size_t q_begin, size_t q_end, const uint64_t *counts,
						const phylo_packed_homology *buf)
{
	const uint64_t L = c->L;
	size_t o = 0;
	for (size_t j = q_begin; j < q_end; j++) {
		size_t m = counts[j - q_begin];
		if (m && !buf) return c->fail("phylo_import_packed: null buffer");
		c->homs[j].resize(m);
		for (size_t t = 0; t < m; t++, o++) {
			phylo_homology h;
			h.index_reference_projected = buf[o].start;
			h.index_query = buf[o].index_query;
			h.length = buf[o].length;
			h.direction = (int32_t)buf[o].direction;
			h._pad = 0;
			// inverse of homology::reverseEh (src/process.h:72-80)
			h.index_reference = h.direction ? 2 * L + 1 - h.length - h.index_reference_projected : h.index_reference_projected;
			c->homs[j][t] = h;
		}
	}
	return 0;
}

int phylo_import_packed(phylo_ctx *c, size_t q_begin, size_t q_end, const uint64_t *counts,
						const phylo_packed_homology *buf)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !counts) return c->fail("phylo_import_packed: bad arguments");
	if (!c->have_ref) return c->fail("phylo_import_packed: no reference set");
	if (settle_anchor(c)) return 1; // (as phylo_set_homologies)
	if (!c->host_stale.empty()) {
		for (size_t g = q_begin; g < q_end; g++) c->host_stale[g] = 0; // replaced below
		if (ensure_host_lists(c, 0, q_begin) || ensure_host_lists(c, q_end, c->n)) return 1;
		c->host_stale.clear();
	}
	c->homs_staged = false;
	return unpack_lists(c, q_begin, q_end, counts, buf);
}

static_assert(sizeof(DevHom) == sizeof(phylo_packed_homology), "the wire record is the device record");

// lists that are device-resident already → back to back in query order: one block per genome
__global__ void gather_lists_kernel(const DevHom *__restrict__ src, const uint64_t *__restrict__ desc, DevHom *__restrict__ dst)
{
	const uint64_t begin = desc[3 * blockIdx.x], count = desc[3 * blockIdx.x + 1], to = desc[3 * blockIdx.x + 2];
	for (uint64_t t = threadIdx.x; t < count; t += blockDim.x) dst[to + t] = src[begin + t];
}

int phylo_export_packed_device(phylo_ctx *c, size_t q_begin, size_t q_end, void *dev_dst, size_t cap, uint64_t *counts,
							   size_t *total)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !counts || !total) return c->fail("phylo_export_packed_device: bad arguments");
	HIPOK(c, hipSetDevice(c->device));
	// phase A may have left these lists on the device only (device sort + filter): copy from there
	bool on_device = !c->host_stale.empty() && c->att_homs && q_end > q_begin;
	for (size_t j = q_begin; j < q_end && on_device; j++) on_device = c->host_stale[j] != 0;
	if (on_device && fetch_att_ranges(c)) return 1;
	if (on_device) {
		const size_t m = q_end - q_begin;
		size_t tot = 0;
		for (size_t j = q_begin; j < q_end; j++) {
			counts[j - q_begin] = c->att_count[j];
			tot += c->att_count[j];
		}
		*total = tot;
		if (!dev_dst || cap < tot || !tot) return 0;
		HIPOK(c, c->h_mat.ensure(3 * m + 8));
		HIPOK(c, c->b_subst.ensure(3 * m + 8)); // scratch for the descriptors (the tallies are not live between phases)
		uint64_t *desc = c->h_mat.p, to = 0;
		for (size_t j = q_begin; j < q_end; j++) {
			desc[3 * (j - q_begin)] = c->att_begin[j];
			desc[3 * (j - q_begin) + 1] = c->att_count[j];
			desc[3 * (j - q_begin) + 2] = to;
			to += c->att_count[j];
		}
		HIPOK(c, hipMemcpyAsync(c->b_subst.p, desc, 3 * m * 8, hipMemcpyHostToDevice, c->stream));
		hipLaunchKernelGGL(gather_lists_kernel, dim3((uint32_t)m), dim3(256), 0, c->stream, c->att_homs,
						   (const uint64_t *)c->b_subst.p, (DevHom *)dev_dst);
		HIPOK(c, hipGetLastError());
		return sync_stream(c);
	}
	if (ensure_host_lists(c, q_begin, q_end)) return 1;
	size_t tot = 0;
	for (size_t j = q_begin; j < q_end; j++) {
		counts[j - q_begin] = c->homs[j].size();
		tot += c->homs[j].size();
	}
	*total = tot;
	if (!dev_dst || cap < tot) return 0; // sizing call
	if (!tot) return 0;
	HIPOK(c, c->h_devhom.ensure(tot + 1));
	DevHom *dh = c->h_devhom.p;
	std::vector<size_t> off(q_end - q_begin + 1, 0);
	for (size_t j = q_begin; j < q_end; j++) off[j - q_begin + 1] = off[j - q_begin] + c->homs[j].size();
	workers(c).run(q_end - q_begin, [&](size_t t) {
		size_t o = off[t];
		for (const phylo_homology &h : c->homs[q_begin + t])
			dh[o++] = DevHom{(uint32_t)h.index_reference_projected, (uint32_t)h.index_query, (uint32_t)h.length,
							 (uint32_t)h.direction};
	});
	HIPOK(c, hipMemcpyAsync(dev_dst, dh, tot * sizeof(DevHom), hipMemcpyHostToDevice, c->stream));
	return sync_stream(c);
}

int phylo_attach_packed_device(phylo_ctx *c, const void *dev_records, const uint64_t *begin, const uint64_t *count,
							   size_t keep_begin, size_t keep_end)
{
	if (!c) return 1;
	if (!begin || !count || keep_begin > keep_end || keep_end > c->n) return c->fail("phylo_attach_packed_device: bad arguments");
	if (!c->have_ref) return c->fail("phylo_attach_packed_device: no reference set");
	HIPOK(c, hipSetDevice(c->device));
	// a phase A queued by phylo_anchor_block_device is taken in first: its export kernel still writes h_rng, and a later
	// settle would lay its lists over the ones attached here
	if (settle_anchor(c)) return 1;
	const size_t N = c->n;
	HIPOK(c, c->h_rng.ensure(2 * N));
	HIPOK(c, c->b_hom_rng.ensure(2 * N));
	uint64_t total = 0;
	for (size_t g = 0; g < N; g++) {
		if (begin[g] + count[g] > 0xffffffffull) return c->fail("phylo_attach_packed_device: more than 2^32 records");
		c->h_rng.p[2 * g] = (uint32_t)begin[g];
		c->h_rng.p[2 * g + 1] = (uint32_t)(begin[g] + count[g]);
		total += count[g];
	}
	if (total && !dev_records) return c->fail("phylo_attach_packed_device: null records");
	HIPOK(c, hipMemcpyAsync(c->b_hom_rng.p, c->h_rng.p, 2 * N * 4, hipMemcpyHostToDevice, c->stream));
	// the pileup needs every list sorted by projected start, disjoint and inside the reference (what phase A's
	// filter leaves): a buffer that is anything else is refused here rather than tallied wrongly later
	HIPOK(c, c->b_flag.ensure(8));
	HIPOK(c, hipMemsetAsync(c->b_flag.p + 1, 0, 4, c->stream));
	launch_check_lists((const DevHom *)dev_records, c->b_hom_rng.p, (uint32_t)N, c->L, c->b_flag.p + 1, c->stream);
	uint32_t bad_lists = 0;
	HIPOK(c, hipMemcpyAsync(&bad_lists, c->b_flag.p + 1, 4, hipMemcpyDeviceToHost, c->stream));
	if (sync_stream(c)) return 1;
	if (bad_lists) return c->fail("phylo_attach_packed_device: a genome's list is not sorted by projected start, disjoint and inside the reference");
	c->att_homs = (const DevHom *)dev_records;
	c->att_rng_on_device = false;
	c->att_begin.assign(begin, begin + N);
	c->att_count.assign(count, count + N);
	// lists of the kept range that this context has only on the device so far (device sort +
	// filter) stay to be fetched — from the new buffer, which holds them too
	std::vector<uint8_t> was = c->host_stale;
	c->host_stale.assign(N, 1);
	for (size_t g = keep_begin; g < keep_end; g++) c->host_stale[g] = was.size() == N ? was[g] : 0;
	c->homs_staged = true;
	c->eager_valid = false;
	return 0;
}

// ── the exchange between ranks without the host in it ──
// A rank's block: 4 header words {records in the block, overflow, queries, this rank's phase A needs the host},
// max_queries list lengths, then cap records of 16 bytes.  Every rank's block has the same size, so one all-gather
// assembles all of them; the receiving side works the per-genome ranges out on the device.
static const uint32_t XB_HDR = 4;
// flt_flags / misc (phylo_anchor_block_device: the block is written behind a phase A nobody has waited for): the filter's
// per-query flags (a list with tied starts, which only the host's std::sort orders as the reference does) and the chains'
// counters (misc[3]: scratch overflow) — either makes this block, and with it the pass, one to repeat the long way
__global__ __launch_bounds__(256) void block_export_kernel(const DevHom *__restrict__ src, const uint32_t *__restrict__ rng, uint32_t nq,
															 uint32_t maxq, uint32_t cap, uint32_t *__restrict__ block,
															 const uint32_t *__restrict__ flt_flags, const uint32_t *__restrict__ misc,
															 uint32_t *__restrict__ host_out, uint32_t *__restrict__ zero8)
{
	const uint32_t j = blockIdx.x; // a slot of the header: the block's query j, or padding
	__shared__ uint32_t part[256];
	uint32_t acc = 0;
	for (uint32_t t = threadIdx.x; t < j && t < nq; t += blockDim.x) acc += rng[2 * t + 1] - rng[2 * t];
	part[threadIdx.x] = acc;
	__syncthreads();
	for (uint32_t s2 = 128; s2 > 0; s2 >>= 1) {
		if (threadIdx.x < s2) part[threadIdx.x] += part[threadIdx.x + s2];
		__syncthreads();
	}
	const uint32_t off = part[0];
	const uint32_t cnt = j < nq ? rng[2 * j + 1] - rng[2 * j] : 0u;
	if (j + 1 == maxq) { // the header (no memset before this kernel: every word of it is written here, once)
		uint32_t needs_host = 0;
		if (flt_flags) {
			for (uint32_t t = threadIdx.x; t < nq; t += blockDim.x) needs_host |= flt_flags[t];
			if (threadIdx.x == 0) needs_host |= misc[3];
		}
		needs_host = (uint32_t)__syncthreads_or((int)(needs_host != 0));
		if (threadIdx.x == 0) {
			block[0] = off + cnt;
			block[1] = off + cnt > cap ? 1u : 0u; // (the slots' sums ascend: the last one holds the block's total)
			block[2] = nq;
			block[3] = needs_host ? 1u : 0u;
		}
	}
	if (threadIdx.x == 0) block[XB_HDR + j] = cnt;
	if (host_out) { // what anchor_impl's copies bring to the host on the path that waits (h_rng's layout)
		if (threadIdx.x == 0 && j < nq) {
			host_out[2 * j] = rng[2 * j];
			host_out[2 * j + 1] = rng[2 * j + 1];
			host_out[2 * nq + 1 + j] = flt_flags[j];
		}
		if (j == 0 && threadIdx.x < 8) host_out[3 * nq + 1 + threadIdx.x] = misc[threadIdx.x];
		if (j == 0 && threadIdx.x == 8) host_out[2 * nq] = flt_flags[-1]; // (a_flt[0]: the lists' total)
	}
	if (zero8 && j == 0 && threadIdx.x < 8) zero8[threadIdx.x] = 0; // b_flag for the attach and the comparison that follow
	if (off + cnt > cap) return;
	DevHom *dst = (DevHom *)(block + XB_HDR + maxq) + off;
	const DevHom *from = src + (j < nq ? rng[2 * j] : 0u);
	for (uint32_t t = threadIdx.x; t < cnt; t += blockDim.x) dst[t] = from[t];
}
// one thread block per rank: the ranges of its genomes in the gathered buffer (in records from the buffer's start)
__global__ __launch_bounds__(256) void block_attach_kernel(const uint32_t *__restrict__ all, const uint32_t *__restrict__ bounds,
															 uint32_t maxq, uint32_t cap, uint32_t *__restrict__ rng, uint32_t *__restrict__ flags)
{
	const uint32_t r = blockIdx.x;
	const uint32_t words = XB_HDR + maxq + 4u * cap;
	const uint32_t *blk = all + (size_t)r * words;
	const uint32_t g0 = bounds[r], nq = bounds[r + 1] - g0;
	// an overflowed or mismatched block: reported (flags[2]), and its genomes' lists are left empty — the lengths in
	// its header reach beyond the records it holds, and the kernels that follow must not read there
	const bool usable = !(blk[1] || blk[2] != nq || blk[0] > cap);
	if (threadIdx.x == 0 && !usable) flags[2] = 1;
	if (threadIdx.x == 0 && blk[3]) flags[6] = 1; // that rank's phase A needs the host: the pass is repeated (TRI_TAIL)
	__shared__ uint32_t carry;
	__shared__ uint32_t scan[256];
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	const uint32_t base = (uint32_t)(((size_t)r * words + XB_HDR + maxq) / 4u);
	for (uint32_t t0 = 0; t0 < nq; t0 += 256) {
		const uint32_t t = t0 + threadIdx.x;
		const uint32_t cnt = (usable && t < nq) ? blk[XB_HDR + t] : 0u;
		scan[threadIdx.x] = cnt;
		__syncthreads();
		for (uint32_t d = 1; d < 256; d <<= 1) {
			const uint32_t v = threadIdx.x >= d ? scan[threadIdx.x - d] : 0u;
			__syncthreads();
			scan[threadIdx.x] += v;
			__syncthreads();
		}
		const uint32_t begin = carry + scan[threadIdx.x] - cnt;
		if (t < nq) {
			const bool inside = begin + cnt <= cap; // (lengths that do not add up to the header's total)
			if (!inside) flags[2] = 1;
			rng[2 * (g0 + t)] = base + (inside ? begin : 0u);
			rng[2 * (g0 + t) + 1] = base + (inside ? begin + cnt : 0u);
		}
		__syncthreads();
		if (threadIdx.x == 255) carry += scan[255];
		__syncthreads();
	}
}

size_t phylo_exchange_block_bytes(size_t max_queries, size_t cap_records) { return (XB_HDR + max_queries + 4 * cap_records) * 4; }

int phylo_export_block_device(phylo_ctx *c, size_t q_begin, size_t q_end, void *dev_block, size_t max_queries, size_t cap_records)
{
	if (!c) return 1;
	if (q_begin > q_end || q_end > c->n || !dev_block) return c->fail("phylo_export_block_device: bad arguments");
	const size_t nq = q_end - q_begin;
	if (max_queries < nq || max_queries % 4 || max_queries == 0) return c->fail("phylo_export_block_device: max_queries must be a multiple of 4 and hold the block's queries");
	if (XB_HDR + max_queries + 4 * cap_records >= 0xffffffffull) return c->fail("phylo_export_block_device: block too large");
	if (settle_anchor(c)) return 1;
	// the lists must be where phase A's device filter left them: this context's buffer, ranges by local query index
	bool on_device = !c->host_stale.empty() && c->att_homs == c->b_homs.p && c->plan_valid && c->plan_qb == q_begin && c->plan_qe == q_end;
	for (size_t j = q_begin; j < q_end && on_device; j++) on_device = c->host_stale[j] != 0;
	HIPOK(c, hipSetDevice(c->device));
	if (!on_device) {
		// the lists live on the host (a query with tied starts went through std::sort, the host filter was asked for,
		// lists were installed by the caller): the block is put together here and uploaded — the other ranks' blocks
		// do not care how this one was made
		if (ensure_host_lists(c, q_begin, q_end)) return 1;
		size_t tot = 0;
		for (size_t j = q_begin; j < q_end; j++) tot += c->homs[j].size();
		const bool over = tot > cap_records;
		const size_t words = XB_HDR + max_queries + (over ? 0 : 4 * tot);
		HIPOK(c, c->h_devhom.ensure(words / 4 + 2));
		uint32_t *blk = (uint32_t *)c->h_devhom.p;
		blk[0] = (uint32_t)tot;
		blk[1] = over ? 1u : 0u;
		blk[2] = (uint32_t)nq;
		blk[3] = 0;
		for (size_t t = 0; t < max_queries; t++) blk[XB_HDR + t] = t < nq ? (uint32_t)c->homs[q_begin + t].size() : 0u;
		if (!over) {
			DevHom *rec = (DevHom *)(blk + XB_HDR + max_queries);
			for (size_t j = q_begin; j < q_end; j++)
				for (const phylo_homology &h : c->homs[j])
					*rec++ = DevHom{(uint32_t)h.index_reference_projected, (uint32_t)h.index_query, (uint32_t)h.length, (uint32_t)h.direction};
		}
		HIPOK(c, hipMemcpyAsync(dev_block, blk, words * 4, hipMemcpyHostToDevice, c->stream));
		return sync_stream(c); // (the staging buffer is reused by other calls)
	}
	return queue_block_export(c, nq, dev_block, max_queries, cap_records, nullptr, nullptr, nullptr);
}

int phyabi::queue_block_export(phylo_ctx *c, size_t nq, void *dev_block, size_t max_queries, size_t cap_records, const uint32_t *flt_flags,
							   const uint32_t *misc, uint32_t *host_out)
{
	uint32_t *host_dev = nullptr, *zero8 = nullptr;
	if (host_out) { // (the queued pass: page-locked words as the device addresses them; b_flag zeroed on the way)
		HIPOK(c, hipHostGetDevicePointer((void **)&host_dev, host_out, 0));
		HIPOK(c, c->b_flag.ensure(8));
		zero8 = c->b_flag.p;
		c->flags_zeroed = true;
	}
	hipLaunchKernelGGL(block_export_kernel, dim3((uint32_t)max_queries), dim3(256), 0, c->stream, (const DevHom *)c->b_homs.p,
					   (const uint32_t *)c->b_hom_rng.p, (uint32_t)nq, (uint32_t)max_queries, (uint32_t)cap_records, (uint32_t *)dev_block, flt_flags,
					   misc, host_dev, zero8);
	HIPOK(c, hipGetLastError());
	return 0;
}

int phyabi::fetch_att_ranges(phylo_ctx *c)
{
	if (settle_anchor(c)) return 1;
	if (!c->att_rng_on_device) return 0;
	const size_t N = c->n;
	std::vector<uint32_t> r(2 * N);
	uint32_t fl[2] = {0, 0};
	HIPOK(c, hipSetDevice(c->device));
	HIPOK(c, hipMemcpyAsync(r.data(), c->b_hom_rng.p, 2 * N * 4, hipMemcpyDeviceToHost, c->stream));
	if (c->att_unchecked) HIPOK(c, hipMemcpyAsync(fl, c->b_flag.p + 1, 8, hipMemcpyDeviceToHost, c->stream));
	HIPOK(c, hipStreamSynchronize(c->stream));
	if (c->att_unchecked && (fl[0] || fl[1])) return c->fail("the lists gathered from the ranks are not usable (overflowed block or unsorted list)");
	c->att_begin.assign(N, 0);
	c->att_count.assign(N, 0);
	for (size_t g = 0; g < N; g++) {
		c->att_begin[g] = r[2 * g];
		c->att_count[g] = r[2 * g + 1] - r[2 * g];
	}
	c->att_rng_on_device = false;
	return 0;
}

int phylo_attach_blocks_device(phylo_ctx *c, const void *dev_all, size_t world, const size_t *bounds, size_t max_queries,
							   size_t cap_records, size_t keep_begin, size_t keep_end)
{
	if (!c) return 1;
	if (!dev_all || !world || !bounds || keep_begin > keep_end || keep_end > c->n) return c->fail("phylo_attach_blocks_device: bad arguments");
	if (!c->have_ref) return c->fail("phylo_attach_blocks_device: no reference set");
	if (bounds[0] != 0 || bounds[world] != c->n) return c->fail("phylo_attach_blocks_device: the blocks must cover all genomes");
	const size_t words = XB_HDR + max_queries + 4 * cap_records;
	if (max_queries % 4 || world * words >= 0xffffffffull) return c->fail("phylo_attach_blocks_device: bad block shape");
	for (size_t r = 0; r < world; r++)
		if (bounds[r + 1] < bounds[r] || bounds[r + 1] - bounds[r] > max_queries) return c->fail("phylo_attach_blocks_device: bad bounds");
	HIPOK(c, hipSetDevice(c->device));
	const size_t N = c->n;
	HIPOK(c, c->b_hom_rng.ensure(2 * N + world + 2));
	HIPOK(c, c->b_flag.ensure(8));
	// (a phase A queued by phylo_anchor_block_device: its lists are in the gathered buffer now and what its flags say went
	// round with its block; its statistics are collected at the next wait.  h_rng still receives its copies: the bounds go elsewhere)
	const bool queued = c->anchor_pending && c->pend_range; // (then the host has none of this rank's own lists yet either)
	if (queued) c->pend_stats_only = true;
	uint32_t *d_bounds = c->b_hom_rng.p + 2 * N;
	bool same = c->xb_bounds_at == d_bounds && c->xb_bounds.size() == world + 1;
	for (size_t r = 0; r <= world && same; r++) same = c->xb_bounds[r] == (uint32_t)bounds[r];
	if (!same) { // (the ranks' bounds change with the genomes, not from pass to pass)
		HIPOK(c, c->h_cnt.ensure(world + 8));
		uint32_t *hb = c->h_cnt.p; // pinned: the copy below must not wait for pageable staging
		c->xb_bounds.resize(world + 1);
		for (size_t r = 0; r <= world; r++) hb[r] = c->xb_bounds[r] = (uint32_t)bounds[r];
		HIPOK(c, hipMemcpyAsync(d_bounds, hb, (world + 1) * 4, hipMemcpyHostToDevice, c->stream));
		c->xb_bounds_at = d_bounds;
	}
	if (!(queued && c->flags_zeroed)) { // (a queued pass: its block export zeroed the flags)
		HIPOK(c, hipMemsetAsync(c->b_flag.p + 1, 0, 8, c->stream));
		HIPOK(c, hipMemsetAsync(c->b_flag.p + 6, 0, 4, c->stream));
	}
	c->flags_zeroed = false;
	hipLaunchKernelGGL(block_attach_kernel, dim3((uint32_t)world), dim3(256), 0, c->stream, (const uint32_t *)dev_all, d_bounds,
					   (uint32_t)max_queries, (uint32_t)cap_records, c->b_hom_rng.p, c->b_flag.p);
	launch_check_lists((const DevHom *)dev_all, c->b_hom_rng.p, (uint32_t)N, c->L, c->b_flag.p + 1, c->stream);
	HIPOK(c, hipGetLastError());
	// nothing is waited for: b_flag[1] (a list that is not sorted and disjoint) and b_flag[2] (a block that overflowed
	// its capacity) are read with the result of the comparison that follows
	c->att_homs = (const DevHom *)dev_all;
	c->att_rng_on_device = true;
	c->att_unchecked = true;
	std::vector<uint8_t> was = c->host_stale;
	c->host_stale.assign(N, 1);
	for (size_t g = keep_begin; g < keep_end && !queued; g++) c->host_stale[g] = was.size() == N ? was[g] : 0;
	c->homs_staged = true;
	c->eager_valid = false;
	return 0;
}

int phylo_complete_delete(phylo_ctx *c)
{
	if (!c) return 1;
	if (ensure_host_lists(c, 0, c->n)) return 1;
	c->host_stale.clear();
	c->homs = complete_delete(c->homs);
	c->homs_staged = false;
	return 0;
}

} // extern "C"

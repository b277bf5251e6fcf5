// abi_compare.hip — phase B of process() (/root/reference/src/process.cxx:517-549): the pileup (projection + pair
// tallies; pileup_kernels.hip) or the explicit segment list (seqcmp_kernels.hip), the result's way home, and both
// phases as one call (phylo_anchor_compare, phylo_process).
#include "abi_ctx.hpp"

using namespace phy;
using namespace phyabi;

extern "C" {

// The pileup of part `part` of `nparts`: a range of 64-window tiles of the reference.
// Every part projects and compares ALL genomes over its own range, so both kernels
// shrink with the number of parts and the partial tallies simply add up.
int phyabi::make_pileup(phylo_ctx *c, size_t part, size_t nparts, Pileup *out)
{
	Pileup P;
	P.N = (uint32_t)c->n;
	P.Npad = (uint32_t)((c->n + 63) / 64 * 64);
	P.L = c->L;
	uint32_t Wall = (c->L + 31) / 32;
	uint32_t ntile = (Wall + 63) / 64;
	uint32_t t0 = (uint32_t)((uint64_t)ntile * part / nparts), t1 = (uint32_t)((uint64_t)ntile * (part + 1) / nparts);
	P.w0 = t0 * 64;
	uint32_t wend = std::min<uint32_t>(Wall, t1 * 64);
	P.W = wend > P.w0 ? wend - P.w0 : 0;
	size_t plane_words = (size_t)P.W * P.Npad;
	HIPOK(c, c->b_planes.ensure(plane_words * 5));
	for (int p = 0; p < 5; p++) P.plane[p] = c->b_planes.p + plane_words * p;
	*out = P;
	return 0;
}

// compare(list, list) of process.cxx:566-611 as a segment generator (host):
// every overlapping (ha, hb) pair becomes one seqcmp / revseqcmp segment
// (process.cxx:620-658).  Used by the segment-list backend.
static void pair_segments(const phylo_ctx *c, size_t i, size_t j, std::vector<Segment> &out)
{
	const auto &ha = c->homs[i], &hb = c->homs[j];
	size_t right = 0;
	for (const phylo_homology &h : ha) {
		uint64_t hs = h.index_reference_projected, he = hs + h.length;
		while (right < hb.size() && hb[right].index_reference_projected + hb[right].length <= hs) right++;
		for (size_t r = right; r < hb.size(); r++) {
			const phylo_homology &o = hb[r];
			uint64_t os = o.index_reference_projected, oe = os + o.length;
			if (os >= he) break;
			uint64_t cs = std::max(hs, os), ce = std::min(he, oe);
			if (cs >= ce) continue;
			phylo_homology hat = trim_homology(h, cs, ce), hbt = trim_homology(o, cs, ce);
			Segment sg;
			sg.len = (uint32_t)(ce - cs);
			if (h.direction == o.direction) {
				sg.a = c->goff[i] + hat.index_query;
				sg.b = c->goff[j] + hbt.index_query;
				sg.rev = 0;
			} else if (o.direction == 1) { // account_rev(sa + hat.start_query(), sb, hbt.end_query(), n)
				sg.a = c->goff[i] + hat.index_query;
				sg.b = c->goff[j] + hbt.index_query + hbt.length - sg.len;
				sg.rev = 1;
			} else {
				sg.a = c->goff[j] + hbt.index_query;
				sg.b = c->goff[i] + hat.index_query + hat.length - sg.len;
				sg.rev = 1;
			}
			out.push_back(sg);
		}
	}
}

static int compare_segments(phylo_ctx *c, size_t part, size_t nparts, uint64_t *subst, uint64_t *homologs)
{
	if (ensure_host_lists(c, 0, c->n)) return 1; // (settles a queued phase A first)
	size_t N = c->n;
	std::vector<Segment> segs;
	std::vector<uint32_t> seg_pair; // pair index of each segment
	std::vector<std::pair<uint32_t, uint32_t>> pairs;
	size_t pid = 0;
	for (size_t i = 0; i < N; i++)
		for (size_t j = i + 1; j < N; j++, pid++) {
			if (pid % nparts != part) continue;
			size_t before = segs.size();
			pair_segments(c, i, j, segs);
			pairs.emplace_back((uint32_t)i, (uint32_t)j);
			seg_pair.resize(segs.size(), (uint32_t)(pairs.size() - 1));
			(void)before;
		}
	std::vector<uint64_t> out(segs.size());
	if (!segs.empty() && run_segments(c, c->d_genomes, segs.data(), segs.size(), out.data(), "seqcmp_batch")) return 1;
	double sites = 0;
	for (size_t s = 0; s < segs.size(); s++) {
		auto pr = pairs[seg_pair[s]];
		size_t a = (size_t)pr.first * N + pr.second, b = (size_t)pr.second * N + pr.first;
		subst[a] += out[s];
		homologs[a] += segs[s].len;
		subst[b] = subst[a];
		homologs[b] = homologs[a];
		sites += segs[s].len;
	}
	c->stats["count:compare_sites"] += sites;
	c->stats["count:segments"] += (double)segs.size();
	return 0;
}

// dev_out: leave the tallies in the caller's device buffers (subst / homologs are device pointers)
// The result on its way to the host: both matrices as symmetric u32 (a tally is at most the reference's length,
// below 2^31) — half the bytes of the u64 matrices across PCIe; the host widens them row by row, which is a streaming
// pass.  (Mirroring on the host instead is a strided walk over 16 MB: measured 0.5-2 ms at N = 1024, round 3.)
__global__ __launch_bounds__(256) void sym32_from_matrices_kernel(uint32_t N, const unsigned long long *__restrict__ s,
																   const unsigned long long *__restrict__ h, uint32_t *__restrict__ out)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, NN = (uint64_t)N * N;
	if (t >= NN) return;
	const uint32_t i = (uint32_t)(t / N), j = (uint32_t)(t % N);
	const uint64_t src = i < j ? t : (uint64_t)j * N + i;
	out[t] = i == j ? 0u : (uint32_t)s[src];
	out[NN + t] = i == j ? 0u : (uint32_t)h[src];
}
__global__ __launch_bounds__(256) void sym32_from_triangle_kernel(uint32_t N, const uint32_t *__restrict__ tri, uint32_t *__restrict__ out)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, NN = (uint64_t)N * N;
	if (t >= NN) return;
	const uint32_t i = (uint32_t)(t / N), j = (uint32_t)(t % N);
	const uint32_t a = i < j ? i : j, b = i < j ? j : i;
	const uint64_t P = (uint64_t)N * (N - 1) / 2, k = (uint64_t)a * (2ull * N - a - 1) / 2 + (b - a - 1);
	out[t] = i == j ? 0u : tri[k];
	out[NN + t] = i == j ? 0u : tri[P + k];
}
// The tallies (upper triangles of two N x N u64 matrices on the device) straight into the result's page-locked home
// (phylo_result_matrices) as the two symmetric matrices process() returns: 16-byte stores over PCIe, no staging copy and
// no pass on the host cores.  A thread takes two neighbouring columns of one row.
__global__ __launch_bounds__(256) void matrices_to_home_kernel(uint32_t N, const unsigned long long *__restrict__ s, const unsigned long long *__restrict__ h,
																unsigned long long *__restrict__ ds, unsigned long long *__restrict__ dh)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, half = ((uint64_t)N + 1) / 2;
	if (t >= (uint64_t)N * half) return;
	const uint32_t i = (uint32_t)(t / half), j0 = (uint32_t)(t % half) * 2u;
	unsigned long long vs[2] = {0, 0}, vh[2] = {0, 0};
#pragma unroll
	for (uint32_t e = 0; e < 2; e++) {
		const uint32_t j = j0 + e;
		if (j < N && j != i) {
			const uint64_t src = i < j ? (uint64_t)i * N + j : (uint64_t)j * N + i;
			vs[e] = s[src];
			vh[e] = h[src];
		}
	}
	const uint64_t o = (uint64_t)i * N + j0;
	if (j0 + 1 < N && (o & 1) == 0) {
		*(ulonglong2 *)(ds + o) = ulonglong2{vs[0], vs[1]};
		*(ulonglong2 *)(dh + o) = ulonglong2{vh[0], vh[1]};
	} else {
		ds[o] = vs[0];
		dh[o] = vh[0];
		if (j0 + 1 < N) {
			ds[o + 1] = vs[1];
			dh[o + 1] = vh[1];
		}
	}
}

// sym (2 N^2 u32 in pinned memory) -> the caller's two N x N u64 matrices; returns the sum of the homologs matrix
static double widen_result(phylo_ctx *c, const uint32_t *sym, uint64_t *subst, uint64_t *homologs)
{
	const size_t N = c->n, NN = N * N, parts = NN >= ((size_t)1 << 18) ? 32 : 1;
	std::vector<double> part_sites(parts, 0.0);
	auto widen = [&](size_t t) {
		const size_t a = NN * t / parts, b = NN * (t + 1) / parts;
		for (size_t k = a; k < b; k++) subst[k] = sym[k];
		uint64_t acc = 0;
		for (size_t k = a; k < b; k++) {
			const uint32_t v = sym[NN + k];
			homologs[k] = v;
			acc += v;
		}
		part_sites[t] = (double)acc;
	};
	if (parts > 1) workers(c).run(parts, widen);
	else widen(0);
	double sites = 0;
	for (double v : part_sites) sites += v;
	return sites;
}

// u32 upper triangle: tri[k] = substitutions, tri[P + k] = homologs of pair (i < j), k = i (2N - i - 1) / 2 + (j - i - 1):
// what crosses the wire between ranks (a tally is at most the reference's length, which is below 2^31)
// flag (may be null): the TRI_TAIL words of the part's report go behind the triangle on the way — what this part's comparison
// has to say, in a form that adds up over the parts like the tallies do: {the projection's list of '!' overflowed, a gathered
// list is not sorted and disjoint, a gathered block overflowed its capacity, 1 per part, a rank's phase A needs the host
// (phylo_anchor_block_device: the verdict rode in its block's header), 0, 0, 0}.  A rank queues its whole comparison and the
// collective behind it without a host round trip; whoever reads the summed triangle reads the ranks' reports with it.
__global__ __launch_bounds__(256) void pack_triangle_kernel(uint32_t N, const unsigned long long *__restrict__ s,
															 const unsigned long long *__restrict__ h, uint32_t *__restrict__ tri,
															 const uint32_t *__restrict__ flag, int att)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (flag && t == 0) {
		uint32_t *tail = tri + (uint64_t)N * (N - 1);
		tail[0] = (flag[0] & 2u) ? 1u : 0u;
		tail[1] = att && flag[1] ? 1u : 0u;
		tail[2] = att && flag[2] ? 1u : 0u;
		tail[3] = 1u;
		tail[4] = att && flag[6] ? 1u : 0u;
		tail[5] = tail[6] = tail[7] = 0u;
	}
	if (t >= (uint64_t)N * N) return;
	const uint32_t i = (uint32_t)(t / N), j = (uint32_t)(t % N);
	if (i >= j) return;
	const uint64_t P = (uint64_t)N * (N - 1) / 2, k = (uint64_t)i * (2ull * N - i - 1) / 2 + (j - i - 1);
	tri[k] = (uint32_t)s[t];
	tri[P + k] = (uint32_t)h[t];
}

// A (summed) triangle straight into the caller's two n x n u64 matrices in host memory that the device can address
// (registered by phylo_triangle_to_matrices): 16-byte stores over PCIe, no staging copy and no pass on the host cores.
// A thread takes two neighbouring columns of one row; *sites += the homologs it wrote.
__global__ __launch_bounds__(256) void triangle_to_host_kernel(uint32_t N, const uint32_t *__restrict__ tri, unsigned long long *__restrict__ s,
																 unsigned long long *__restrict__ h, unsigned long long *__restrict__ sites)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, half = ((uint64_t)N + 1) / 2;
	unsigned long long mine = 0;
	if (t < (uint64_t)N * half) {
		const uint32_t i = (uint32_t)(t / half), j0 = (uint32_t)(t % half) * 2u;
		const uint64_t P = (uint64_t)N * (N - 1) / 2;
		unsigned long long vs[2] = {0, 0}, vh[2] = {0, 0};
#pragma unroll
		for (uint32_t e = 0; e < 2; e++) {
			const uint32_t j = j0 + e;
			if (j < N && j != i) {
				const uint32_t a = i < j ? i : j, b = i < j ? j : i;
				const uint64_t k = (uint64_t)a * (2ull * N - a - 1) / 2 + (b - a - 1);
				vs[e] = tri[k];
				vh[e] = tri[P + k];
			}
		}
		const uint64_t o = (uint64_t)i * N + j0;
		if (j0 + 1 < N && (o & 1) == 0) {
			*(ulonglong2 *)(s + o) = ulonglong2{vs[0], vs[1]};
			*(ulonglong2 *)(h + o) = ulonglong2{vh[0], vh[1]};
		} else {
			s[o] = vs[0];
			h[o] = vh[0];
			if (j0 + 1 < N) {
				s[o + 1] = vs[1];
				h[o + 1] = vh[1];
			}
		}
		mine = vh[0] + vh[1];
	}
	for (int d = 32; d; d >>= 1) mine += __shfl_xor(mine, d, 64);
	if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(sites, mine);
}

// out_mode 0: the caller's host matrices; 1: the caller's device matrices (subst / homologs are device pointers);
// 2: the caller's device u32 triangle + its four flag words (subst is the device pointer, homologs unused)
static int compare_pileup(phylo_ctx *c, size_t part, size_t nparts, uint64_t *subst, uint64_t *homologs, int out_mode = 0)
{
	const bool dev_out = out_mode != 0;
	size_t N = c->n;
	hipStream_t st = c->stream;
	// (a phase A queued by phylo_anchor_block_device whose blocks were never gathered and attached: its lists first)
	if (c->anchor_pending && c->pend_range && !c->pend_stats_only && settle_anchor(c)) return 1;
	Pileup P;
	if (make_pileup(c, part, nparts, &P)) return 1;

	// filtered homologies → device, unless phase A staged them there already
	double t0 = now_ms();
	if (!c->homs_staged) {
		if (ensure_host_lists(c, 0, N)) return 1;
		c->host_stale.clear();
		c->att_homs = nullptr;
		c->att_rng_on_device = false;
		std::vector<uint32_t> hom_rng(2 * N);
		size_t tot = 0;
		for (size_t g = 0; g < N; g++) {
			hom_rng[2 * g] = (uint32_t)tot;
			tot += c->homs[g].size();
			hom_rng[2 * g + 1] = (uint32_t)tot;
		}
		HIPOK(c, c->h_devhom.ensure(tot + 1));
		DevHom *dh = c->h_devhom.p;
		std::atomic<size_t> bad{(size_t)-1};
		std::atomic<bool> entangled{false};
		workers(c).run(N, [&](size_t g) {
			size_t o = hom_rng[2 * g];
			uint64_t prev_end = 0;
			for (const phylo_homology &h : c->homs[g]) {
				if (h.index_reference_projected + h.length > c->L) bad = g;
				if (h.index_reference_projected < prev_end) entangled = true;
				prev_end = h.index_reference_projected + h.length;
				dh[o++] = DevHom{(uint32_t)h.index_reference_projected, (uint32_t)h.index_query, (uint32_t)h.length,
								 (uint32_t)h.direction};
			}
		});
		if (bad != (size_t)-1) return c->fail("genome %zu: homology reaches beyond the reference", bad.load());
		if (entangled) {
			// Lists installed by the caller (phylo_set_homologies, phylo_import_*) that are not sorted and disjoint on
			// the reference: the pileup would not be compare(list, list) of process.cxx:566-611 for them — the segment
			// backend, which restates that merge-join literally, takes the call.
			c->stats["count:compare_calls_rerouted_to_segments"] += 1;
			if (dev_out) {
				std::vector<uint64_t> hs(N * N, 0), hh(N * N, 0);
				if (compare_segments(c, part, nparts, hs.data(), hh.data())) return 1;
				HIPOK(c, hipMemcpy(subst, hs.data(), N * N * 8, hipMemcpyHostToDevice));
				HIPOK(c, hipMemcpy(homologs, hh.data(), N * N * 8, hipMemcpyHostToDevice));
				return 0;
			}
			std::fill(subst, subst + N * N, 0);
			std::fill(homologs, homologs + N * N, 0);
			return compare_segments(c, part, nparts, subst, homologs);
		}
		HIPOK(c, c->b_hom_rng.ensure(2 * N));
		HIPOK(c, c->b_homs.ensure(tot + 1));
		HIPOK(c, hipMemcpyAsync(c->b_hom_rng.p, hom_rng.data(), 2 * N * 4, hipMemcpyHostToDevice, st));
		if (tot) HIPOK(c, hipMemcpyAsync(c->b_homs.p, dh, tot * sizeof(DevHom), hipMemcpyHostToDevice, st));
		if (sync_stream(c)) return 1; // hom_rng goes out of scope
		c->stats["ms:compare_hom_upload"] += now_ms() - t0;
	}
	HIPOK(c, c->b_flag.ensure(8));
	HIPOK(c, c->b_first.ensure(project_index_entries(P) + 1));
	unsigned long long *acc_s, *acc_h; // where the pair kernel accumulates
	if (out_mode == 1) {
		acc_s = (unsigned long long *)subst;
		acc_h = (unsigned long long *)homologs;
	} else {
		HIPOK(c, c->b_subst.ensure(2 * N * N)); // both tallies in one buffer: one fill
		acc_s = c->b_subst.p;
		acc_h = c->b_subst.p + N * N;
	}
	HIPOK(c, c->h_mat.ensure(2 * N * N + 8));
	// the caller's matrices are the result's page-locked home (phylo_result_matrices): the device writes them itself
	const bool to_home = out_mode == 0 && c->res.map && c->res.n == N && subst == c->res.subst() && homologs == c->res.homologs();
	if (out_mode == 0 && !to_home) HIPOK(c, c->b_sym32.ensure(2 * N * N + 4));
	auto zero_tallies = [&]() -> hipError_t {
		if (acc_h == acc_s + N * N) return hipMemsetAsync(acc_s, 0, 2 * N * N * 8, st);
		const hipError_t e = hipMemsetAsync(acc_s, 0, N * N * 8, st);
		return e != hipSuccess ? e : hipMemsetAsync(acc_h, 0, N * N * 8, st);
	};
	HIPOK(c, zero_tallies());
	const DevHom *dev_homs = c->att_homs ? c->att_homs : c->b_homs.p;
	// phase A may have projected the lists already (whole reference, i.e. part 0 of 1)
	const bool projected = c->homs_staged && c->eager_valid && part == 0 && nparts == 1;
	HIPOK(c, c->b_bang.ensure(2 * (size_t)c->bang_cap + 2));
	if (!projected) {
		if (N && P.W) { // (the tile index zeroes the projection's flag words on the way: [0] its '!' flag, [3] the count of listed '!')
			launch_tile_index(P, query_src(c), dev_homs, c->b_hom_rng.p, c->b_first.p, 0, (uint32_t)N, st, c->b_flag.p);
		} else {
			HIPOK(c, hipMemsetAsync(c->b_flag.p, 0, 4, st)); // (words 1 and 2 belong to phylo_attach_blocks_device)
			HIPOK(c, hipMemsetAsync(c->b_flag.p + 3, 0, 4, st));
		}
	}
	// Three planes and the plain pair kernel unless '!' turns up among the projected positions (the projection raises
	// a flag); then all five planes and the kernel that reads D and B.  The context remembers the outcome of its last
	// call and goes ahead on that assumption — projection, pairs, mirror image and the copy back are queued without
	// waiting for the flag, which is read with the result; only when it says '!' and the plain kernel ran is the
	// work repeated with five planes (once per set of genomes: the next call expects it).  The other way round —
	// five planes and no '!' after all — the result is right as it stands, since B is empty.
	uint32_t *flagp = (uint32_t *)(c->h_mat.p + 2 * N * N);
	// pair tiles (ig, jt) holding at least one pair i<j
	std::vector<uint32_t> tiles;
	uint32_t nig = (uint32_t)((N + PAIR_IG - 1) / PAIR_IG), njt = (uint32_t)((N + PAIR_JT - 1) / PAIR_JT);
	for (uint32_t ig = 0; ig < nig; ig++)
		for (uint32_t jt = 0; jt < njt; jt++) {
			if ((uint64_t)ig * PAIR_IG >= (uint64_t)jt * PAIR_JT + PAIR_JT - 1) continue; // no i<j inside
			tiles.push_back((ig << 16) | jt);
		}
	// (the tile lists depend on the number of genomes and the pair kernel only: uploaded when those change)
	const uint64_t tiles_key = (uint64_t)N * 2u + (c->opt_pairs_kernel == 0 ? 1u : 0u) + 1u;
	if (!tiles.empty()) HIPOK(c, c->b_tiles.ensure(tiles.size() + (N / 64 + 2) * (N / 64 + 2))); // room for the matrix-core kernel's tiles behind them
	// tiles of the matrix-core kernel: 64 x 64 genomes, ti <= tj
	std::vector<uint32_t> mtiles;
	if (c->opt_pairs_kernel == 0) {
		const uint32_t T = pairs_mfma_tile(), nt = (uint32_t)((N + T - 1) / T);
		for (uint32_t a = 0; a < nt; a++)
			for (uint32_t b = a; b < nt; b++) mtiles.push_back((a << 16) | b);
		HIPOK(c, c->b_tiles.ensure(tiles.size() + mtiles.size()));
	}
	if (c->tiles_key != tiles_key || c->tiles_at != c->b_tiles.p) {
		if (!tiles.empty()) HIPOK(c, hipMemcpyAsync(c->b_tiles.p, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice, st));
		if (!mtiles.empty()) HIPOK(c, hipMemcpyAsync(c->b_tiles.p + tiles.size(), mtiles.data(), mtiles.size() * 4, hipMemcpyHostToDevice, st));
		if (sync_stream(c)) return 1; // (the vectors are this call's)
		c->tiles_key = tiles_key;
		c->tiles_at = c->b_tiles.p;
	}
	bool do_correct = false; // three planes under the matrix-core kernel: the listed '!' are settled before the tallies leave
	bool queued_report = false; // the part's report is written with the packed triangle (out_mode 2, matrix-core path)
	auto finish_tallies = [&]() { // the packed triangle for the wire; mirror images for the matrices (u32 on the way to the host)
		if (do_correct && c->bang_cap) {
			KernelSpan s(c, "pileup_bang_correct");
			launch_bang_correct(P, query_src(c), dev_homs, c->b_hom_rng.p, c->b_bang.p, c->b_flag.p + 3, c->bang_cap, acc_s, st);
		}
		if (out_mode == 2) // (the queued path: the part's report rides behind the triangle; the '!' corrections before it may have raised flag[0])
			hipLaunchKernelGGL(pack_triangle_kernel, dim3((uint32_t)((N * N + 255) / 256)), dim3(256), 0, st, (uint32_t)N, acc_s, acc_h, (uint32_t *)subst,
							   queued_report ? c->b_flag.p : (const uint32_t *)nullptr, c->att_unchecked ? 1 : 0);
		else if (out_mode == 1)
			launch_symmetrise((uint32_t)N, acc_s, acc_h, st);
		else if (to_home)
			hipLaunchKernelGGL(matrices_to_home_kernel, dim3((uint32_t)((N * ((N + 1) / 2) + 255) / 256)), dim3(256), 0, st, (uint32_t)N, acc_s, acc_h,
							   (unsigned long long *)((char *)c->res.dev + 4096), (unsigned long long *)((char *)c->res.dev + 4096) + c->res.matrix_words());
		else
			hipLaunchKernelGGL(sym32_from_matrices_kernel, dim3((uint32_t)((N * N + 255) / 256)), dim3(256), 0, st, (uint32_t)N, acc_s, acc_h, c->b_sym32.p);
	};
	auto pairs = [&](bool bang) -> int {
		if (tiles.empty() || !P.W) {
			finish_tallies();
			return 0;
		}
		if (!bang && !mtiles.empty()) {
			// Without '!' the tallies are a contraction over {-1, 0, 1} channels: the matrix cores take it
			// (pileup_kernels.hip: pairs_mfma_kernel).  Window chunks: the chunk's rows of three planes in an XCD's L2,
			// ~16 wavefronts per CU in all (measured best at N = 256; indifferent at N = 1024), a multiple of 8 chunks
			// (dealt round-robin over the XCDs) of a whole number of steps (2 windows) each.
			const uint32_t row_bytes = 3u * P.Npad * 4u;
			const uint32_t l2_fit = std::max<uint32_t>(64, (3u << 20) / row_bytes);
			const uint32_t want_chunks = std::max<uint32_t>(1, ((uint32_t)c->n_cu * 16u) / (uint32_t)mtiles.size());
			uint32_t wchunk = std::max<uint32_t>(48, (P.W + want_chunks - 1) / want_chunks);
			wchunk = std::min(wchunk, l2_fit);
			const uint32_t groups = (P.W + 8u * wchunk - 1) / (8u * wchunk);
			wchunk = std::max<uint32_t>(2, (P.W + 8u * groups - 1) / (8u * groups));
			wchunk = (wchunk + 1u) & ~1u;
			if (c->opt_pairs_wchunk) wchunk = (c->opt_pairs_wchunk + 1u) & ~1u;
			wchunk = std::min(wchunk, pairs_mfma_max_wchunk() & ~1u);
			// (the kernel addresses a chunk's rows with 32-bit byte offsets, the loads it issues ahead included)
			wchunk = std::min<uint32_t>(wchunk, (uint32_t)((0xffffffffull / ((uint64_t)P.Npad * 4u)) & ~1ull) - 16u);
			// A wavefront flushes its 64 x 64 tallies with 64-bit atomics: with many chunks (N = 1024: 592 of the 264 windows
			// the L2 holds the rows of) that is a fifth of the kernel — so a wavefront takes up to four chunks of its XCD in
			// a row, as long as that leaves the chip some sixty wavefronts per CU to balance with.
			uint32_t cpw = 1;
			{
				const uint64_t nwc = (P.W + wchunk - 1) / wchunk, waves = nwc * mtiles.size();
				cpw = (uint32_t)std::min<uint64_t>(4, std::max<uint64_t>(1, waves / ((uint64_t)c->n_cu * 64u)));
#ifdef PHY_DEV_HOOKS
				if (const char *e = getenv("PHY_PAIRS_CPW")) cpw = (uint32_t)std::max(1, atoi(e)); // experiments
#endif
				// (the f32 accumulators are flushed once per cpw chunks and are exact below 2^24: the clamp comes last)
				cpw = std::min<uint32_t>(cpw, std::max<uint32_t>(1u, pairs_mfma_max_wchunk() / wchunk));
			}
			if (c->profile == 1 && !c->b_clk.p) { // (stat "clock:pairs_mfma_mhz")
				HIPOK(c, c->b_clk.ensure(2));
				HIPOK(c, hipMemsetAsync(c->b_clk.p, 0, 16, st));
				c->stats["clock:pairs_mfma_mhz"] = 0;
			}
			{
				KernelSpan s(c, "pileup_pairs_mfma");
				launch_pairs_mfma(P, c->b_tiles.p + tiles.size(), (uint32_t)mtiles.size(), wchunk, acc_s, acc_h, st, cpw, c->profile == 1 ? c->b_clk.p : nullptr);
			}
			HIPOK(c, hipGetLastError());
			finish_tallies();
			return 0;
		}
		// window chunks: small enough that a chunk's plane rows (3 or 5 planes x Npad x 4 B
		// per window) fit an XCD's 4 MiB L2 with room to spare, and small enough that
		// tiles x chunks fills the chip several times over; at least 64 windows
		uint32_t row_bytes = (bang ? 5u : 3u) * P.Npad * 4u;
		uint32_t l2_fit = std::max<uint32_t>(64, (3u << 20) / row_bytes);
		// (four rounds of the chip's n_cu x 32 wavefront slots: with one round — what n_cu x 32 gave at N = 256 — the
		// wavefronts all end together and the tail is a whole wavefront long; measured 1.30 -> 1.13 ms on C3)
		uint32_t want_chunks = std::max<uint32_t>(1, ((uint32_t)c->n_cu * 128u) / (uint32_t)tiles.size());
		uint32_t wchunk = std::max<uint32_t>(64, (P.W + want_chunks - 1) / want_chunks);
		wchunk = std::min(wchunk, l2_fit);
		{ // chunks are dealt round-robin over the 8 XCDs: a multiple of 8 of them keeps the XCDs level
			const uint32_t groups = (P.W + 8u * wchunk - 1) / (8u * wchunk);
			wchunk = std::max<uint32_t>(1, (P.W + 8u * groups - 1) / (8u * groups));
		}
		if (c->opt_pairs_wchunk) wchunk = c->opt_pairs_wchunk;
		{
			KernelSpan s(c, bang ? "pileup_pairs_bang" : "pileup_pairs");
			launch_pairs(P, bang, c->b_tiles.p, (uint32_t)tiles.size(), wchunk, acc_s, acc_h, st);
		}
		HIPOK(c, hipGetLastError());
		finish_tallies();
		return 0;
	};
	auto project = [&](bool five) -> int {
		KernelSpan s(c, five ? "pileup_project5" : "pileup_project");
		launch_project(P, five, query_src(c), dev_homs, c->b_hom_rng.p, c->b_first.p, c->b_flag.p, 0, P.Npad / project_genomes_per_tile(), st,
					   c->b_bang.p, c->bang_cap, c->proj_resident);
		return 0;
	};
	uint64_t *hs = c->h_mat.p;
	auto fetch = [&]() -> int { // the flag, and the result unless it stays on the device
		HIPOK(c, hipGetLastError());
		HIPOK(c, hipMemcpyAsync(flagp, c->b_flag.p, 28, hipMemcpyDeviceToHost, st));
		if (!dev_out && !to_home) HIPOK(c, hipMemcpyAsync(hs, c->b_sym32.p, 2 * N * N * 4, hipMemcpyDeviceToHost, st));
		return sync_stream(c);
	};
	// The matrix-core path (option "pairs_kernel" = 0) works on three planes whatever the genomes hold: the projection
	// lists the '!' it meets — a handful: contig joins inside homologies — and launch_bang_correct settles them after the
	// pair kernel.  (Only a list beyond its capacity — lists installed by a caller that overlap on the query — falls
	// back to the five planes below.)
	const bool sparse = !mtiles.empty() && !(projected && c->eager_five);
	const bool have_five = sparse ? false : (projected ? c->eager_five : c->pileup_five); // the planes this attempt works on
	bool bang = !sparse && c->pileup_five && have_five;
	if (!projected) {
		c->eager_valid = false; // the planes now hold this call's projection (another part, other planes), not phase A's
		if (project(have_five)) return 1;
	}
	double t1 = now_ms();
	do_correct = sparse;
	if (out_mode == 2 && sparse) {
		// A part's triangle on the matrix-core path: everything is queued — pair kernel, the '!' corrections, the packed
		// triangle with, behind it, what this part has to report (pack_triangle_kernel) — and the call returns without
		// a host round trip: the caller's collective goes straight behind it on the stream.
		queued_report = true;
		if (pairs(bang)) return 1;
		HIPOK(c, hipGetLastError());
		c->att_unchecked = false;
		c->stats["ms:compare_project_phase"] += t1 - t0;
		c->stats["ms:compare_pairs_phase"] += now_ms() - t1;
		return 0;
	}
	if (pairs(bang) || fetch()) return 1;
	const bool att_bad = c->att_unchecked && (flagp[1] || flagp[2] || flagp[6]);
	c->att_unchecked = false;
	if (att_bad)
		return c->fail(flagp[2]   ? "the lists gathered from the ranks overflowed their blocks' capacity (phylo_attach_blocks_device)"
					   : flagp[6] ? "a rank's phase A needs the host (a list with tied starts, or scratch that overflowed): repeat the pass with phylo_anchor + phylo_export_block_device"
								  : "a gathered list is not sorted by projected start, disjoint and inside the reference");
	uint32_t flag = *flagp;
	if (sparse) flag = (flag & 2u) ? 1u : 0u; // only a '!' list beyond its capacity sends this path to the five planes
	do_correct = false;
	if (flag && !bang) { // '!' among the projected positions, and the plain kernel ran: once more with all five planes
		c->stats["count:compare_repeated_with_five_planes"] += 1;
		HIPOK(c, zero_tallies());
		HIPOK(c, hipMemsetAsync(c->b_flag.p, 0, 4, st));
		bang = true;
		if (project(true) || pairs(true) || fetch()) return 1;
		flag = *flagp;
	}
	if (!sparse) c->pileup_five = flag != 0;
	if (out_mode == 2) { // (this path has looked at its flags itself: nothing to report but "a part")
		const uint32_t tail[TRI_TAIL] = {0, 0, 0, 1, 0, 0, 0, 0};
		HIPOK(c, hipMemcpyAsync((uint32_t *)subst + N * (N - 1), tail, sizeof tail, hipMemcpyHostToDevice, st));
		HIPOK(c, hipStreamSynchronize(st));
	}
	if (dev_out) {
		c->stats["ms:compare_project_phase"] += t1 - t0;
		c->stats["ms:compare_pairs_phase"] += now_ms() - t1;
		c->stats["pileup:bang"] = flag;
		return 0;
	}
	double t2 = now_ms();
	// out of the pinned buffer into the caller's matrices, widened (16 MB at N = 1024: worth several threads)
	double sites = to_home ? 0.0 : widen_result(c, (const uint32_t *)hs, subst, homologs);
	sites *= 0.5;
	c->stats["ms:compare_project_phase"] += t1 - t0;
	c->stats["ms:compare_pairs_phase"] += t2 - t1;
	c->stats["ms:compare_symmetrise"] += now_ms() - t2;
	c->stats["count:compare_sites"] += sites;
	c->stats["pileup:bang"] = flag;
	return 0;
}

int phylo_compare(phylo_ctx *c, size_t part, size_t nparts, uint64_t *subst, uint64_t *homologs)
{
	if (!c) return 1;
	if (!subst || !homologs) return c->fail("null output matrix");
	if (nparts == 0 || part >= nparts) return c->fail("bad part %zu of %zu", part, nparts);
	if (!c->have_ref) return c->fail("phylo_compare: no reference set");
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	size_t N = c->n;
	int rc;
	if (c->backend == 1) { // the segment backend adds into the matrices; the pileup one writes every cell
		std::fill(subst, subst + N * N, 0);
		std::fill(homologs, homologs + N * N, 0);
		rc = compare_segments(c, part, nparts, subst, homologs);
	} else {
		rc = compare_pileup(c, part, nparts, subst, homologs);
	}
	c->stats["ms:compare_total"] += now_ms() - t0;
	c->stats["n:compare_calls"] += 1;
	return rc;
}

int phylo_compare_device(phylo_ctx *c, size_t part, size_t nparts, uint64_t *dev_subst, uint64_t *dev_homologs)
{
	if (!c) return 1;
	if (!dev_subst || !dev_homologs) return c->fail("null output matrix");
	if (nparts == 0 || part >= nparts) return c->fail("bad part %zu of %zu", part, nparts);
	if (!c->have_ref) return c->fail("phylo_compare_device: no reference set");
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	int rc;
	if (c->backend == 1) { // the segment backend tallies on the host: copy its result over
		size_t N = c->n;
		std::vector<uint64_t> s(N * N, 0), h(N * N, 0);
		rc = compare_segments(c, part, nparts, s.data(), h.data());
		if (!rc) {
			HIPOK(c, hipMemcpy(dev_subst, s.data(), N * N * 8, hipMemcpyHostToDevice));
			HIPOK(c, hipMemcpy(dev_homologs, h.data(), N * N * 8, hipMemcpyHostToDevice));
		}
	} else {
		rc = compare_pileup(c, part, nparts, dev_subst, dev_homologs, 1);
	}
	c->stats["ms:compare_total"] += now_ms() - t0;
	c->stats["n:compare_calls"] += 1;
	return rc;
}

int phylo_compare_triangle_device(phylo_ctx *c, size_t part, size_t nparts, uint32_t *dev_tri)
{
	if (!c) return 1;
	if (!dev_tri) return c->fail("null output triangle");
	if (nparts == 0 || part >= nparts) return c->fail("bad part %zu of %zu", part, nparts);
	if (!c->have_ref) return c->fail("phylo_compare_triangle_device: no reference set");
	if (c->backend == 1) return c->fail("phylo_compare_triangle_device: the segment backend tallies on the host (use phylo_compare)");
	HIPOK(c, hipSetDevice(c->device));
	double t0 = now_ms();
	const int rc = compare_pileup(c, part, nparts, (uint64_t *)dev_tri, nullptr, 2);
	c->stats["ms:compare_total"] += now_ms() - t0;
	c->stats["n:compare_calls"] += 1;
	return rc;
}

size_t phylo_triangle_words(size_t n) { return n * (n - 1) + TRI_TAIL; }

// the caller's result matrix as the device sees it, when the caller keeps handing the same buffer over: registered
// (mapped) the second time it is seen, for as long as the context lives
static void *host_matrix_on_device(phylo_ctx *c, void *p, size_t bytes)
{
	// (a buffer of its own pages only: an allocation of a megabyte comes from mmap, while a small one shares its pages
	// with whatever else the heap holds — and the runtime then takes copies to those neighbours for copies into the
	// registered range)
	if (c->res.map && c->res.n == c->n) { // the library's own page-locked home of the result (phylo_result_matrices): nothing to register
		if (p == (void *)c->res.subst()) return (char *)c->res.dev + 4096;
		if (p == (void *)c->res.homologs()) return (char *)c->res.dev + 4096 + c->res.matrix_words() * 8;
	}
	if (!c->opt_result_zero_copy || bytes < ((size_t)1 << 20)) return nullptr;
	for (auto &r : c->host_regs)
		if (r.ptr == p && r.bytes == bytes) {
			if (r.dev) return r.dev;
			if (r.failed) return nullptr;
			void *d = nullptr;
			if (hipHostRegister(p, bytes, hipHostRegisterMapped) != hipSuccess || hipHostGetDevicePointer(&d, p, 0) != hipSuccess || !d) {
				(void)hipGetLastError();
				r.failed = true;
				return nullptr;
			}
			r.dev = d;
			return d;
		}
	if (c->host_regs.size() >= 8) { // callers that keep changing buffers: forget the oldest
		if (c->host_regs.front().dev) (void)hipHostUnregister(c->host_regs.front().ptr);
		c->host_regs.erase(c->host_regs.begin());
	}
	c->host_regs.push_back(phylo_ctx::HostReg{p, bytes, nullptr, false});
	return nullptr;
}

int phylo_triangle_to_matrices(phylo_ctx *c, const uint32_t *dev_tri, uint64_t *subst, uint64_t *homologs)
{
	if (!c) return 1;
	if (!dev_tri || !subst || !homologs) return c->fail("phylo_triangle_to_matrices: null argument");
	HIPOK(c, hipSetDevice(c->device));
	const size_t N = c->n;
	if (!N) return 0;
	HIPOK(c, c->h_mat.ensure(2 * N * N + 8));
	HIPOK(c, c->b_sym32.ensure(2 * N * N + 4));
	const double t0 = now_ms();
	uint32_t *tail = (uint32_t *)(c->h_mat.p + 2 * N * N); // the parts' flags (phylo_triangle_words), read with the result
	unsigned long long *h_sites = (unsigned long long *)(c->h_mat.p + 2 * N * N + 4);
	void *ds = nullptr, *dh = nullptr;
	if ((((uintptr_t)subst | (uintptr_t)homologs) & 15u) == 0) { // (the kernel stores 16 bytes at a time)
		ds = host_matrix_on_device(c, subst, N * N * 8);
		dh = host_matrix_on_device(c, homologs, N * N * 8);
	}
	double sites = 0;
	if (ds && dh) {
		// the caller reuses its matrices: the device writes them itself (triangle_to_host_kernel)
		HIPOK(c, c->b_flag.ensure(8));
		unsigned long long *d_sites = (unsigned long long *)(c->b_flag.p + 4);
		HIPOK(c, hipMemsetAsync(d_sites, 0, 8, c->stream));
		const uint64_t threads = (uint64_t)N * ((N + 1) / 2);
		hipLaunchKernelGGL(triangle_to_host_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->stream, (uint32_t)N, dev_tri,
						   (unsigned long long *)ds, (unsigned long long *)dh, d_sites);
		HIPOK(c, hipGetLastError());
		HIPOK(c, hipMemcpyAsync(tail, dev_tri + N * (N - 1), TRI_TAIL * 4, hipMemcpyDeviceToHost, c->stream));
		HIPOK(c, hipMemcpyAsync(h_sites, d_sites, 8, hipMemcpyDeviceToHost, c->stream));
		if (sync_stream(c)) return 1;
		sites = (double)*h_sites;
		c->stats["ms:triangle_zero_copy"] += now_ms() - t0;
	} else {
		hipLaunchKernelGGL(sym32_from_triangle_kernel, dim3((uint32_t)((N * N + 255) / 256)), dim3(256), 0, c->stream, (uint32_t)N, dev_tri, c->b_sym32.p);
		HIPOK(c, hipGetLastError());
		HIPOK(c, hipMemcpyAsync(c->h_mat.p, c->b_sym32.p, 2 * N * N * 4, hipMemcpyDeviceToHost, c->stream));
		HIPOK(c, hipMemcpyAsync(tail, dev_tri + N * (N - 1), TRI_TAIL * 4, hipMemcpyDeviceToHost, c->stream));
		if (sync_stream(c)) return 1;
		const double t1 = now_ms();
		sites = widen_result(c, (const uint32_t *)c->h_mat.p, subst, homologs);
		c->stats["ms:triangle_copy"] += t1 - t0;
		c->stats["ms:triangle_widen"] += now_ms() - t1;
	}
	c->stats["count:compare_sites"] += 0.5 * sites;
	if (settle_anchor(c)) return 1; // (a phase A queued by phylo_anchor_block_device: its statistics)
	if (tail[4]) return c->fail("a rank's phase A needs the host (a list with tied starts, or scratch that overflowed): repeat the pass with phylo_anchor + phylo_export_block_device");
	if (tail[2]) return c->fail("the lists gathered from the ranks overflowed their blocks' capacity (phylo_attach_blocks_device)");
	if (tail[1]) return c->fail("a gathered list is not sorted by projected start, disjoint and inside the reference");
	if (tail[0]) return c->fail("more '!' inside homologies than the genomes hold separators (lists installed by a caller that overlap on the query): compare with option pairs_kernel = 1");
	return 0;
}

// what phylo_anchor does with the flags it waits for, after a deferred call's stream has been synchronised by somebody
// else: 0 lists and ranges are in place, 1 error, 2 a list needs the host (the optimistic state is withdrawn)
static int anchor_finish(phylo_ctx *c)
{
	const size_t N = c->n;
	const uint32_t *hr = c->h_rng.p, *dmisc = hr + 3 * N + 1;
	c->anchor_pending = false;
	size_t flagged = 0;
	for (size_t j = 0; j < N; j++) flagged += hr[2 * N + 1 + j] != 0;
	if (dmisc[3] || flagged) {
		c->homs_staged = false;
		c->eager_valid = false;
		c->att_homs = nullptr;
		c->host_stale.clear();
		if (dmisc[3]) return c->fail("phase A scratch overflow (code %u: 1 chunk log, 2 bridge pool, 3 homology buffer)", dmisc[3]);
		return 2;
	}
	c->att_rng_on_device = false;
	if (c->att_begin.size() != N) {
		c->att_begin.assign(N, 0);
		c->att_count.assign(N, 0);
	}
	c->host_stale.assign(N, 0);
	for (size_t j = 0; j < N; j++) {
		c->att_begin[j] = hr[2 * j];
		c->att_count[j] = hr[2 * j + 1] - hr[2 * j];
		c->host_stale[j] = 1;
	}
	c->stats["ms:anchor_setup"] += c->pend_t1 - c->pend_t0;
	c->stats["ms:anchor_total"] += c->pend_t2 - c->pend_t0; // (the host's part: the device's time is in phase B's wait)
	c->stats["n:anchor_calls"] += 1;
	c->stats["n:anchor_calls_without_a_wait"] += 1;
	c->stats["count:query_bases"] += c->pend_total;
	c->stats["count:chunks"] += c->pend_nch;
	c->stats["count:filtered_homologies"] += (double)hr[2 * N];
	c->stats["count:pool_blocks_used"] += dmisc[2];
	c->stats["count:overrun_runs"] += dmisc[5];
	c->stats["count:overrun_bytes_compared"] += dmisc[6];
	c->stats["anchor:chunk"] = c->pend_C;
	return 0;
}

// phylo_anchor(all genomes) + phylo_compare_all as the one call they are in the reference (process(), process.cxx:408-556):
// phase B is queued behind phase A without the host reading phase A's flags in between — one wait instead of two — and
// the flags are read with the result.  A list that needs the host after all (two homologies with the same projected
// start: the reference's order of such ties is libstdc++'s) sends the call the long way round.
int phylo_anchor_compare(phylo_ctx *c, uint64_t *subst, uint64_t *homologs)
{
	if (!c) return 1;
	int rc = anchor_impl(c, 0, c->n, 1);
	if (rc) return rc;
	if (!c->anchor_pending) return phylo_compare(c, 0, 1, subst, homologs);
	rc = phylo_compare(c, 0, 1, subst, homologs); // (synchronises the stream whichever way it ends)
	if (c->anchor_pending && hipStreamSynchronize(c->stream) != hipSuccess) return c->fail("phylo_anchor_compare: the device failed");
	const int f = anchor_finish(c);
	if (f == 1) return 1;
	if (f == 0) return rc;
	c->stats["count:anchor_compare_calls_repeated"] += 1;
	rc = anchor_impl(c, 0, c->n, 0);
	if (rc) return rc;
	return phylo_compare(c, 0, 1, subst, homologs);
}

int phylo_compare_all(phylo_ctx *c, uint64_t *subst, uint64_t *homologs)
{
	return phylo_compare(c, 0, 1, subst, homologs);
}

int phylo_process(phylo_ctx *c, size_t ref_idx, int flags, uint64_t *subst, uint64_t *homologs)
{
	if (!c) return 1;
	int rc = phylo_set_reference(c, ref_idx, nullptr, 0);
	if (rc) return rc;
	if (!(flags & PHYLO_COMPLETE_DELETION)) return phylo_anchor_compare(c, subst, homologs);
	rc = phylo_anchor(c, 0, c->n);
	if (rc) return rc;
	rc = phylo_complete_delete(c);
	if (rc) return rc;
	return phylo_compare_all(c, subst, homologs);
}

} // extern "C"

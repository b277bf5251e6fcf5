// phylonium_amd_cli.cpp — host driver over the C ABI with phylonium's command
// line: FASTA files in, PHYLIP distance matrix out.
//
// Mirrors /root/reference/src/phylonium.cxx:89-299 (option parsing, reference
// choice, one or two passes), src/io.cxx:36-233 (genome names, FASTA → joined
// sequence, warnings, matrix output) and the flag-gated consumers of the
// tallies in src/process.cxx:467-513 (-p) and src/evo_model.cxx:136-147
// (bootstrap).  The hot path itself — process() — is libphylonium_amd.so.
// Written from the behaviour, not from the reference's text.
#include <algorithm>
#include <array>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <getopt.h>
#include <iostream>
#include <limits>
#include <numeric>
#include <random>
#include <string>
#include <strings.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/phylonium_amd.h"
#include "fasta_reader.hpp"
#include "fmt_e4.hpp"

namespace {

const char *PROG = "phylonium-amd";
int RETURN_CODE = EXIT_SUCCESS;

enum Flags { F_VERBOSE = 1, F_EXTRA_VERBOSE = 2, F_COMPLETE_DELETION = 4, F_PROGRESS = 8, F_POSITIONS = 16, F_ANI = 32, F_RAW = 64 };

using phyfasta::Genome;
using phyfasta::PackedGenome;
using phyfasta::read_genomes;

// Length of every genome (nucleotides + separators).  With the packed ingest (the default) the bytes of a
// genome exist on the host only where something asks for them: the reference (suffix array), -p.
std::vector<size_t> GLEN;

[[noreturn]] void die(const std::string &msg)
{
	fprintf(stderr, "%s: %s\n", PROG, msg.c_str());
	exit(1);
}
void soft_err(const std::string &msg)
{
	RETURN_CODE |= EXIT_FAILURE;
	fprintf(stderr, "%s: %s\n", PROG, msg.c_str());
}

double now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Tally {
	uint64_t subst = 0, homologs = 0;
};
using Matrix = std::vector<Tally>;

double dist_of(const Tally &t, int flags, bool zero_on_error = false)
{
	int kind = (flags & F_RAW) ? 1 : (flags & F_ANI) ? 2 : 0;
	return phylo_estimate(kind, t.subst, t.homologs, zero_on_error);
}

// just_print, io.cxx:141-163: "%.4e" ("%.4g" for ANI: std::dec leaves the float field at its default) is what
// the stream formats print.  Rows are formatted on the host threads and written in order.
size_t PRINT_THREADS = 1;
void print_phylip(const std::vector<Genome> &q, const std::vector<double> &d, int flags)
{
	size_t N = q.size();
	const bool ani = (flags & F_ANI) != 0;
	const size_t nt = std::max<size_t>(1, std::min(PRINT_THREADS, N / 16 + 1));
	std::vector<std::string> part(nt);
	auto work = [&](size_t t) {
		std::string &o = part[t];
		const size_t i0 = N * t / nt, i1 = N * (t + 1) / nt;
		o.reserve((i1 - i0) * (N * 12 + 32));
		char buf[64];
		for (size_t i = i0; i < i1; i++) {
			o += q[i].name;
			for (size_t j = 0; j < N; j++) {
				const double v = i == j ? 0.0 : d[i * N + j];
				buf[0] = buf[1] = ' ';
				o.append(buf, 2 + (ani ? (size_t)snprintf(buf + 2, sizeof buf - 2, "%.4g", v) : phyfmt::e4(buf + 2, v)));
			}
			o += '\n';
		}
	};
	std::vector<std::thread> pool;
	for (size_t t = 1; t < nt; t++) pool.emplace_back(work, t);
	work(0);
	for (auto &t : pool) t.join();
	std::cout << N << std::endl;
	for (auto &o : part) std::cout.write(o.data(), (std::streamsize)o.size());
	std::cout.flush();
}

// print_matrix, io.cxx:165-233
void print_matrix(const std::vector<Genome> &q, const Matrix &m, int flags, unsigned long bootstrap, size_t ref_idx,
				  std::mt19937 &prng)
{
	size_t N = q.size();
	std::vector<double> d(N * N);
	// The distances (a logarithm each) and the warnings (io.cxx:106-139: they come first, row by row) are worked out on
	// the host threads — a million pairs at N = 1024 — and the warnings then leave in the reference's order.
	std::vector<std::vector<std::string>> warn(N);
	{
		const size_t nt = std::max<size_t>(1, std::min(PRINT_THREADS, N / 16 + 1));
		auto work = [&](size_t t) {
			for (size_t i = t; i < N; i += nt) // (row i checks i pairs: dealt round-robin the threads get the same share)
				for (size_t j = 0; j < N; j++) {
					const double v = d[i * N + j] = dist_of(m[i * N + j], flags);
					if (j >= i) continue;
					char buf[1024];
					if (std::isnan(v)) {
						snprintf(buf, sizeof buf,
								 "For the two sequences '%s' and '%s' the distance computation failed and is reported as nan.",
								 q[i].name.c_str(), q[j].name.c_str());
						warn[i].push_back(buf);
					} else {
						double c1 = (double)m[i * N + j].homologs / GLEN[i];
						double c2 = (double)m[i * N + j].homologs / GLEN[j];
						if (c1 < 0.2 || c2 < 0.2) {
							snprintf(buf, sizeof buf,
									 "For the two sequences '%s' and '%s' less than 20%% homology were found (%f and %f, "
									 "respectively).",
									 q[i].name.c_str(), q[j].name.c_str(), c1, c2);
							warn[i].push_back(buf);
						}
					}
				}
		};
		std::vector<std::thread> pool;
		for (size_t t = 1; t < nt; t++) pool.emplace_back(work, t);
		work(0);
		for (auto &t : pool) t.join();
	}
	for (size_t i = 0; i < N; i++)
		for (const std::string &w : warn[i]) soft_err(w);
	print_phylip(q, d, flags);
	for (unsigned long k = 0; k < bootstrap; k++) { // evo_model::bootstrap, evo_model.cxx:136-147
		std::vector<double> b(N * N);
		for (size_t t = 0; t < N * N; t++) {
			Tally r = m[t];
			double rate = r.subst / (double)r.homologs;
			std::binomial_distribution<> dist((int)r.homologs, rate);
			r.subst = (uint64_t)dist(prng);
			b[t] = dist_of(r, flags);
		}
		print_phylip(q, b, flags);
	}
	if (flags & F_VERBOSE) {
		double sum = 0;
		size_t counter = 0;
		for (size_t i = 0; i < N; i++)
			for (size_t j = 0; j < i; j++) {
				if (std::isnan(d[i * N + j])) continue;
				sum += (double)m[i * N + j].homologs / GLEN[i] + (double)m[i * N + j].homologs / GLEN[j];
				counter += 2;
			}
		size_t aligned = 0, total = 0;
		for (size_t i = 0; i < N; i++) {
			if (i == ref_idx) continue;
			aligned += m[ref_idx * N + i].homologs;
			total += GLEN[i];
		}
		std::cerr << "avg coverage:\t" << sum / counter << std::endl;
		std::cerr << "alignment:\t" << aligned << "\t" << total << "\t" << aligned / (double)total << std::endl;
	}
}

// pick_first_pass, phylonium.cxx:360-382 — median length by nth_element, then
// the first genome equal to it
size_t pick_first_pass(const std::vector<Genome> &q, const std::vector<PackedGenome> &pk, int flags)
{
	std::vector<size_t> idx(q.size());
	std::iota(idx.begin(), idx.end(), 0);
	std::nth_element(idx.begin(), idx.begin() + idx.size() / 2, idx.end(), [&](size_t a, size_t b) { return GLEN[a] < GLEN[b]; });
	size_t chosen = idx[idx.size() / 2];
	size_t ref = chosen;
	auto same = [&](size_t a, size_t b) {
		if (q[a].name != q[b].name || GLEN[a] != GLEN[b]) return false;
		if (pk.empty()) return q[a].nucl == q[b].nucl;
		return pk[a].bad == pk[b].bad && memcmp(pk[a].q2, pk[b].q2, (GLEN[a] + 15) / 16 * sizeof(uint32_t)) == 0;
	};
	for (size_t i = 0; i < q.size(); i++)
		if (same(i, chosen)) {
			ref = i;
			break;
		}
	if (flags & F_VERBOSE) std::cerr << "chosen reference: " << q[ref].name << std::endl;
	return ref;
}

// pick_second_pass, phylonium.cxx:317-344 — the row with the smallest sum of JC distances
size_t pick_second_pass(size_t N, const Matrix &m)
{
	double best = std::numeric_limits<double>::max();
	size_t arg = 0;
	for (size_t i = 0; i < N; i++) {
		double sum = 0.0;
		for (size_t j = 0; j < N; j++) sum += phylo_estimate(0, m[i * N + j].subst, m[i * N + j].homologs, 1);
		if (sum < best) {
			best = sum;
			arg = i;
		}
	}
	return arg;
}

struct Run {
	phylo_ctx *ctx = nullptr;      // the context (rank 0's, with several GPUs)
	phylo_group *grp = nullptr;    // --gpus N: one context and one host thread per GPU (csrc/group.hip)
	std::vector<Genome> *q = nullptr;
	int flags = 0;
	std::string refpos_file;
};

void ok(Run &r, int rc)
{
	if (rc) die(phylo_last_error(r.ctx));
}
void gok(Run &r, int rc)
{
	if (rc) die(phylo_group_last_error(r.grp));
}

// -p FILE: reference positions of the core alignment with the segregating
// sites of every block (process.cxx:471-513, 665-723)
void write_positions(Run &r, size_t ref_idx)
{
	auto &q = *r.q;
	size_t N = q.size();
	std::vector<const phylo_homology *> H(N);
	std::vector<size_t> n(N);
	for (size_t g = 0; g < N; g++) ok(r, phylo_get_homologies(r.ctx, g, &H[g], &n[g]));
	for (size_t g = 0; g < N; g++) // packed ingest: the bytes come back from the device
		if (q[g].nucl.size() != GLEN[g]) {
			q[g].nucl.resize(GLEN[g]);
			ok(r, phylo_get_genome(r.ctx, g, q[g].nucl.data()));
		}
	std::ofstream out(r.refpos_file);
	size_t counter = 1;
	const std::string &subject = q[ref_idx].nucl;
	for (size_t i = 0; i < n[0]; i++) {
		const phylo_homology &h0 = H[0][i];
		std::vector<char> seg(h0.length, 0);
		for (size_t m = 0; m < N; m++) {
			const phylo_homology &hm = H[m][i];
			// after complete deletion all blocks share start and length
			uint64_t len = h0.length;
			const char *a = q[0].nucl.data() + h0.index_query, *b = q[m].nucl.data() + hm.index_query;
			std::vector<char> f(len, 0);
			if (h0.direction == hm.direction) {
				for (uint64_t t = 0; t < len; t++) f[t] = a[t] != b[t];
				if (h0.direction == 1) std::reverse(f.begin(), f.end());
			} else if (hm.direction == 1) {
				for (uint64_t t = 0; t < len; t++) f[t] = (((a[t] ^ b[len - t - 1]) & 6) != 4);
			} else {
				for (uint64_t t = 0; t < len; t++) f[t] = (((b[t] ^ a[len - t - 1]) & 6) != 4);
			}
			for (uint64_t t = 0; t < len; t++) seg[t] |= f[t];
		}
		std::vector<size_t> pos;
		for (size_t t = 0; t < seg.size(); t++)
			if (seg[t]) pos.push_back(t);
		uint64_t start = h0.index_reference_projected, end = start + h0.length;
		out << ">part" << counter++ << "\t(" << (start + 1) << ".." << (end + 1) << ")  " << pos.size();
		for (size_t p : pos) out << "  " << (p + 1);
		out << std::endl;
		out << subject.substr(start, end - start) << std::endl;
	}
}

// process(), process.cxx:408-556, over the C ABI
Matrix process(Run &r, size_t ref_idx, const int64_t *sa = nullptr)
{
	auto &q = *r.q;
	size_t N = q.size();
	if (r.grp) gok(r, phylo_group_set_reference(r.grp, ref_idx, sa, 0));
	else ok(r, phylo_set_reference(r.ctx, ref_idx, sa, 0));
	if (r.flags & F_VERBOSE) std::cerr << "ref: " << q[ref_idx].name << std::endl;
	if ((r.flags & F_VERBOSE) && phylo_reference_cache_quirk(r.ctx))
		std::cerr << "note: phylonium's 6-mer cache over-reports some matches on this reference (src/esa.cxx:174-199); "
					 "its answers are reproduced here, so the distances are phylonium's, not those of the true longest matches" << std::endl;
	// phase A: every GPU anchors its block of the queries, then every GPU holds all lists (rank 0's context
	// hands out any of them)
	std::vector<uint64_t> s(N * N), h(N * N);
	if (!(r.flags & (F_COMPLETE_DELETION | F_POSITIONS))) {
		// nothing between the phases: both as one call — the host waits once; the group repeats a pass whose lists
		// outgrew the planned exchange blocks
		if (r.grp) gok(r, phylo_group_process(r.grp, s.data(), h.data()));
		else ok(r, phylo_anchor_compare(r.ctx, s.data(), h.data()));
		Matrix m(N * N);
		for (size_t k = 0; k < N * N; k++) m[k] = Tally{s[k], h[k]};
		return m;
	}
	if (r.grp) gok(r, phylo_group_anchor(r.grp));
	else ok(r, phylo_anchor(r.ctx, 0, N));
	if (r.flags & F_COMPLETE_DELETION) ok(r, phylo_complete_delete(r.ctx));
	if (r.flags & F_POSITIONS) write_positions(r, ref_idx);
	// phase B: by range of reference windows over the GPUs — or, after the N-way intersection of the lists on the
	// host (complete deletion, src/process.cxx:467-469), on rank 0's GPU, which holds the result of it
	if (r.grp && !(r.flags & F_COMPLETE_DELETION)) gok(r, phylo_group_compare(r.grp, s.data(), h.data()));
	else ok(r, phylo_compare_all(r.ctx, s.data(), h.data()));
	Matrix m(N * N);
	for (size_t k = 0; k < N * N; k++) m[k] = Tally{s[k], h[k]};
	return m;
}

// --bench-steps K: K more passes of process() against the reference in place (phase A + phase B, the index and the genomes
// resident), timed by this host's clock; one JSON line on stderr with the mean step, every step, and per rank the host-side
// milliseconds the group keeps (queued: start of the pass to the last call returning; step: to the rank's rows delivered).
// The result of every pass must be the matrix already printed.
void bench_passes(Run &r, size_t K, const Matrix &want)
{
	const size_t N = r.q->size(), W = r.grp ? phylo_group_size(r.grp) : 1;
	std::vector<uint64_t> s, h;
	uint64_t *ps = nullptr, *ph = nullptr;
	const bool in_place = r.grp && W > 1 && phylo_group_result_matrices(r.grp, &ps, &ph) == 0; // the group's own home of the result
	if (!in_place) {
		s.resize(N * N);
		h.resize(N * N);
		ps = s.data();
		ph = h.data();
	}
	auto pass = [&]() {
		if (r.grp) gok(r, phylo_group_process(r.grp, ps, ph));
		else ok(r, phylo_anchor_compare(r.ctx, ps, ph));
	};
	for (int w = 0; w < 2; w++) pass();
	std::vector<double> ms(K), q_sum(W, 0.0), st_sum(W, 0.0);
	const double t0 = now_s();
	for (size_t k = 0; k < K; k++) {
		const double a = now_s();
		pass();
		ms[k] = (now_s() - a) * 1e3;
		for (size_t rk = 0; r.grp && rk < W; rk++) {
			double v = 0;
			if (!phylo_group_get_stat(r.grp, rk, "group:ms_queued", &v)) q_sum[rk] += v;
			if (!phylo_group_get_stat(r.grp, rk, "group:ms_step", &v)) st_sum[rk] += v;
		}
	}
	const double mean = (now_s() - t0) * 1e3 / (double)K;
	size_t diff = 0;
	for (size_t k = 0; k < N * N; k++) diff += ps[k] != want[k].subst || ph[k] != want[k].homologs;
	double bases = 0;
	for (size_t l : GLEN) bases += (double)l;
	std::vector<double> sorted = ms;
	std::sort(sorted.begin(), sorted.end());
	double rep = 0, sh = 0;
	if (r.grp) {
		phylo_group_get_stat(r.grp, 0, "group:passes_repeated", &rep);
		phylo_group_get_stat(r.grp, 0, "group:shared_result", &sh);
	}
	fprintf(stderr, "bench-steps: {\"steps\": %zu, \"ranks\": %zu, \"backend\": \"%s\", \"rccl_ranks\": %zu, \"ms_per_step\": %.4f, \"median_ms\": %.4f, \"min_ms\": %.4f, "
					"\"Gbp_per_s\": %.2f, \"result_in_place\": %s, \"shared_result\": %s, \"passes_repeated\": %.0f, \"identical_to_printed_matrix\": %s, \"per_rank_ms\": [",
			K, W, r.grp ? phylo_group_backend(r.grp) : "one context", r.grp ? phylo_group_rccl_ranks(r.grp) : (size_t)0, mean, sorted[K / 2], sorted[0],
			bases / (mean * 1e-3) / 1e9, in_place ? "true" : "false", sh != 0 ? "true" : "false", rep, diff == 0 ? "true" : "false");
	for (size_t rk = 0; rk < W; rk++)
		fprintf(stderr, "%s{\"rank\": %zu, \"queued\": %.4f, \"step\": %.4f}", rk ? ", " : "", rk, r.grp ? q_sum[rk] / (double)K : 0.0, r.grp ? st_sum[rk] / (double)K : mean);
	fprintf(stderr, "]}\n");
	if (diff) RETURN_CODE = 3;
}

// --verify-ranks: the matrix of an N-GPU run against two routes that share nothing with the exchange between the
// ranks.  (1) The reference's row from rank 0's lists through seam B0 (phylo_seqcmp_batch: seqcmp / revseqcmp over the
// resident genomes, libs/seqcmp.h:14, libs/revseqcmp.h:25, summed over each query's list).  (2) A sub-matrix of up to 32
// genomes — the reference, the first genome of every rank's block, the rest evenly spread — against a fresh one-context
// run of those genomes alone (a pair's tallies depend on the reference and the two genomes only, process.cxx:524-529):
// lists damaged or mixed up on their way between the ranks change it.  One JSON line on stderr; false = a check failed.
bool verify_group_result(Run &r, const std::vector<PackedGenome> &pk, const std::vector<const uint32_t *> &pk_q2, size_t ref_idx, const Matrix &m, int device)
{
	const size_t N = GLEN.size(), W = phylo_group_size(r.grp);
	std::vector<size_t> row_bad, rows;
	for (size_t k = 0; k <= W + 3; k++) {
		const size_t j = k < W ? phylo_group_rank_begin(r.grp, k) : k == W ? 1 : k == W + 1 ? N / 3 : k == W + 2 ? N / 2 : N - 1;
		if (j < N && j != ref_idx && std::find(rows.begin(), rows.end(), j) == rows.end() && rows.size() < 12) rows.push_back(j);
	}
	for (size_t j : rows) {
		const phylo_homology *h = nullptr;
		size_t n = 0;
		ok(r, phylo_get_homologies(r.ctx, j, &h, &n));
		std::vector<uint32_t> ga(n, (uint32_t)ref_idx), gb(n, (uint32_t)j);
		std::vector<uint64_t> oa(n), ob(n), ln(n), out(n);
		std::vector<uint8_t> rev(n);
		uint64_t hsum = 0, ssum = 0;
		for (size_t t = 0; t < n; t++) {
			oa[t] = h[t].index_reference_projected, ob[t] = h[t].index_query, ln[t] = h[t].length, rev[t] = h[t].direction != 0;
			hsum += h[t].length;
		}
		if (n) ok(r, phylo_seqcmp_batch(r.ctx, n, ga.data(), oa.data(), gb.data(), ob.data(), ln.data(), rev.data(), out.data()));
		for (uint64_t v : out) ssum += v;
		if (ssum != m[ref_idx * N + j].subst || hsum != m[ref_idx * N + j].homologs) row_bad.push_back(j);
	}
	std::vector<size_t> idx = {ref_idx};
	const size_t want = std::min<size_t>(N, 32);
	for (size_t k = 0; k < W; k++) idx.push_back(std::min(N - 1, phylo_group_rank_begin(r.grp, k)));
	for (size_t t = 0; t < want; t++) idx.push_back(std::min(N - 1, t * N / want));
	std::sort(idx.begin(), idx.end());
	idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
	while (idx.size() > std::max<size_t>(want, W + 1)) { // (never the reference)
		const size_t drop = idx.back() == ref_idx ? idx.size() - 2 : idx.size() - 1;
		idx.erase(idx.begin() + drop);
	}
	const size_t M = idx.size(), sub_ref = std::find(idx.begin(), idx.end(), ref_idx) - idx.begin();
	std::vector<const uint32_t *> q2(M), bad(M);
	std::vector<size_t> len(M), nbad(M);
	for (size_t t = 0; t < M; t++) q2[t] = pk_q2[idx[t]], bad[t] = pk[idx[t]].bad.data(), len[t] = pk[idx[t]].len, nbad[t] = pk[idx[t]].bad.size();
	std::vector<uint64_t> s2(M * M), h2(M * M);
	phylo_ctx *c2 = nullptr;
	std::string one_err;
	if (phylo_ctx_create(&c2, device)) one_err = phylo_last_error(nullptr);
	else if (phylo_set_genomes_packed(c2, M, q2.data(), len.data(), bad.data(), nbad.data()) || phylo_process(c2, sub_ref, 0, s2.data(), h2.data()))
		one_err = phylo_last_error(c2);
	if (c2) phylo_ctx_destroy(c2);
	size_t diff = 0;
	std::string first;
	for (size_t a = 0; a < M && one_err.empty(); a++)
		for (size_t b = 0; b < M; b++)
			if (s2[a * M + b] != m[idx[a] * N + idx[b]].subst || h2[a * M + b] != m[idx[a] * N + idx[b]].homologs) {
				if (diff < 8) first += (first.empty() ? "" : ", ") + ("[" + std::to_string(idx[a]) + ", " + std::to_string(idx[b]) + "]");
				diff++;
			}
	const bool pass = row_bad.empty() && one_err.empty() && diff == 0;
	std::string rb;
	for (size_t j : row_bad) rb += (rb.empty() ? "" : ", ") + std::to_string(j);
	fprintf(stderr, "verify-ranks: {\"ok\": %s, \"n_ranks\": %zu, \"backend\": \"%s\", \"reference_row\": {\"genomes\": %zu, \"mismatching\": [%s]}, "
					"\"submatrix\": {\"genomes\": %zu, \"identical\": %s, \"cells_differing\": %zu, \"first_mismatching_pairs\": [%s]%s%s%s}, \"ranks\": [",
			pass ? "true" : "false", W, phylo_group_backend(r.grp), rows.size(), rb.c_str(), M, diff == 0 && one_err.empty() ? "true" : "false", diff, first.c_str(),
			one_err.empty() ? "" : ", \"error\": \"", one_err.c_str(), one_err.empty() ? "" : "\"");
	for (size_t k = 0; k < W; k++) {
		double a = 0, e = 0, c = 0, d = 0;
		phylo_group_get_stat(r.grp, k, "group:ms_anchor", &a);
		phylo_group_get_stat(r.grp, k, "group:ms_exchange", &e);
		phylo_group_get_stat(r.grp, k, "group:ms_compare", &c);
		phylo_group_get_stat(r.grp, k, "group:ms_reduce", &d);
		fprintf(stderr, "%s{\"rank\": %zu, \"first_genome\": %zu, \"ms_anchor\": %.3f, \"ms_exchange\": %.3f, \"ms_compare\": %.3f, \"ms_reduce\": %.3f}", k ? ", " : "", k,
				phylo_group_rank_begin(r.grp, k), a, e, c, d);
	}
	fprintf(stderr, "]}\n");
	return pass;
}

[[noreturn]] void usage(int status)
{
	const char str[] = {
		"Usage: phylonium-amd [OPTIONS] FILES...\n"
		"\tFILES... can be any sequence of FASTA files, each file representing one genome.\n\n"
		"Options:\n"
		"  -2, --2pass          Enable two-pass algorithm\n"
		"  -b, --bootstrap=N    Print additional bootstrap matrices\n"
		"  --complete-deletion  Delete the whole aligned column in case of gaps\n"
		"  -p FILE              Print reference positions to FILE (implies complete deletion)\n"
		"    --progress=WHEN    Accepted for compatibility; no progress bar is drawn\n"
		"  -r FILE              Set the reference genome\n"
		"  -t, --threads=N      Host threads (FASTA reading, per-genome sort/filter step)\n"
		"      --sa=WHERE       device (default) or host: who sorts the reference's suffixes\n"
		"      --ingest=HOW     packed (default: 2-bit codes made while reading, a quarter of the\n"
		"                       bytes uploaded) or bytes\n"
		"      --timing         Print where the wall-clock went to stderr\n"
		"      --verify-ranks   With --gpus N: check the result by two other routes (the reference's\n"
		"                       row through the seqcmp/revseqcmp kernels; a sub-matrix of up to 32\n"
		"                       genomes against a one-GPU run of those genomes), print the verdict and\n"
		"                       every rank's timings as one JSON line on stderr; exit status 3 on failure\n"
		"  -d, --device=N       GPU ordinal (default 0; with --gpus: the first of them)\n"
		"      --teardown       Release the device context(s) before exiting (default: exit as soon\n"
		"                       as the output is written)\n"
		"      --bench-steps=K  After the matrix is out: K more passes against the same reference, timed; one\n"
		"                       JSON line on stderr (mean step, per-rank host times); exit status 3 if a pass's\n"
		"                       result differs from the matrix printed\n"
		"      --gpus=N         Shard the queries (phase A) and the reference's windows (phase B) over N\n"
		"                       GPUs: one host thread and one context per GPU, RCCL between them\n"
		"                       (ranks share GPUs when the machine has fewer)\n"
		"  -v, --verbose        Print additional information\n"
		"      --distance=OPT   Choose between raw, jc corrected and ANI\n"
		"  -h, --help           Display this help and exit\n"
		"      --version        Output version information\n"};
	fprintf(status == EXIT_SUCCESS ? stdout : stderr, "%s", str);
	exit(status);
}

} // namespace

int main(int argc, char *argv[])
{
	// the bootstrap's engine (src/phylonium.cxx:78-91 seeds it from std::random_device); PHYLONIUM_AMD_SEED
	// fixes the seed so that tests can reproduce the resampled matrices
	std::random_device rd;
	const char *seed_env = getenv("PHYLONIUM_AMD_SEED");
	std::mt19937 prng(seed_env ? (std::mt19937::result_type)strtoul(seed_env, nullptr, 10) : rd());
	int version_flag = 0, timing = 0, verify_ranks = 0, teardown = 0, flags = 0, device = 0, gpus = 0;
	bool packed_ingest = true, host_sa = false;
	long threads = 0, bench_steps = 0;
	bool two_pass = false;
	unsigned long bootstrap = 0;
	std::string reference_name, refpos_file;

	static struct option long_options[] = {{"2pass", no_argument, NULL, '2'},
										   {"bootstrap", required_argument, NULL, 'b'},
										   {"complete-deletion", no_argument, NULL, 0},
										   {"distance", required_argument, NULL, 0},
										   {"help", no_argument, NULL, 'h'},
										   {"progress", optional_argument, NULL, 0},
										   {"threads", required_argument, NULL, 't'},
										   {"device", required_argument, NULL, 'd'},
										   {"verbose", no_argument, NULL, 'v'},
										   {"version", no_argument, &version_flag, 1},
										   {"timing", no_argument, &timing, 1},
										   {"teardown", no_argument, &teardown, 1},
										   {"verify-ranks", no_argument, &verify_ranks, 1},
										   {"ingest", required_argument, NULL, 0},
										   {"sa", required_argument, NULL, 0},
										   {"gpus", required_argument, NULL, 0},
										   {"bench-steps", required_argument, NULL, 0},
										   {0, 0, 0, 0}};
	for (;;) {
		int option_index = 0;
		int c = getopt_long(argc, argv, "2b:d:hp:r:t:v", long_options, &option_index);
		if (c == -1) break;
		switch (c) {
			case 0: {
				std::string name = long_options[option_index].name;
				if (name == "complete-deletion") flags |= F_COMPLETE_DELETION;
				if (name == "sa") {
					if (strcasecmp(optarg, "host") == 0) host_sa = true;
					else if (strcasecmp(optarg, "device") != 0) usage(EXIT_FAILURE);
				}
				if (name == "bench-steps") {
					bench_steps = atol(optarg);
					if (bench_steps < 1 || bench_steps > 100000) usage(EXIT_FAILURE);
				} else if (name == "gpus") {
					gpus = atoi(optarg);
					if (gpus < 1 || gpus > 64) usage(EXIT_FAILURE);
				}
				if (name == "ingest") {
					if (strcasecmp(optarg, "bytes") == 0) packed_ingest = false;
					else if (strcasecmp(optarg, "packed") != 0) usage(EXIT_FAILURE);
				}
				if (name == "distance") {
					if (strcasecmp(optarg, "raw") == 0) flags |= F_RAW;
					else if (strcasecmp(optarg, "jc") == 0) {
					} else if (strcasecmp(optarg, "ani") == 0) flags |= F_ANI;
					else
						soft_err(std::string("ignoring argument for --distance '") + optarg +
								 "' expected one of 'raw', 'jc', or 'ani'");
				}
				break;
			}
			case '2': two_pass = true; break;
			case 'b': {
				errno = 0;
				char *end;
				unsigned long b = strtoul(optarg, &end, 10);
				if (errno || end == optarg || *end != '\0' || b == 0) {
					soft_err(std::string("Expected a positive number for -b argument, but '") + optarg +
							 "' was given. Ignoring -b argument.");
					break;
				}
				bootstrap = b - 1;
				break;
			}
			case 'd': device = atoi(optarg); break;
			case 'h': usage(EXIT_SUCCESS);
			case 'p':
				flags |= F_POSITIONS | F_COMPLETE_DELETION;
				refpos_file = optarg;
				break;
			case 'r': reference_name = optarg; break;
			case 't': threads = strtol(optarg, nullptr, 10); break;
			case 'v': flags |= (flags & F_VERBOSE) ? F_EXTRA_VERBOSE : F_VERBOSE; break;
			default: usage(EXIT_FAILURE);
		}
	}
	if (flags & F_POSITIONS) {
		std::ifstream probe(refpos_file);
		if (probe.good()) die("output file '" + refpos_file + "' already exists");
	}
	if (version_flag) {
		printf("%s\n", phylo_version());
		return 0;
	}
	std::vector<std::string> files(argv + optind, argv + argc);
	if (!reference_name.empty()) { // cleanup_names, phylonium.cxx:384-391
		files.push_back(reference_name);
		std::sort(files.begin(), files.end());
		files.erase(std::unique(files.begin(), files.end()), files.end());
	}
	if (files.size() < 2) usage(EXIT_FAILURE);

	// --timing: where the wall-clock goes (stderr), FASTA in to PHYLIP out
	double t_start = now_s(), t_read, t_ctx, t_upload, t_done;
	// The device context (HIP start-up, ~0.3 s) is created while the files are being read.
	Run r;
	std::string ctx_error;
	if (gpus > 0) packed_ingest = true; // the ranks exchange the packed form
	std::thread ctx_thread([&] {
		if (gpus > 0) {
			// rank k on GPU (device + k) modulo the machine's GPUs: with fewer GPUs than ranks they share them
			int count = 0;
			std::vector<int> devs;
			if (phylo_host_device_count(&count) || count < 1) {
				ctx_error = "no usable HIP device";
				return;
			}
			for (int k = 0; k < gpus; k++) devs.push_back((device + k) % count);
			if (phylo_group_create(&r.grp, (size_t)gpus, devs.data())) ctx_error = phylo_group_last_error(nullptr);
			else r.ctx = phylo_group_ctx(r.grp, 0);
		} else if (phylo_ctx_create(&r.ctx, device)) {
			ctx_error = phylo_last_error(nullptr);
		}
	});
	size_t read_threads = threads > 0 ? (size_t)threads : std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
	PRINT_THREADS = read_threads;
	std::string read_error;
	std::vector<Genome> q;
	std::vector<PackedGenome> pk;
	uint32_t *pk_arena = nullptr;
	if (packed_ingest) {
		pk = phyfasta::read_genomes_packed(files, read_threads, &read_error, &pk_arena);
		q.resize(pk.size());
		for (size_t i = 0; i < pk.size(); i++) q[i].name = pk[i].name;
	} else {
		q = read_genomes(files, read_threads, &read_error);
	}
	t_read = now_s();
	ctx_thread.join();
	if (!read_error.empty()) die(read_error);
	if (!ctx_error.empty()) die(ctx_error);
	t_ctx = now_s();
	GLEN.resize(q.size());
	for (size_t i = 0; i < q.size(); i++) GLEN[i] = packed_ingest ? (size_t)pk[i].len : q[i].nucl.size();

	size_t ref_idx;
	if (reference_name.empty()) ref_idx = pick_first_pass(q, pk, flags);
	else ref_idx = std::find(files.begin(), files.end(), reference_name) - files.begin();
	// The reference's suffix array: built by the library on the device (the default), or — `--sa=host`, the north
	// star's placement — on the host cores, on a thread of its own while the genomes are uploaded.
	std::vector<int64_t> sa;
	int sa_rc = 0;
	std::thread sa_thread;
	if (host_sa) {
		if (packed_ingest) q[ref_idx].nucl = phyfasta::unpack_genome(pk[ref_idx]);
		sa.resize(2 * q[ref_idx].nucl.size() + 1);
		sa_thread = std::thread([&] { sa_rc = phylo_host_reference_suffix_array(q[ref_idx].nucl.data(), q[ref_idx].nucl.size(), sa.data()); });
	}

	r.q = &q;
	r.flags = flags;
	r.refpos_file = refpos_file;
	if (r.grp) {
		if (threads > 0) gok(r, phylo_group_set_option(r.grp, "host_threads", threads));
		gok(r, phylo_group_set_option(r.grp, "sa_builder", host_sa ? 0 : 1));
	} else {
		if (threads > 0) ok(r, phylo_set_option(r.ctx, "host_threads", threads));
		ok(r, phylo_set_option(r.ctx, "sa_builder", host_sa ? 0 : 1));
	}
	std::vector<const uint32_t *> pk_q2; // the genomes' packed codes (they stay mapped until the process ends: --verify-ranks reads them again)
	if (packed_ingest) {
		std::vector<const uint32_t *> q2(q.size()), bad(q.size());
		std::vector<size_t> nbad(q.size());
		for (size_t i = 0; i < q.size(); i++) {
			q2[i] = pk[i].q2;
			bad[i] = pk[i].bad.data();
			nbad[i] = pk[i].bad.size();
		}
		if (r.grp) gok(r, phylo_group_set_genomes_packed(r.grp, q.size(), q2.data(), GLEN.data(), bad.data(), nbad.data()));
		else ok(r, phylo_set_genomes_packed(r.ctx, q.size(), q2.data(), GLEN.data(), bad.data(), nbad.data()));
		// pk_arena stays mapped until the process ends: unmapping 1.3 GB that has just been the source of
		// device copies costs 0.3 s here (measured at 1024 genomes; the GPU driver's MMU notifier walks the
		// range), the process's exit does not
		pk_q2 = q2;
		for (auto &g : pk)
			if (!g.own) g.q2 = nullptr;
	} else {
		std::vector<const char *> seq(q.size());
		for (size_t i = 0; i < q.size(); i++) seq[i] = q[i].nucl.data();
		ok(r, phylo_set_genomes(r.ctx, q.size(), seq.data(), GLEN.data()));
	}
	t_upload = now_s();
	if (sa_thread.joinable()) sa_thread.join();
	double t_sa = now_s();

	Matrix m = process(r, ref_idx, host_sa && sa_rc == 0 ? sa.data() : nullptr);
	std::vector<int64_t>().swap(sa);
	if (two_pass) {
		ref_idx = pick_second_pass(q.size(), m);
		m = process(r, ref_idx);
	}
	const double t_proc = now_s();
	print_matrix(q, m, flags, bootstrap, ref_idx, prng);
	t_done = now_s();
	if (verify_ranks && r.grp && !(flags & (F_COMPLETE_DELETION | F_POSITIONS)) && !verify_group_result(r, pk, pk_q2, ref_idx, m, device)) RETURN_CODE = 3;
	if (bench_steps > 0 && !(flags & (F_COMPLETE_DELETION | F_POSITIONS))) bench_passes(r, (size_t)bench_steps, m);
	if (timing) {
		auto stat = [&](const char *k) {
			double v = 0;
			phylo_get_stat(r.ctx, k, &v);
			return v / 1e3;
		};
		double bases = 0;
		for (size_t l : GLEN) bases += (double)l;
		fprintf(stderr,
				"timing: genomes %zu  bases %.0f  total %.3f s | read %.3f (%zu threads, %s)  wait-for-device %.3f  upload %.3f (device memory %.3f  copies %.3f  "
				"install %.3f)  "
				"wait-for-suffix-array %.3f (%s)  process+print %.3f (print %.3f)  "
				"[suffix array %.3f  rest of the index %.3f (device allocations %.3f)  anchor %.3f  compare %.3f]\n",
				q.size(), bases, t_done - t_start, t_read - t_start, read_threads, packed_ingest ? "packed" : "bytes", t_ctx - t_read, t_upload - t_ctx, stat("ms:genomes_alloc"), stat("ms:genomes_copy"),
				stat("ms:genomes_install"),
				t_sa - t_upload, host_sa ? "built on a host thread since the files were read" : "built on the device with the index", t_done - t_sa, t_done - t_proc, stat("ms:ref_suffix_array"), stat("ms:ref_total") - stat("ms:ref_suffix_array"), stat("ms:ref_alloc"),
				stat("ms:anchor_total"), stat("ms:compare_total"));
	}
	if (timing && r.grp) {
		fprintf(stderr, "timing: %zu ranks over %s; last pass per rank (ms) anchor / exchange / compare / reduce:", phylo_group_size(r.grp),
				phylo_group_backend(r.grp));
		for (size_t k = 0; k < phylo_group_size(r.grp); k++) {
			double a = 0, e = 0, c2 = 0, d2 = 0;
			phylo_group_get_stat(r.grp, k, "group:ms_anchor", &a);
			phylo_group_get_stat(r.grp, k, "group:ms_exchange", &e);
			phylo_group_get_stat(r.grp, k, "group:ms_compare", &c2);
			phylo_group_get_stat(r.grp, k, "group:ms_reduce", &d2);
			fprintf(stderr, "  [%zu] %.2f / %.2f / %.2f / %.2f", k, a, e, c2, d2);
		}
		fprintf(stderr, "\n");
	}
	// The matrix is out: leave.  Taking the HIP runtime down in an orderly way (contexts, code objects, the driver's
	// queues) costs a tenth of a second that nobody waits for a result in (tools/microbench/startup.hip: the same for an
	// empty program); --teardown does it all the same.
	std::cout.flush();
	fflush(stdout);
	fflush(stderr);
	if (!teardown) _exit(RETURN_CODE);
	if (r.grp) phylo_group_destroy(r.grp);
	else phylo_ctx_destroy(r.ctx);
	if (timing) fprintf(stderr, "timing: releasing the device context%s %.3f s\n", r.grp ? "s" : "", now_s() - t_done);
	return RETURN_CODE;
}

/* phylonium_amd.h — C ABI of the MI355X-native anchor + pairwise-compare path.
 *
 * Drop-in boundary for phylonium's hot path.  The reference has no FFI; its
 * natural seams and what replaces each one here:
 *
 *   B2  std::vector<evo_model> process(const sequence &subject,
 *                                      const std::vector<sequence> &queries)
 *       /root/reference/src/process.h:12, src/process.cxx:408-556
 *         → phylo_set_genomes + phylo_set_reference + phylo_anchor +
 *           phylo_compare_all   (or phylo_process, all four in one call)
 *   B1  evo_model::{account, account_rev, operator+=, estimate_*}
 *       src/evo_model.h:27-36, src/evo_model.cxx:53-131
 *         → the two uint64 tallies per pair written by phylo_compare_*;
 *           phylo_estimate for the host-side distance formulas
 *   B0  size_t seqcmp(const char*, const char*, size_t)      libs/seqcmp.h:14
 *       size_t revseqcmp(const char*, const char*, size_t)   libs/revseqcmp.h:25
 *         → phylo_seqcmp / phylo_revseqcmp (same signature, host pointers) and
 *           phylo_seqcmp_batch over device-resident genomes
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  Every int
 * function returns 0 on success and nonzero on error; the library never calls
 * exit().  phylo_last_error() gives the message.  One host thread drives a
 * context; the library owns its HIP stream; contexts are independent.
 * All compute runs on the GPU — there is no CPU fallback; without a usable
 * device phylo_ctx_create fails.
 */
#ifndef PHYLONIUM_AMD_H
#define PHYLONIUM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct phylo_ctx phylo_ctx;

/* Same fields as class homology, src/process.h:18-26. direction: 0 forward,
 * 1 reverse (homology::dir). */
typedef struct phylo_homology {
	uint64_t index_reference;
	uint64_t index_reference_projected;
	uint64_t index_query;
	uint64_t length;
	int32_t direction;
	int32_t _pad;
} phylo_homology;

/* flags for phylo_process (src/global.h:7-16) */
#define PHYLO_COMPLETE_DELETION 4

/* ── context ── */
int phylo_ctx_create(phylo_ctx **out, int device);
void phylo_ctx_destroy(phylo_ctx *ctx);
/* Run this context's work on the caller's HIP stream (a hipStream_t of the context's device; NULL: back to the
 * context's own): a host that issues collectives on a stream of its own (RCCL, torch.distributed) then needs no
 * synchronisation between its calls and the library's.  The context waits for its current work first. */
int phylo_ctx_set_stream(phylo_ctx *ctx, void *hip_stream);
int phylo_ctx_device(const phylo_ctx *ctx);
/* ctx may be NULL: last error of a failed phylo_ctx_create on this thread. */
const char *phylo_last_error(const phylo_ctx *ctx);
/* Tunables, mostly for tests — every setting gives the same results:
 *   "chunk"            phase-A chunk length (a multiple of 64; 0: the library's plan)
 *   "kmer"             k of the k-mer slot table (0: smallest k with 4^k >= |S|)
 *   "filter"           where phase A's sort + chain filter runs: 0 and 2 on the device (a query whose list has two
 *                      entries with the same projected start still goes to the host, as the reference's order of such
 *                      ties is libstdc++'s), 1 on the host cores
 *   "filter_kernel"    the device filter: 0 stretch by stretch, 1 the general dependent scan only
 *   "fold_blocks"      blocks per query of the fold kernel (0: the library chooses)
 *   "spec_blocks"      at most that many blocks of the speculative chain kernel (0: the library's plan); with a few,
 *                      every lane takes chunk after chunk from the work queue
 *   "lean_force_slow"  1: every step of the chain kernels through their wave-cooperative slow resolver
 *   "cache_quirk"      1 (default): a subject on which the reference's 6-mer cache holds over-deep intervals is matched
 *                      as the reference matches it; 0: true longest matches (results then differ from the reference's)
 *   "sa_builder"       the reference's suffix array when the caller brings none: 1 on the device, 0 on the host cores
 *   "compare_backend"  0 pileup, 1 explicit segment list (the literal seqcmp/revseqcmp calls)
 *   "pairs_kernel"     phase B's pair tallies: 0 on the matrix cores, 1 on the vector ALUs
 *   "pairs_wchunk"     windows per chunk of the pair kernels (0: chosen from the L2 size)
 *   "result_zero_copy" 0 (default) / 1: see phylo_triangle_to_matrices
 *   "host_threads"     size of the context's host worker pool
 *   "profile"          1: time every kernel with HIP events ("ms:<kernel>" stats; two events and ~4 us a kernel),
 *                      2: the speculative chain kernel only */
int phylo_set_option(phylo_ctx *ctx, const char *key, long value);
/* Accumulated since the last phylo_reset_stats: "ms:<kernel>", "n:<kernel>"
 * (HIP-event time and launch count per kernel when profiling is on),
 * "ms:anchor_total", "ms:compare_total", "ms:host_sort_filter",
 * "bytes:compare_sites", ... Returns nonzero for an unknown key. */
int phylo_get_stat(phylo_ctx *ctx, const char *key, double *out);
int phylo_reset_stats(phylo_ctx *ctx);
/* NUL-separated list of stat keys, double-NUL terminated; returns bytes needed. */
size_t phylo_stat_keys(phylo_ctx *ctx, char *buf, size_t cap);

/* ── genomes: `queries` of process(), src/process.cxx:408-409 ──
 * Bytes are the joined sequences as phylonium holds them: A,C,G,T with '!'
 * between contigs (src/sequence.cxx:171-199). Copied to the device. */
int phylo_set_genomes(phylo_ctx *ctx, size_t n, const char *const *seq, const size_t *len);
/* Same, for genomes already resident in device memory (one allocation;
 * genome j at dev_base+offsets[j], offsets multiples of 64 and >= 64, each
 * genome followed by at least 64 zero bytes — the kernels read whole 16-byte
 * pieces up to 32 bytes before and 64 bytes after a genome, and prefetch
 * 128-byte windows: 256 readable bytes must follow the last genome's padding).
 * Borrowed until the next set_genomes or destroy. */
int phylo_set_genomes_device(phylo_ctx *ctx, size_t n, const void *dev_base, const uint64_t *offsets,
							 const uint64_t *lens);

/* Same genomes, handed over as 2-bit codes: q2[j] holds (len[j] + 15) / 16 words, 16 bases per word, the
 * first base in bits 31..30, A0 C1 G2 T3, codes behind the last base 0; bad[j][0..nbad[j]) are the ascending
 * positions of the '!' separators (their code is 0; code bits behind a genome's end or under a separator that are not
 * 0 are cleared by the device).  This is the form phase A's kernels read, so it is copied
 * into place — a quarter of the bytes of phylo_set_genomes cross PCIe — and the byte form the other kernels
 * read is written by the device.  The FASTA side of it: phylo_host_read_fasta_packed (src/sequence.cxx:109-199
 * fused with the packing). */
int phylo_set_genomes_packed(phylo_ctx *ctx, size_t n, const uint32_t *const *q2, const size_t *len,
							 const uint32_t *const *bad, const size_t *nbad);
/* The packed form already in device memory (one buffer for all genomes: word w holds the codes of bytes
 * [16w, 16w + 16) of the layout phylo_set_genomes_device describes — genome j at byte offset offsets[j] — and 0
 * everywhere else; the buffer must reach 16 bytes' worth of words past the last genome's padding).  Copied; the
 * byte form is written by the device.  What a rank of a multi-GPU run holds after gathering the other ranks'
 * packed blocks. */
int phylo_set_genomes_packed_device(phylo_ctx *ctx, size_t n, const void *dev_q2, const uint64_t *offsets,
									const uint64_t *lens, const uint32_t *const *bad, const size_t *nbad);
/* Genome i as bytes (len[i] of them, no terminator), copied from the device: a host that ingested packed
 * genomes needs the reference's bytes for the suffix array, and all of them for -p. */
int phylo_get_genome(phylo_ctx *ctx, size_t i, char *buf);

/* ── reference: `esa ref(subject)` + threshold, src/process.cxx:413-417 ──
 * sa: suffix array of S = subject + '#' + revcomp(subject), 2L+1 entries,
 * exactly what divsufsort64 returns at src/esa.cxx:74; NULL builds it on the
 * device (option "sa_builder" = 1, the default: prefix doubling over radix sorts, csrc/sa_kernels.hip) or on the
 * host cores ("sa_builder" = 0: bucket sort / SA-IS). threshold 0 computes
 * min_anchor_length(0.025, gc, 2L+1) as src/process.cxx:416-417.  The array is checked on the device while the LCP
 * values are made (every suffix against its successor): one that is not the suffix array of S makes the call fail; a
 * device-built one that fails is built again on the host cores. */
int phylo_set_reference(phylo_ctx *ctx, size_t ref_idx, const int64_t *sa, size_t threshold);
size_t phylo_threshold(const phylo_ctx *ctx);
/* The suffix array the current reference's index was built from (2L+1 entries, as divsufsort64 would return
 * them at src/esa.cxx:74), copied from the device: with sa == NULL above it was built there (option
 * "sa_builder": 1, the default, prefix doubling on the device; 0 the host cores). */
int phylo_reference_suffix_array(phylo_ctx *ctx, int64_t *sa);
/* 1 when the reference build does NOT report the longest match on this subject: its 6-mer interval cache
 * stores an over-deep interval when a nucleotide string of <= 4 characters occurs at least twice in S and
 * only in front of the same contig join (src/esa.cxx:174-199) — possible for references of a few kbp
 * in several contigs, never beyond.  By default (option "cache_quirk" = 1) this library reproduces what the
 * reference answers there, so results equal the reference's; with "cache_quirk" = 0 it computes the true longest
 * matches and results then differ.  A host may want to say so (phylonium-amd -v does). */
int phylo_reference_cache_quirk(const phylo_ctx *ctx);

/* ── phase A: anchor_homologies + sort + filter_overlaps_max for queries
 * [q_begin, q_end), src/process.cxx:433-458 ── */
int phylo_anchor(phylo_ctx *ctx, size_t q_begin, size_t q_end);
/* ctx-owned result of phase A (or of phylo_set_homologies) for genome j.  After phase A the
 * lists may exist only in device memory (that is where phase B reads them); they are copied
 * to the host on the first call that asks for them. */
int phylo_get_homologies(phylo_ctx *ctx, size_t j, const phylo_homology **h, size_t *n);
/* Install lists computed elsewhere (another rank). */
int phylo_set_homologies(phylo_ctx *ctx, size_t j, const phylo_homology *h, size_t n);
/* 16-byte wire form of a homology for the exchange between ranks (lossless:
 * index_reference follows from the projected start, the length and the
 * direction, src/process.h:72-80). */
typedef struct phylo_packed_homology {
	uint32_t start; /* index_reference_projected */
	uint32_t index_query;
	uint32_t length;
	uint32_t direction;
} phylo_packed_homology;

/* Bulk forms for the exchange between ranks: counts[j - q_begin] and the lists
 * of genomes [q_begin, q_end) back to back in buf. Export returns the number
 * of entries via *total and copies them when cap suffices (call with cap = 0
 * to size the buffer). */
int phylo_export_homologies(phylo_ctx *ctx, size_t q_begin, size_t q_end, uint64_t *counts,
							phylo_homology *buf, size_t cap, size_t *total);
int phylo_import_homologies(phylo_ctx *ctx, size_t q_begin, size_t q_end, const uint64_t *counts,
							const phylo_homology *buf);
/* The same with 16-byte records (needs a reference to be set: the conversion uses L). */
int phylo_export_packed(phylo_ctx *ctx, size_t q_begin, size_t q_end, uint64_t *counts,
						phylo_packed_homology *buf, size_t cap, size_t *total);
int phylo_import_packed(phylo_ctx *ctx, size_t q_begin, size_t q_end, const uint64_t *counts,
						const phylo_packed_homology *buf);
/* Device-resident forms for one process per GPU (the records never visit the host
 * between ranks).  Export: the lists of genomes [q_begin, q_end) back to back into
 * device memory dev_dst (cap records; call with dev_dst = NULL to size it).
 * Attach: from now on genome g's filtered list is dev_records[begin[g] ..
 * begin[g] + count[g]) in device memory, for all g — e.g. the output of an
 * all-gather of every rank's export.  The buffer is borrowed until the next
 * phylo_anchor / phylo_set_* call.  Host-side lists of genomes in
 * [keep_begin, keep_end) are kept as they are (the caller computed them here);
 * the others are read back from the buffer only if somebody asks for them.  Every list must be sorted by
 * projected start, pairwise disjoint and inside the reference (what phase A leaves; checked on the device:
 * the call fails otherwise).  Lists installed through the host forms (phylo_set_homologies, phylo_import_*)
 * may be anything compare() of src/process.cxx:566-611 accepts: overlapping or unsorted ones are tallied by
 * the segment backend. */
int phylo_export_packed_device(phylo_ctx *ctx, size_t q_begin, size_t q_end, void *dev_dst, size_t cap,
							   uint64_t *counts, size_t *total);
int phylo_attach_packed_device(phylo_ctx *ctx, const void *dev_records, const uint64_t *begin,
							   const uint64_t *count, size_t keep_begin, size_t keep_end);
/* The same exchange without the host in it (one process per GPU, or one host thread per GPU): every rank writes a
 * block of fixed shape — 4 header words {records, overflow, queries, 0}, max_queries list lengths (u32, a multiple
 * of 4 of them), cap_records records of 16 bytes; phylo_exchange_block_bytes gives its size — one all-gather of the
 * blocks assembles the lists of all ranks on every GPU, and the receiving side works the per-genome ranges out on
 * the device.  Neither call waits for the device: they queue work on the context's stream (phylo_ctx_set_stream).
 * Export needs the lists where phase A's device filter left them (phylo_anchor(q_begin, q_end) was this context's
 * last list-changing call; nonzero otherwise: use phylo_export_packed_device).  Attach: bounds[r] .. bounds[r+1] are
 * rank r's genomes (bounds[0] = 0, bounds[world] = n); a block that overflowed cap_records, or a list that is not
 * sorted, disjoint and inside the reference, makes the NEXT call that looks at the lists fail (phylo_compare*,
 * phylo_get_homologies) — size cap_records from an earlier step's counts and repeat with the host forms then. */
size_t phylo_exchange_block_bytes(size_t max_queries, size_t cap_records);
int phylo_export_block_device(phylo_ctx *ctx, size_t q_begin, size_t q_end, void *dev_block, size_t max_queries,
							  size_t cap_records);
int phylo_attach_blocks_device(phylo_ctx *ctx, const void *dev_all, size_t world, const size_t *bounds, size_t max_queries,
							   size_t cap_records, size_t keep_begin, size_t keep_end);
/* Phase A of a rank's block of queries with its exchange block written behind it — phylo_anchor(q_begin, q_end) +
 * phylo_export_block_device without the host round trip between them (the loop of src/process.cxx:433-458 for this
 * rank's queries, queued): nothing is waited for, the caller's all-gather goes straight behind it on the context's
 * stream.  What phylo_anchor would have learnt at its wait — a list with tied projected starts, which only the host's
 * std::sort orders as the reference does; scratch that overflowed — rides in the block's header to every rank and
 * comes back in the summed triangle's report (word 4, below): the pass is then repeated with phylo_anchor +
 * phylo_export_block_device.  Until phylo_attach_blocks_device has been given the gathered blocks, any call that asks
 * for this context's lists waits for the queued phase A first. */
int phylo_anchor_block_device(phylo_ctx *ctx, size_t q_begin, size_t q_end, void *dev_block, size_t max_queries,
							  size_t cap_records);
/* complete_delete over all genomes' lists, src/process.cxx:467-469,725-776 (host). */
int phylo_complete_delete(phylo_ctx *ctx);

/* ── phase B: the pair grid, src/process.cxx:517-549 ──
 * subst / homologs: caller-owned N*N row-major, symmetric, diagonal 0.
 * Parts (part in [0,nparts)) partition the work — the pileup backend by range
 * of reference windows, the segment backend by pair — and each call writes the
 * tallies of its part; summing the outputs of all parts gives the full matrix
 * (one all-reduce across ranks). */
int phylo_compare(phylo_ctx *ctx, size_t part, size_t nparts, uint64_t *subst, uint64_t *homologs);
int phylo_compare_all(phylo_ctx *ctx, uint64_t *subst, uint64_t *homologs);
/* phylo_compare with the two N*N tallies left in device memory (for a device-side all-reduce). */
int phylo_compare_device(phylo_ctx *ctx, size_t part, size_t nparts, uint64_t *dev_subst, uint64_t *dev_homologs);

/* The tallies of a part as a u32 upper triangle in device memory — what crosses the wire between ranks: tri[k]
 * substitutions and tri[P + k] homologs of pair i < j, k = i (2n - i - 1) / 2 + (j - i - 1), P = n (n - 1) / 2 (a tally
 * is at most the reference's length, < 2^31; a quarter of the bytes of the two u64 matrices), followed by eight words of
 * the part's own: what its comparison has to report ('!' list overflow, a gathered list out of order, a gathered block
 * beyond its capacity, 1 per part, a rank's phase A needs the host (phylo_anchor_block_device), 0, 0, 0) —
 * phylo_triangle_words(n) = n (n - 1) + 8 words in all.  On the default (matrix-core) path the
 * call queues its kernels and returns without waiting for them: the parts' triangles AND their reports add up (one
 * all-reduce / reduce of phylo_triangle_words(n) words on the context's stream), and phylo_triangle_to_matrices — on the
 * rank that wants the result — writes the two symmetric n x n matrices process() returns and fails if any part
 * reported.  With option "result_zero_copy" = 1 a caller that hands the same (16-byte aligned) host matrices of a
 * megabyte or more over again and again has them written by the device directly (they are registered with the HIP runtime
 * on their second use and stay so while the context lives).  Off by default: such memory must not be freed, reallocated
 * or forked over while it is registered — set the option back to 0 (which lets go of it) before any of that. */
size_t phylo_triangle_words(size_t n);
int phylo_compare_triangle_device(phylo_ctx *ctx, size_t part, size_t nparts, uint32_t *dev_tri);
int phylo_triangle_to_matrices(phylo_ctx *ctx, const uint32_t *dev_tri, uint64_t *subst, uint64_t *homologs);

/* The result's own home on the host: page-locked memory the library owns, which the device writes directly — no
 * registration of the caller's pages (option "result_zero_copy"), no staging copy.  shm_name = NULL: private to this
 * context; phylo_triangle_to_matrices (and phylo_compare*) recognise the two pointers phylo_result_matrices hands out
 * and write there over PCIe.  shm_name = "/name": a POSIX shared-memory segment for the `ranks` ranks of one node (one
 * process per GPU, or one thread per GPU): one rank creates it (create = 1), the others open it, the creator unlinks the
 * name once all have (phylo_result_unlink; the memory lives as long as a mapping does).  Then, after the all-reduce
 * of the parts' triangles, every rank's device writes ITS rows of both matrices over its own PCIe link:
 * phylo_triangle_rows_to_result queues rows [row_begin, row_end), waits for this context's stream, records the
 * delivery in the segment's header and — wait_ranks > 0: the rank(s) that want the result — returns when ranks
 * 0 .. wait_ranks - 1 have recorded theirs (the ranks call it in step, once per pass).  report (8 words, may be NULL):
 * the summed triangle's report as described above.  2 N^2 x 8 bytes cross eight links instead of one: at N = 1024 the
 * result's way home shrinks from 0.32 ms to 0.05. */
int phylo_result_open(phylo_ctx *ctx, const char *shm_name, int create, size_t n, size_t ranks);
int phylo_result_unlink(phylo_ctx *ctx);
void phylo_result_close(phylo_ctx *ctx);
int phylo_result_matrices(phylo_ctx *ctx, uint64_t **subst, uint64_t **homologs);
/* A rank that gives a pass up after the others may already be waiting for its rows says so in the segment's header: their
 * phylo_triangle_rows_to_result returns an error at once (otherwise after a minute).  The segment is closed and opened
 * anew afterwards. */
int phylo_result_abandon(phylo_ctx *ctx, size_t rank);
int phylo_triangle_rows_to_result(phylo_ctx *ctx, const uint32_t *dev_tri, size_t row_begin, size_t row_end, size_t rank,
								  size_t wait_ranks, uint32_t *report);

/* ── B2 in one call ── */
int phylo_process(phylo_ctx *ctx, size_t ref_idx, int flags, uint64_t *subst, uint64_t *homologs);
/* phylo_anchor(all genomes) + phylo_compare_all against the reference already set, as the one call they are in
 * process() (src/process.cxx:408-556): phase B is queued behind phase A and the host waits once, for the result. */
int phylo_anchor_compare(phylo_ctx *ctx, uint64_t *subst, uint64_t *homologs);

/* ── several GPUs of one node behind one host (csrc/group.hip) ──
 * process() shards without a data-path collective inside either phase: phase A by query block (the loop at
 * src/process.cxx:433-434), phase B by range of reference windows (the pair loop at src/process.cxx:524-529 re-cut so
 * that projection and pair kernel both shrink with the ranks).  A group is one context and one host thread per rank;
 * the ranks meet three times — the packed genomes (an all-gather of the blocks each rank uploaded), the filtered lists
 * after phase A (an all-gather of fixed-shape device blocks), the tallies after phase B (an all-reduce of u32 triangles,
 * after which every rank's device writes its rows of the result into the node's page-locked home of it) — over RCCL /
 * xGMI when every rank has a GPU of its own (the library is loaded when a group is made), by device-to-device copies
 * when ranks share a GPU.  phylo_group_process queues a rank's whole pass on the rank's stream — phylo_anchor_block_device,
 * all-gather in place, phylo_attach_blocks_device, phylo_compare_triangle_device, all-reduce, phylo_triangle_rows_to_result —
 * and waits once; a pass whose summed report asks for it (blocks that overflowed, a list that needs the host's std::sort,
 * more '!' than the lists hold) is repeated by all ranks the long way.  devices: n_ranks ordinals (NULL: rank r on
 * device r modulo the device count).  Results are identical to one context's for any number of ranks. */
typedef struct phylo_group phylo_group;
int phylo_group_create(phylo_group **out, size_t n_ranks, const int *devices);
void phylo_group_destroy(phylo_group *g);
const char *phylo_group_last_error(const phylo_group *g); /* g may be NULL: a failed phylo_group_create on this thread */
size_t phylo_group_size(const phylo_group *g);
phylo_ctx *phylo_group_ctx(phylo_group *g, size_t rank);   /* rank 0 holds every list after phylo_group_anchor */
size_t phylo_group_rank_begin(const phylo_group *g, size_t rank); /* first genome of the rank's block (rank = size: n) */
const char *phylo_group_backend(const phylo_group *g);     /* "rccl", "device-to-device copies" or "one rank" */
/* phylo_set_option on every rank; and the group's own "exchange_cap": records per exchange block of the next plan
 * (0, the default: sized from the lists' lengths, which costs one host round trip per new reference; a pass whose lists
 * outgrow a pinned capacity is repeated by phylo_group_process with a plan of its own) */
int phylo_group_set_option(phylo_group *g, const char *key, long value);
/* phylo_get_stat of a rank, plus "group:ms_anchor", "group:ms_exchange", "group:ms_compare", "group:ms_reduce" (the
 * rank's host-side milliseconds in the last pass: with a queued pass the first three are the time it took to queue the
 * work), "group:ms_queued" (start of the pass to everything queued), "group:ms_step" (to the rank's rows delivered),
 * "group:replans", "group:passes_repeated", "group:shared_result" (1: the node's shared home of the result is in use) */
int phylo_group_get_stat(phylo_group *g, size_t rank, const char *key, double *out);
/* the arguments of phylo_set_genomes_packed: rank r uploads its block of the genomes, one all-gather does the rest */
int phylo_group_set_genomes_packed(phylo_group *g, size_t n, const uint32_t *const *q2, const size_t *len,
								   const uint32_t *const *bad, const size_t *nbad);
int phylo_group_set_reference(phylo_group *g, size_t ref_idx, const int64_t *sa, size_t threshold);
int phylo_group_anchor(phylo_group *g);                                    /* phase A + the lists to every rank */
int phylo_group_compare(phylo_group *g, uint64_t *subst, uint64_t *homologs); /* phase B + the sum + the result */
int phylo_group_process(phylo_group *g, uint64_t *subst, uint64_t *homologs); /* both, as one queue per rank */
/* The group's own home of the result (several ranks, after the first pass): the two n x n matrices every rank's device
 * writes its rows of.  Passed to phylo_group_process / phylo_group_compare as subst / homologs (or NULL, NULL) the result
 * stays there — a caller's own matrices are filled by the ranks' threads, each copying its rows. */
int phylo_group_result_matrices(phylo_group *g, uint64_t **subst, uint64_t **homologs);
/* the ranks RCCL itself counts in the group's communicator (ncclCommCount; 0 when the ranks do not talk over RCCL) */
size_t phylo_group_rccl_ranks(const phylo_group *g);

/* ── B0 ── */
size_t phylo_seqcmp(const char *begin, const char *other, size_t length);
size_t phylo_revseqcmp(const char *begin, const char *other, size_t length);
/* n segments over the resident genomes: compare genome ga[s] at offa[s] with
 * genome gb[s] at offb[s] for len[s] bytes; rev[s] != 0 uses revseqcmp
 * semantics (offb is the start of the reversed window, as in
 * evo_model::account_rev, src/evo_model.cxx:68-75). */
int phylo_seqcmp_batch(phylo_ctx *ctx, size_t n, const uint32_t *ga, const uint64_t *offa,
					   const uint32_t *gb, const uint64_t *offb, const uint64_t *len, const uint8_t *rev,
					   uint64_t *out);

/* ── host-side helpers (no GPU, except the first) ── */
int phylo_host_device_count(int *count); /* HIP devices this process sees; nonzero when the runtime cannot say */
int phylo_host_suffix_array(const char *s, size_t n, int64_t *sa);
/* Suffix array of S = ref + '#' + revcomp(ref) (2*len + 1 entries, src/esa.cxx:72-75), the `sa`
 * argument of phylo_set_reference: lets a host build it on another thread while genomes are
 * still being uploaded. */
int phylo_host_reference_suffix_array(const char *ref, size_t len, int64_t *sa);
size_t phylo_host_min_anchor_length(double p, double gc, size_t l);
/* FASTA files -> genomes as phylo_set_genomes wants them (nucleotides filtered to ACGT and
 * upper-cased, the records of a file joined by '!': src/sequence.cxx:109-199 over libs/pfasta.c),
 * read on up to `threads` host threads.  out[i] receives a buffer owned by the caller (release
 * with phylo_host_free), len[i] its length.  Returns 0, or 1 + the index of the first file in
 * the order given that could not be read (its message: phylo_last_error(NULL)); nothing is
 * handed over in that case. */
int phylo_host_read_fasta(size_t n, const char *const *paths, size_t threads, char **out, size_t *len);
/* The same files as 2-bit codes + separator positions, the arguments of phylo_set_genomes_packed: files are
 * mapped, not copied, and lines of nucleotides are packed 32 bytes at a time (AVX2 + BMI2 when the CPU has
 * them).  q2[i] and bad[i] point into storage owned by *arena (one allocation for all genomes: a thousand
 * buffers of their own would cost a thousand munmaps under the GPU driver's MMU notifier once they have been
 * the source of a copy); release it with phylo_host_free_packed when the genomes are on the device.  Errors as
 * phylo_host_read_fasta. */
int phylo_host_read_fasta_packed(size_t n, const char *const *paths, size_t threads, uint32_t **q2, size_t *len,
								 uint32_t **bad, size_t *nbad, void **arena);
void phylo_host_free_packed(void *arena);
void phylo_host_free(void *p);
/* The first-pass reference of src/phylonium.cxx:360-382: the genome std::nth_element leaves at
 * the middle position when ordering by length (for equal lengths that depends on the library's
 * algorithm, which is why it is offered here rather than restated by each host). */
size_t phylo_host_median_length_index(size_t n, const size_t *len);
/* std::sort by projected start + filter_overlaps_max, in place; returns new n */
size_t phylo_host_sort_filter(phylo_homology *h, size_t n, int do_sort);
/* kind: 0 Jukes-Cantor, 1 raw, 2 ANI (src/evo_model.cxx:100-131) */
double phylo_estimate(int kind, uint64_t subst, uint64_t homologs, int zero_on_error);
/* PHYLIP text of src/io.cxx:141-163; returns bytes needed including NUL */
size_t phylo_format_phylip(size_t n, const char *const *names, const uint64_t *subst,
						   const uint64_t *homologs, int kind, char *out, size_t cap);
const char *phylo_version(void);

#ifdef __cplusplus
}
#endif
#endif

// kernels.h — launch wrappers shared between the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "anchor_core.h"
#include "lean_core.h"

namespace phy {

// lean_kernels.hip: the chain kernels on 2-bit packed operands (default), and the packed tables
int lean_spec_resident_blocks(int n_cu);
void launch_lean_spec(const PhaseA &A, const RefIndex &R, const LeanIndex &X, int n_cu, hipStream_t st, int max_blocks = 0); // max_blocks > 0: no more blocks than that
void launch_lean_overruns(const PhaseA &A, const RefIndex &R, uint32_t nq, hipStream_t st); // between spec and bridge
// Z: up to three short arrays of counters that the kernels behind the bridges start from zero (p[0] of any length, p[1] and
// p[2] of at most a block's worth of words): the prepare kernel zeroes them on its way
struct BridgeZero {
	uint32_t *p[3];
	uint32_t n[3];
};
void launch_lean_bridge(const PhaseA &A, const RefIndex &R, const LeanIndex &X, int n_cu, hipStream_t st, const BridgeZero &Z = BridgeZero{{nullptr, nullptr, nullptr}, {0, 0, 0}});
// the speculative kernel's work queue as items and query descriptors (anchor_core.h), once per plan
void launch_lean_work(const PhaseA &A, const LeanIndex &X, WorkItem *work, QDesc *qdesc, uint32_t nq, hipStream_t st);
void launch_pack2(const uint8_t *src, uint64_t bytes, uint32_t *dst, hipStream_t st); // bytes: a multiple of 16
// Q2 → byte arena (bytes: the whole arena, a multiple of 16), then '!' at the nbad listed positions; code bits of Q2
// outside the genomes or under a separator are cleared on the way
void launch_unpack2(uint32_t *q2, const uint64_t *off, const uint32_t *len, uint32_t n, uint64_t bytes, uint8_t *dst,
					const uint32_t *bad, const uint32_t *bad_off, uint32_t nbad, hipStream_t st);
uint32_t bad_segment_bytes();
// non-ACGT positions per sequence; out == nullptr: count per segment into seg_cnt, else write from seg_off
void launch_bad_positions(const uint8_t *base, const uint64_t *off, const uint32_t *len, const uint32_t *seg_seq,
						  const uint32_t *seg_first, uint32_t nseg, uint32_t *seg_cnt, const uint32_t *seg_off, uint32_t *out,
						  hipStream_t st);

// anchor_kernels.hip
// queries [j0, j1), blocks_per_query blocks each (with more than one a query's homologies leave in the order its
// blocks got their slots, not in query order)
void launch_fold(const PhaseA &A, uint32_t j0, uint32_t j1, uint32_t border, uint32_t thr, RawHom *out,
				 const uint64_t *out_base, const uint32_t *out_cap, uint32_t *out_cnt, hipStream_t st, uint32_t blocks_per_query = 1,
				 bool zero_cnt = true); // zero_cnt = false: out_cnt[j0, j1) is zero already (launch_lean_bridge's BridgeZero)

// filter_kernels.hip: reverseEh + sort + filter_overlaps_max per query on the device; flag[j] = 1
// leaves query j to the host (two entries share a projected start, or the list is too long)
struct DevHom;
void launch_sort_filter(const RawHom *raw, const uint64_t *raw_base, const uint32_t *raw_cnt, uint32_t j0, uint32_t j1, uint32_t border,
						uint32_t threshold, uint32_t ref_local, DevHom *out, uint32_t *rng, uint32_t *total, uint32_t *flag,
						hipStream_t st, int variant = 0, int long_ok = 0); // queries [j0, j1); variant 0: stretch by stretch (default), 1: the general kernel only; long_ok: lists beyond the block's LDS are left to launch_sort_filter_long
// lists of more than long_filter_min_entries() entries (flagged by launch_sort_filter with long_ok): sorted in global memory
size_t long_filter_scratch_bytes();
uint32_t long_filter_min_entries();
void launch_sort_filter_long(const RawHom *raw, const uint64_t *raw_base, const uint32_t *raw_cnt, uint32_t j0, uint32_t j1,
							 uint32_t border, DevHom *out, uint32_t *rng, uint32_t *total, uint32_t *flag, void *scratch,
							 uint32_t *slot_counter, hipStream_t st);

// index_kernels.hip
// S = ref + '#' + revcomp(ref) + 64 zero bytes; *gc += the subject's G/C count (zeroed by the caller)
void launch_build_subject(const uint8_t *ref, uint32_t L, uint8_t *S, unsigned long long *gc, hipStream_t st);
// masks[next_bytes_entries()] (zeroed by the caller): the bytes seen after every nucleotide string of 1..5 letters
uint32_t next_bytes_entries();
void launch_next_bytes(const uint8_t *S, uint32_t n, uint32_t *masks, hipStream_t st);
void launch_lcp(const uint8_t *S, const uint32_t *SA, uint32_t n, uint32_t cap, uint32_t *LCP, uint32_t *capped,
				hipStream_t st);
void launch_kmer_table(const uint8_t *S, uint32_t n, uint32_t k, uint32_t *T, uint32_t *scratch_sums, hipStream_t st);
size_t kmer_table_scratch(uint32_t k);
void launch_sax(const uint8_t *S, const uint32_t *SA, const uint32_t *LCP, uint32_t n, U4 *SAX, hipStream_t st);

// sa_kernels.hip: the suffix array on the device (prefix doubling over rocPRIM's radix sort).  Returns 0, 1 when S
// holds a byte other than ! # A C G T (build it on the host then), 2 on a HIP error.
size_t suffix_array_scratch_bytes(uint32_t n);
int device_suffix_array(const uint8_t *S, uint32_t n, uint32_t *SA, void *scratch, uint32_t *rounds_out, hipStream_t st);

// seqcmp_kernels.hip
struct Segment {
	uint64_t a;   // byte offset of the first string in the genome buffer
	uint64_t b;   // byte offset of the second string (start of the window, also when rev)
	uint32_t len;
	uint32_t rev; // 0: seqcmp, 1: revseqcmp
};
// words behind a part's u32 triangle (phylo_triangle_words): {the projection's list of '!' overflowed, a gathered list is not
// sorted and disjoint, a gathered block overflowed its capacity, 1 per part, a rank's phase A needs the host, 0, 0, 0}
static const uint32_t TRI_TAIL = 8;
static const uint32_t SEQCMP_PIECE = 4096; // bytes of either string a wavefront has in flight: four 16-byte chunks per lane and string
static const uint32_t SEQCMP_ROUND = 1008; // a round: 63 chunks (lane 63 provides its neighbour's fifth dword); a batch's segments are cut into rounds, four rounds a pass
// out[s] = seqcmp / revseqcmp of segment s, ADDED to out[] (zeroed by the caller).
// one != nullptr: the batch's only segment, as the host holds it (it comes with the kernel's arguments: nothing is looked up).
// Otherwise round0 = device array of nseg + 1 prefix sums of the segments' rounds (ceil(len / SEQCMP_ROUND)), nrounds =
// round0[nseg]; hint = device array with the segment round seqcmp_rounds_per_pass() * p lies in, for every pass p; rounds =
// device scratch of seqcmp_rounds_bytes(nrounds) for the rounds' descriptors.
uint32_t seqcmp_rounds_per_pass();
size_t seqcmp_rounds_bytes(uint64_t nrounds);
void launch_seqcmp_batch(const uint8_t *base, const Segment *segs, uint32_t nseg, const uint32_t *round0, const uint32_t *hint, uint32_t nrounds,
						 void *rounds, uint64_t *out, int n_cu, hipStream_t st, const Segment *one = nullptr);

// pileup_kernels.hip
struct DevHom {
	uint32_t start; // index_reference_projected
	uint32_t iq;    // index_query
	uint32_t len;
	uint32_t rev;
};
struct Pileup {
	uint32_t *plane[5]; // V, N0, N1, D (reverse), B ('!'); each [W][Npad] for this part's windows
	uint32_t W;         // words (32 reference positions each) held by this part
	uint32_t w0;        // first word of this part on the reference (multiple of 64)
	uint32_t N, Npad;
	uint32_t L;
};
// where the projection reads the query side: the genomes as 2-bit codes (lean_core.h: Q2; word offset of genome g =
// goff[g] / 16) and their non-ACGT positions (QBAD, list g at [qbad_off[g], qbad_off[g+1]))
struct QuerySrc {
	const uint32_t *q2;
	const uint64_t *goff;
	const uint32_t *qbad;
	const uint32_t *qbad_off;
};
// hom_rng[2g], hom_rng[2g+1]: genome g's sorted, disjoint list is homs[begin, end).
// Both take a range — genomes [g0, g1), genome tiles [tg0, tg1) of project_genomes_per_tile()
// genomes — so that the projection can run for the genomes whose lists are ready.
// zero_flags (may be null): the projection's flag words [0] and [3] of it are zeroed on the way (the kernel runs before the projection)
void launch_tile_index(const Pileup &P, const QuerySrc &Q, const DevHom *homs, const uint32_t *hom_rng, uint32_t *first, uint32_t g0,
					   uint32_t g1, hipStream_t st, uint32_t *zero_flags = nullptr);
// five_planes = false: V, N0, N1 only; *bang_flag is raised when '!' was projected and D, B are needed after all
// bang_list / bang_cap (three planes only; may be null / 0): the projected '!' are listed there — {genome | reverse << 31,
// position} each, counted in bang_flag[3], bit 1 of bang_flag[0] on overflow — for launch_bang_correct
void launch_project(const Pileup &P, bool five_planes, const QuerySrc &Q, const DevHom *homs,
					const uint32_t *hom_rng, const uint32_t *first, uint32_t *bang_flag, uint32_t tg0, uint32_t tg1,
					hipStream_t st, uint32_t *bang_list, uint32_t bang_cap, const int *resident_blocks);
// blocks of the projection the current device holds at once, {three planes, five planes}: worked out once per context
// (phylo_ctx_create) and handed to launch_project
void project_resident_blocks(int out[2]);
// the substitutions the three planes miss: '!' against 'A' in the same direction (one block per listed '!')
void launch_bang_correct(const Pileup &P, const QuerySrc &Q, const DevHom *homs, const uint32_t *hom_rng, const uint32_t *list,
						 const uint32_t *count, uint32_t cap, unsigned long long *subst, hipStream_t st);
uint32_t project_genomes_per_tile();
size_t project_index_entries(const Pileup &P);
// tiles: list of (ig, jt) pairs packed as ig<<16|jt
void launch_pairs(const Pileup &P, bool with_bang, const uint32_t *tiles, uint32_t ntiles, uint32_t wchunk,
				  unsigned long long *subst, unsigned long long *homologs, hipStream_t st);

// the same tallies on the matrix cores (three planes only): tiles of pairs_mfma_tile() x pairs_mfma_tile() genomes,
// packed as ti << 16 | tj with ti <= tj; wchunk <= pairs_mfma_max_wchunk()
uint32_t pairs_mfma_tile();
uint32_t pairs_mfma_max_wchunk();
void launch_pairs_mfma(const Pileup &P, const uint32_t *tiles, uint32_t ntiles, uint32_t wchunk, unsigned long long *subst,
					   unsigned long long *homologs, hipStream_t st, uint32_t cpw = 1, // cpw: window chunks (of one XCD) a wavefront takes in a row
					   unsigned long long *clk = nullptr); // profiling: clk[0] += the wavefronts' shader cycles, clk[1] += their 100 MHz ticks

void launch_symmetrise(uint32_t N, unsigned long long *a, unsigned long long *b, hipStream_t st);
// *bad = 1 unless every genome's list is sorted by projected start, disjoint and inside [0, L)
void launch_check_lists(const DevHom *homs, const uint32_t *hom_rng, uint32_t N, uint32_t L, uint32_t *bad, hipStream_t st);

static const uint32_t PAIR_IG = 16; // i-genomes per block (scalar side)
static const uint32_t PAIR_JT = 64; // j-genomes per block (one per lane)

} // namespace phy

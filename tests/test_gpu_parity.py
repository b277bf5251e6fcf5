"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs.  Integer tallies, thresholds and homology
lists must be bit-exact; distances are computed from the integers with the
same libm, so they are compared with == (tolerance stated by the north star:
1e-12)."""
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from phylonium_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def hom_tuples_gpu(h):
    return [(int(b["direction"]), int(b["index_reference"]), int(b["index_reference_projected"]),
             int(b["index_query"]), int(b["length"])) for b in h]


def hom_tuples_orc(h):
    return [(int(a["rev"]), int(a["iref"]), int(a["iproj"]), int(a["iq"]), int(a["len"])) for a in h]


# Option bundles of the parity checks.  Every check_process call runs under ONE named bundle; which one is a function
# of the test's id and of the call's index inside that test — never of what ran before it — so `pytest -k name` hits
# the same combinations as the whole suite, and every assertion message names the bundle.  PHY_OPTION_BUNDLE=<name or
# index> pins one bundle for every call (to reproduce a failure, or to sweep the suite under one configuration).
# "default" is the library as it ships; the others move the steps that have a second home: the sort + chain filter
# (host cores / the device's general kernel), the pair tallies (vector ALUs instead of the matrix cores), the fold's
# blocks per query, the genomes' way in (2-bit codes + separator positions instead of bytes), and "recheck" repeats
# phase A with every step sent through the chains' slow resolver (the definition, from the raw bytes).
BUNDLES = [
    ("default", {}),
    ("default+recheck", {"recheck": 1}),
    ("packed-ingest", {"packed": 1, "fold_blocks": 3}),
    ("host-filter+valu-pairs", {"filter": 1, "pairs_kernel": 1, "fold_blocks": 8, "recheck": 1}),
    ("general-filter-kernel", {"filter": 2, "filter_kernel": 1, "packed": 1, "fold_blocks": 1}),
    ("device-filter+valu-pairs", {"filter": 2, "pairs_kernel": 1}),
    ("few-chain-blocks", {"spec_blocks": 2, "fold_blocks": 2}),
]
BUNDLE_NAMES = [b[0] for b in BUNDLES]
_calls = {}


def pick_bundle(bundle=None):
    """(name, options) for this check: an explicit name, else PHY_OPTION_BUNDLE, else by test id and call index."""
    import zlib
    forced = bundle if bundle is not None else os.environ.get("PHY_OPTION_BUNDLE")
    if forced is not None and forced != "":
        i = int(forced) if str(forced).isdigit() else BUNDLE_NAMES.index(forced)
        return BUNDLES[i]
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    k = _calls.get(test, 0)
    _calls[test] = k + 1
    return BUNDLES[(zlib.crc32(test.encode()) + k) % len(BUNDLES)]


def check_process(ctx, gs, ref, chunk=0, kmer=0, backend=0, complete_deletion=False, threshold=0, filt=None, bundle=None):
    name, opts = pick_bundle(bundle)
    opts = dict(opts)
    if filt is not None:
        opts["filter"] = filt
    tag = f"[bundle {name}: {opts}; chunk {chunk}, kmer {kmer}, backend {backend}, threshold {threshold}, ref {ref}]"
    ctx.set_option("filter", opts.get("filter", 0))
    ctx.set_option("filter_kernel", opts.get("filter_kernel", 0))
    ctx.set_option("pairs_kernel", opts.get("pairs_kernel", 0))
    ctx.set_option("fold_blocks", opts.get("fold_blocks", 0))
    ctx.set_option("spec_blocks", opts.get("spec_blocks", 0))
    ctx.set_option("chunk", chunk)
    ctx.set_option("kmer", kmer)
    ctx.set_option("compare_backend", backend)
    try:
        if opts.get("packed"):
            ctx.set_genomes_packed([api.pack_genome(g) for g in gs])
        else:
            ctx.set_genomes(gs)
        ctx.set_reference(ref, threshold=threshold)
        r = O.Run(gs, ref, threshold=threshold).process(complete_deletion=complete_deletion)
        assert ctx.threshold == r.threshold, tag
        # a subject on which the reference's 6-mer cache holds an over-deep interval (esa.cxx:174-199) is flagged, and
        # the reference's answers on it are reproduced all the same (option "cache_quirk", default on)
        assert ctx.reference_cache_quirk == bool(O.Esa(gs[ref]).cache_quirks()), tag
        ctx.anchor()
        if complete_deletion:
            ctx.complete_delete()
        for j in range(len(gs)):
            assert hom_tuples_gpu(ctx.homologies(j)) == hom_tuples_orc(r.homologies(j)), f"homologies of genome {j} {tag}"
        s, h = ctx.compare()
        so, ho = r.matrix()
        assert (h == ho).all(), f"homologs differ {tag}"
        assert (s == so).all(), f"substitutions differ {tag}"
        if opts.get("recheck"):
            # phase A once more with every step sent to the wavefront's slow resolver: the lists must come out the same
            ctx.set_option("lean_force_slow", 1)
            try:
                ctx.anchor()
                if complete_deletion:
                    ctx.complete_delete()
                for j in range(len(gs)):
                    assert hom_tuples_gpu(ctx.homologies(j)) == hom_tuples_orc(r.homologies(j)), \
                        f"homologies of genome {j} (slow resolver) {tag}"
            finally:
                ctx.set_option("lean_force_slow", 0)
        for i in range(len(gs)):
            for j in range(len(gs)):
                a, b = api.estimate("jc", s[i, j], h[i, j]), O.estimate("jc", so[i, j], ho[i, j])
                assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-12, tag
    finally:
        for key in ("chunk", "kmer", "compare_backend", "filter", "filter_kernel", "pairs_kernel", "fold_blocks", "spec_blocks"):
            ctx.set_option(key, 0)
    return s, h


# ── B0: seqcmp / revseqcmp ──
def test_seqcmp_batch_all_lengths_and_alignments(ctx):
    rng = np.random.default_rng(1)
    alpha = np.frombuffer(b"ACGT!", np.uint8)
    a = rng.choice(alpha, 6000, p=[.24, .24, .24, .24, .04])
    b = a.copy()
    idx = rng.random(6000) < 0.2
    b[idx] = rng.choice(alpha, int(idx.sum()))
    ctx.set_genomes([a, b])
    ga, oa, gb, ob, ln, rv, want = [], [], [], [], [], [], []
    # (a round of the batch kernel is 63 chunks = 1008 bytes, a wavefront's pass four rounds: lengths around both)
    for n in list(range(0, 301)) + [511, 512, 513, 991, 992, 993, 1007, 1008, 1009, 1023, 1024, 1025, 2015, 2016, 2017, 4031, 4032, 4033, 4096, 5000]:
        for offa, offb in ((0, 0), (1, 0), (3, 7), (13, 2), (64, 65), (2, 2), (6, 1)):
            if max(offa, offb) + n > 6000:
                continue
            for rev in (0, 1):
                ga.append(0); gb.append(1); oa.append(offa); ob.append(offb); ln.append(n); rv.append(rev)
                x, y = a[offa:offa + n], b[offb:offb + n]
                want.append(O.revseqcmp(x, y, n) if rev else O.seqcmp(x, y, n))
    got = ctx.seqcmp_batch(ga, oa, gb, ob, ln, rv)
    assert got.tolist() == want


def test_seqcmp_batch_few_long_segments_are_split_over_the_wavefronts(ctx):
    """A few long segments: their rounds are dealt out over all wavefronts, four rounds a pass (seqcmp_pass_kernel) — lengths
    around the pass's size, empty segments between others (the rounds' hint walks over them), every alignment, both
    directions, a segment of megabytes beside one of a byte; and one call with a single segment (one seqcmp():
    seqcmp_one_kernel)."""
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"ACGT!", np.uint8)
    L = 3_600_000
    a = rng.choice(alpha, L, p=[.24, .24, .24, .24, .04])
    b = a.copy()
    idx = rng.random(L) < 0.15
    b[idx] = rng.choice(alpha, int(idx.sum()))
    ctx.set_genomes([a, b])
    ga, oa, gb, ob, ln, rv, want = [], [], [], [], [], [], []
    for k, n in enumerate([1, 0, 17, 4095, 4096, 4097, 8191, 12288, 70001, 1_000_003, 0, 2_500_000, 5, 3_599_990]):
        for rev in (0, 1):
            offa, offb = (3 * k + rev) % 11, (7 * k) % 13
            if max(offa, offb) + n > L:
                offa = offb = 0
            ga.append(k % 2); gb.append(1 - k % 2); oa.append(offa); ob.append(offb); ln.append(n); rv.append(rev)
            x, y = (a, b)[k % 2][offa:offa + n], (a, b)[1 - k % 2][offb:offb + n]
            want.append(O.revseqcmp(x, y, n) if rev else O.seqcmp(x, y, n))
    got = ctx.seqcmp_batch(ga, oa, gb, ob, ln, rv)
    assert got.tolist() == want
    for rev in (0, 1):
        got = ctx.seqcmp_batch([0], [5], [1], [2], [L - 5], [rev])
        assert got.tolist() == [(O.revseqcmp if rev else O.seqcmp)(a[5:], b[2:L - 3], L - 5)]


def test_seqcmp_batch_of_ragged_short_segments(ctx):
    """The shape evo_model::account produces (src/evo_model.cxx:55): thousands of segments of 0.1-10 kbp at any byte offset,
    a tenth reversed, empty ones among them, '!' in the strings — a wavefront's pass holds rounds of up to four different
    segments, every lane's 16 bytes are cut out of dword-aligned loads and its neighbour's; every tally against the oracle."""
    rng = np.random.default_rng(12)
    alpha = np.frombuffer(b"ACGT!", np.uint8)
    L = 400_000
    gs = [rng.choice(alpha, L, p=[.24, .24, .24, .24, .04]) for _ in range(3)]
    gs[1][::7] = gs[0][::7]
    gs[2] = synth.revcomp(gs[0])  # (so that reversed segments meet complements)
    ctx.set_genomes(gs)
    NS = 6000
    ln = np.minimum(10000, np.maximum(0, rng.exponential(2600, NS))).astype(np.int64)
    ln[rng.random(NS) < 0.03] = 0
    short = rng.random(NS) < 0.05
    ln[short] = rng.integers(1, 40, int(short.sum()))
    ga, gb = rng.integers(0, 3, NS), rng.integers(0, 3, NS)
    oa = (rng.random(NS) * (L - ln)).astype(np.int64)
    ob = (rng.random(NS) * (L - ln)).astype(np.int64)
    rv = (rng.random(NS) < 0.1).astype(np.uint8)
    got = ctx.seqcmp_batch(ga, oa, gb, ob, ln, rv)
    for s in range(NS):
        x, y = gs[ga[s]][oa[s]:oa[s] + ln[s]], gs[gb[s]][ob[s]:ob[s] + ln[s]]
        want = O.revseqcmp(x, y, int(ln[s])) if rv[s] else O.seqcmp(x, y, int(ln[s]))
        assert int(got[s]) == want, (s, int(ln[s]), int(oa[s]) % 4, int(ob[s]) % 4, int(rv[s]))


def test_b0_reference_signatures(ctx):
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, 100000, dtype=np.uint8)  # arbitrary bytes, like any char*
    b = a.copy()
    b[rng.random(100000) < 0.3] = 7
    for n in (0, 1, 15, 16, 17, 4096, 8193, 99999, 100000):
        assert api.seqcmp(a, b, n) == O.seqcmp(a, b, n)
        assert api.revseqcmp(a, b, n) == O.revseqcmp(a, b, n)
    # unaligned starts, as a caller's char* may be
    assert api.seqcmp(a[3:], b[7:], 90001) == O.seqcmp(a[3:], b[7:], 90001)
    assert api.revseqcmp(a[5:], b[1:], 90001) == O.revseqcmp(a[5:], b[1:], 90001)


# ── the path ──
@pytest.mark.parametrize("chunk", [0, 64, 192, 256, 320])
def test_process_star(ctx, chunk):
    gs = synth.make_genomes(6, 30000, seed=3, d_range=(0.01, 0.25))
    check_process(ctx, gs, 0, chunk=chunk)
    check_process(ctx, gs, 4, chunk=chunk, backend=1)


def _five_sets():
    """Five small sets that between them reach every branch of the path: substitutions only; indels, inversions and
    contigs; a tree at low divergence; '!' inside homologies on both strands; near-identical genomes (overruns)."""
    rng = np.random.default_rng(31)
    base = synth.random_base(30000, rng)
    bang = [base.copy()]
    for g in range(4):
        x = synth.mutate(base, 0.02, rng)
        x = (synth.revcomp(x) if g % 2 else x).copy()
        x[rng.integers(1000, 29000, 6)] = ord("!")
        bang.append(x)
    a = synth.random_base(40000, rng)
    b = a.copy()
    b[[7000, 7001, 23000]] = synth.random_base(3, rng)
    near = [a, a.copy(), b, np.concatenate([a[:15000], a[15010:]]), synth.mutate(a, 0.0003, rng)]
    return [(synth.make_genomes(6, 30000, seed=3, d_range=(0.01, 0.25)), 0),
            (synth.make_genomes(7, 40000, seed=5, d_range=(0.01, 0.3), indel_per_mbp=500, inv_frac=0.1, contigs=3, inv_len=(100, 1500)), 5),
            (synth.make_genomes(9, 60000, seed=9, d_range=(0.0005, 0.03), tree=True, indel_per_mbp=200, inv_frac=0.05), 7),
            (bang, 0), (near, 4)]


def test_lanes_that_take_chunk_after_chunk(ctx):
    """Two blocks of the speculative chain kernel (option spec_blocks) for thousands of chunks: every lane works through
    chunk after chunk, claiming the next one an eighth of a chunk ahead — its work item and its query's descriptor come
    in beside the trips' loads (lean_kernels.hip) — from chunks of a few steps, at whose end the claim is still on its
    way, to chunks of thousands of positions; contigs, so that where the query's list of '!' stands travels with the item."""
    gs = synth.make_genomes(6, 150000, seed=77, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.05, contigs=5,
                            inv_len=(300, 3000))
    for chunk in (64, 256, 1024, 4096):
        check_process(ctx, gs, 0, chunk=chunk, bundle="few-chain-blocks")
    check_process(ctx, gs, 3, chunk=512, bundle="few-chain-blocks")


@pytest.mark.parametrize("bundle", ["default", "host-filter+valu-pairs", "general-filter-kernel", "few-chain-blocks"])
def test_named_bundles_over_the_same_five_sets(ctx, bundle):
    """The library's default configuration and two named non-default bundles over the same five input sets: whatever
    the per-test rotation picks elsewhere, these combinations are always exercised on these inputs."""
    for gs, ref in _five_sets():
        check_process(ctx, gs, ref, bundle=bundle)
        check_process(ctx, gs, ref, chunk=128, bundle=bundle)


@pytest.mark.parametrize("seed", [5, 6])
def test_process_indels_inversions_contigs(ctx, seed):
    gs = synth.make_genomes(7, 40000, seed=seed, d_range=(0.01, 0.3), indel_per_mbp=500, inv_frac=0.1,
                            contigs=3, inv_len=(100, 1500))
    for ref in (0, 5):
        s, h = check_process(ctx, gs, ref, chunk=128)
        s1, h1 = check_process(ctx, gs, ref, chunk=64, kmer=3, backend=1)
        assert (s == s1).all() and (h == h1).all()


def test_process_tree_low_divergence(ctx):
    gs = synth.make_genomes(9, 80000, seed=9, d_range=(0.0005, 0.03), tree=True, indel_per_mbp=200, inv_frac=0.05)
    check_process(ctx, gs, 0)
    check_process(ctx, gs, 7, chunk=128)


def test_process_bang_inside_homology(ctx):
    """A contig break inside a homologous stretch: '!' is projected, so the
    five-plane pair kernel has to run (seqcmp: '!'!='A'; revseqcmp: '!' acts as 'A')."""
    rng = np.random.default_rng(31)
    base = synth.random_base(30000, rng)
    gs = [base.copy()]
    for g in range(4):
        s = synth.mutate(base, 0.02, rng)
        if g % 2:
            s = synth.revcomp(s)
        s = s.copy()
        for pos in rng.integers(1000, 29000, 6):
            s[pos] = ord("!")
        gs.append(s)
    ref = base.copy()
    ref[15000] = ord("!")
    gs.append(ref)
    s, h = check_process(ctx, gs, 0, chunk=128)
    check_process(ctx, gs, 5, chunk=128)
    s1, h1 = check_process(ctx, gs, 0, backend=1)
    assert (s == s1).all() and (h == h1).all()
    assert ctx.stat("pileup:bang") is not None


def test_process_edge_cases(ctx):
    rng = np.random.default_rng(24)
    a = synth.random_base(5000, rng)
    gs = [a, a.copy(), a[:5].copy(), a[10:11].copy(), np.zeros(0, np.uint8), a[200:230].copy(),
          np.frombuffer(b"!!!!", np.uint8).copy(), np.frombuffer(b"ACGT!ACGT", np.uint8).copy(),
          synth.random_base(3000, rng), synth.revcomp(a)]
    check_process(ctx, gs, 0, chunk=64)
    check_process(ctx, gs, 1)


def test_process_complete_deletion(ctx):
    gs = synth.make_genomes(5, 30000, seed=12, d_range=(0.01, 0.1), indel_per_mbp=600, inv_frac=0.05)
    check_process(ctx, gs, 1, complete_deletion=True)


@pytest.mark.parametrize("chunk", [64, 256, 0])
def test_process_near_identical_genomes(ctx, chunk):
    """Genomes equal to the reference over many chunk lengths (copies, three substitutions in 40 kbp, a
    deletion, a 25 kbp inversion, '!' inside identical sequence): the speculative chains cut their
    comparisons one chunk length past the chunk's end and the ends are resolved afterwards
    (lean_core.h: overruns); lists and tallies equal the oracle's."""
    rng = np.random.default_rng(31)
    a = synth.random_base(40000, rng)
    b = a.copy()
    b[[7000, 7001, 23000]] = synth.random_base(3, rng)
    c = np.concatenate([a[:15000], a[15010:]])
    d = np.concatenate([a[:5000], synth.revcomp(a[5000:30000]), a[30000:]])
    e = synth.split_contigs(a.copy(), 3, rng)
    gs = [a, a.copy(), b, c, d, e, synth.mutate(a, 0.0003, rng)]
    check_process(ctx, gs, 0, chunk=chunk)
    check_process(ctx, gs, 6, chunk=chunk)
    rep = synth.random_base(6000, rng)
    f = np.concatenate([synth.random_base(9000, rng), rep, synth.random_base(4000, rng), rep, synth.random_base(7000, rng), rep,
                        synth.random_base(3000, rng)])
    g = f.copy()
    g[[12000, 20500, 31000]] = synth.random_base(3, rng)
    check_process(ctx, [f, g, f.copy()], 1, chunk=chunk or 512)


def test_device_filter_on_entangled_lists(ctx):
    """Reference with a 3 kbp stretch present eight times and queries that carry diverged copies of it: the
    raw lists hold many overlapping homologies, i.e. long entangled stretches for the stretch-wise chain
    filter (some beyond what one thread takes: handed to the general kernel on the device)."""
    rng = np.random.default_rng(41)
    rep = synth.random_base(3000, rng)
    parts = []
    for i in range(8):
        parts += [synth.random_base(int(rng.integers(500, 4000)), rng), synth.mutate(rep, 0.01 * i, rng)]
    a = np.concatenate(parts + [synth.random_base(2000, rng)])
    gs = [a] + [synth.mutate(a, d, rng) for d in (0.005, 0.02, 0.05, 0.1)]
    for bundle in ("device-filter+valu-pairs", "general-filter-kernel"):  # filter_kernel 0 and 1, both on the device
        for ref in (0, 2):
            check_process(ctx, gs, ref, bundle=bundle)


def test_installed_lists_that_overlap_go_through_the_segment_backend(ctx):
    """compare() of process.cxx:566-611 takes any two lists; the pileup equals it only for sorted, disjoint
    ones.  Lists installed by the caller that overlap are tallied by the segment backend (same numbers as
    the oracle's merge-join on those lists); a device buffer of such lists is refused by the attach call."""
    gs = synth.make_genomes(4, 20000, seed=51, d_range=(0.02, 0.15), inv_frac=0.05, inv_len=(200, 900))
    ctx.set_genomes(gs)
    ctx.set_reference(0)
    ctx.anchor()
    lists = [np.array(ctx.homologies(j)) for j in range(4)]
    # genome 2: its third block once more, shifted by 5 positions on the reference and in the query — overlapping its original
    extra = lists[2][2:3].copy()
    for f in ("index_reference", "index_reference_projected", "index_query"):
        extra[f] += 5
    extra["length"] -= 10
    bad = np.concatenate([lists[2][:3], extra, lists[2][3:]])
    ctx.set_homologies(2, bad)
    s, h = ctx.compare()
    ctx.set_option("compare_backend", 1)
    s1, h1 = ctx.compare()
    ctx.set_option("compare_backend", 0)
    assert (s == s1).all() and (h == h1).all()

    def conv(x):
        o = np.zeros(len(x), O.HOM_DTYPE)
        o["rev"], o["iref"], o["iproj"] = x["direction"], x["index_reference"], x["index_reference_projected"]
        o["iq"], o["len"] = x["index_query"], x["length"]
        return o
    use = [lists[0], lists[1], bad, lists[3]]
    for i in range(4):
        for j in range(i + 1, 4):
            ss, hh = O.compare_lists(gs[i], conv(use[i]), gs[j], conv(use[j]))
            assert (int(s[i, j]), int(h[i, j])) == (ss, hh), (i, j)
    assert ctx.stat("count:compare_calls_rerouted_to_segments") >= 1
    # the same lists as a device buffer: refused
    import torch
    counts = np.array([len(x) for x in use], np.uint64)
    flat = np.zeros(int(counts.sum()), api.PACKED)
    allh = np.concatenate(use)
    flat["start"], flat["index_query"] = allh["index_reference_projected"], allh["index_query"]
    flat["length"], flat["direction"] = allh["length"], allh["direction"]
    t = torch.from_numpy(flat.view(np.uint8)).to("cuda:0")
    begin = np.concatenate(([0], np.cumsum(counts[:-1]))).astype(np.uint64)
    with pytest.raises(api.PhyloniumError):
        ctx.attach_packed_device(t.data_ptr(), begin, counts, 0, 0)


@pytest.mark.timeout(600)
def test_lists_beyond_the_filter_blocks_lds_stay_on_the_device(ctx):
    """Queries of 16 Mbp with 400 indels per Mbp leave more than 4096 raw homologies each: sorted by the long-list
    kernel in global memory (tiles in LDS), filtered stretch by stretch, no host step; lists equal the oracle's."""
    gs = synth.make_genomes(3, 16_000_000, seed=61, d_range=(0.03, 0.12), indel_per_mbp=400, inv_frac=0.03, contigs=3,
                            inv_len=(500, 4000))
    ctx.set_option("filter", 2)
    try:
        ctx.set_genomes(gs)
        refb = bytes(gs[0])
        sa = api.host_suffix_array(refb + b"#" + O.revcomp(refb))
        ctx.set_reference(0, sa=sa)
        ctx.reset_stats()
        ctx.anchor()
        assert not ctx.stat("count:queries_left_to_the_host")
        r = O.Run(gs, 0).process(sa=sa, threads=8)
        assert ctx.threshold == r.threshold
        for j in range(3):
            got, want = hom_tuples_gpu(ctx.homologies(j)), hom_tuples_orc(r.homologies(j))
            assert got == want, j
            if j:
                assert r.homologies(j, filtered=False).size > 4096
        s, h = ctx.compare()
        so, ho = r.matrix()
        assert (s == so).all() and (h == ho).all()
    finally:
        ctx.set_option("filter", 0)


def test_reference_cache_quirk_is_reported(ctx):
    """A reference on which phylonium's 6-mer cache stores an over-deep interval (the only two occurrences of
    a short nucleotide string stand in front of a contig join, esa.cxx:174-199): flagged; an ordinary
    multi-contig reference: not flagged."""
    rng = np.random.default_rng(12)
    bad = np.frombuffer(b"CCGT!AAAAGT!CCCC", np.uint8)  # "GT" occurs twice, both times in front of a contig join
    assert O.Esa(bad).cache_quirks() > 0
    ctx.set_genomes([bad, bad.copy()])
    ctx.set_reference(0)
    assert ctx.reference_cache_quirk
    # ... and what the reference answers there is reproduced: queries that walk into the over-deep keys
    q1 = np.frombuffer(b"CCGTAAAAAGTACCCCGTCAAAGTTCCCC" * 3, np.uint8)
    q2 = np.frombuffer(b"GTAGTCGTGGTTGTAAGTACGTCC" * 4, np.uint8)
    for thr in (0, 4, 6):
        check_process(ctx, [bad, q1, q2, synth.revcomp(q1)], 0, threshold=thr)
    # with the option off the product gives the true longest matches: the raw lists differ from the reference's
    gs = [bad, q1, q2, synth.revcomp(q1)]
    r = O.Run(gs, 0, threshold=4).process()
    ctx.set_option("cache_quirk", 0)
    try:
        ctx.set_genomes(gs)
        ctx.set_reference(0, threshold=4)
        ctx.anchor()
        assert any(hom_tuples_gpu(ctx.homologies(j)) != hom_tuples_orc(r.homologies(j)) for j in range(4))
    finally:
        ctx.set_option("cache_quirk", 1)
    good = synth.split_contigs(synth.random_base(30000, rng), 4, rng)
    assert O.Esa(good).cache_quirks() == 0
    ctx.set_genomes([good, synth.mutate(good, 0.05, rng)])
    ctx.set_reference(0)
    assert not ctx.reference_cache_quirk


def test_process_repeats(ctx):
    rng = np.random.default_rng(23)
    unit = synth.random_base(700, rng)
    sp = [synth.random_base(1500, rng) for _ in range(5)]
    a = np.concatenate([sp[0], unit, sp[1], unit, sp[2], synth.revcomp(unit), sp[3],
                        np.frombuffer(b"A" * 300 + b"AC" * 200 + b"T" * 100, np.uint8), sp[4]])
    b = synth.mutate(a, 0.02, rng)
    c = synth.mutate(np.concatenate([sp[2], unit, unit, sp[0], np.frombuffer(b"A" * 500, np.uint8)]), 0.01, rng)
    for ref in (0, 1, 2):
        check_process(ctx, [a, b, c], ref, chunk=64)


def test_both_phases_as_one_call(ctx):
    """phylo_anchor_compare: phase B queued behind phase A, phase A's flags read with the result.  Same tallies and
    lists as the two calls and as the oracle — on plain sets, on sets whose lists go to the host after all (exact
    repeats: homologies with equal projected starts; the call then goes the long way round), with '!' in the
    genomes, and again after the options that change where lists live."""
    rng = np.random.default_rng(77)
    unit = synth.random_base(700, rng)
    sp = [synth.random_base(1500, rng) for _ in range(5)]
    a = np.concatenate([sp[0], unit, sp[1], unit, sp[2], synth.revcomp(unit), sp[3], sp[4]])
    repeats = [a, synth.mutate(a, 0.02, rng), np.concatenate([sp[2], unit, unit, sp[0]])]
    # a query that carries stretches of the reference once forward and once reverse-complemented between random
    # flanks: the two homologies of a stretch project onto the same reference interval — equal starts, the host's case
    # (a homology begins where a step of the chain lands, a base or a few into the stretch: with 38 stretches some pairs
    # begin at the same base — asserted below on the oracle's raw list)
    ref40 = synth.random_base(40000, rng)
    pieces = []
    for x in range(1000, 39000, 1000):
        seg = ref40[x:x + 500]
        pieces += [synth.random_base(100, rng), seg, synth.random_base(100, rng), synth.revcomp(seg)]
    ties = [ref40, np.concatenate(pieces + [synth.random_base(200, rng)]), synth.mutate(ref40, 0.03, rng)]
    starts = [int(x["iproj"]) for x in O.Run(ties, 0).process(compare=False).homologies(1, filtered=False)]
    assert len(set(starts)) < len(starts)
    sets = [(synth.make_genomes(6, 30000, seed=12, d_range=(0.01, 0.25), indel_per_mbp=500, inv_frac=0.05, contigs=3), 2),
            (repeats, 0), (repeats, 2), (ties, 0),
            (synth.make_genomes(70, 5000, seed=13, d_range=(0.01, 0.2)), 5)]
    repeated = 0
    for gs, ref in sets:
        r = O.Run(gs, ref).process(threads=4)
        so, ho = r.matrix()
        ctx.set_option("filter", 0)
        ctx.set_genomes(gs)
        ctx.set_reference(ref)
        for attempt in range(2):
            before = ctx.stat("count:anchor_compare_calls_repeated") or 0
            s, h = ctx.anchor_compare()
            repeated += (ctx.stat("count:anchor_compare_calls_repeated") or 0) - before
            assert (h == ho).all() and (s == so).all(), (ref, attempt)
            for j in range(min(len(gs), 8)):
                assert hom_tuples_gpu(ctx.homologies(j)) == hom_tuples_orc(r.homologies(j)), (ref, attempt, j)
        ctx.anchor()  # the two calls after the one: nothing of the deferred state is left behind
        s, h = ctx.compare()
        assert (h == ho).all() and (s == so).all()
        ctx.set_option("filter", 1)  # lists through the host: the one call has nothing to defer
        s, h = ctx.anchor_compare()
        assert (h == ho).all() and (s == so).all()
        ctx.set_option("filter", 0)
    assert repeated >= 1  # (the set with equal starts)
    # the same on a stream of the caller's (phylo_ctx_set_stream): a context of its own, a side stream of torch's
    import torch
    side = torch.cuda.Stream(device=0)
    c2 = api.Context(0)
    try:
        c2.set_stream(side.cuda_stream)
        gs, ref = sets[0]
        c2.set_genomes(gs)
        c2.set_reference(ref)
        so, ho = O.Run(gs, ref).process(threads=4).matrix()
        for _ in range(3):
            s, h = c2.anchor_compare()
            assert (h == ho).all() and (s == so).all()
    finally:
        c2.close()
    s, h = ctx.process(ref_idx=1)  # phylo_process takes the same road
    r = O.Run(sets[-1][0], 1).process(threads=4)
    so, ho = r.matrix()
    assert (h == ho).all() and (s == so).all()


_CORRUPT_SA_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from phylonium_amd import api, synth
gs = synth.make_genomes(6, 30000, seed=61, d_range=(0.01, 0.2), indel_per_mbp=300, inv_frac=0.05, contigs=2)
with api.Context(0) as ctx:
    ctx.set_genomes(gs)
    s, h = ctx.process(2)
    print("rejected", int(ctx.stat("ref:sa_device_rejected", 0)), "on_device", int(ctx.stat("ref:sa_on_device", 0)))
    np.save(sys.argv[1], np.stack([s, h]))
"""


def test_a_device_built_suffix_array_that_is_wrong_is_caught_and_rebuilt(ctx, tmp_path):
    """The LCP kernel compares every suffix with its successor, which proves the array it is given (index_kernels.hip): a
    caller's array that is not the suffix array of S is refused; the device builder's own, damaged on purpose (development
    build, PHYLONIUM_AMD_TEST_CORRUPT_SA: two entries overwritten, one of them far outside S), is thrown away, built again on
    the host cores, and the result is the oracle's — no fault, no wrong matrix."""
    import subprocess
    gs = synth.make_genomes(6, 30000, seed=61, d_range=(0.01, 0.2), indel_per_mbp=300, inv_frac=0.05, contigs=2)
    so, ho = O.Run(gs, 2).process().matrix()
    refb = bytes(gs[2])
    sa = api.host_suffix_array(refb + b"#" + O.revcomp(refb)).copy()
    ctx.set_genomes(gs)
    ctx.set_reference(2, sa=sa)  # the right array: accepted
    sa[1000], sa[1001] = sa[1001], sa[1000]
    with pytest.raises(api.PhyloniumError, match="not the suffix array"):
        ctx.set_reference(2, sa=sa)
    sa[1000] = 2 ** 31  # an entry outside S: refused, not followed
    with pytest.raises(api.PhyloniumError, match="not the suffix array|out of range"):
        ctx.set_reference(2, sa=sa)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "m.npy")
    env = dict(os.environ, PHYLONIUM_AMD_LIB=os.path.join(root, "phylonium_amd", "libphylonium_amd_dev.so"), PHYLONIUM_AMD_TEST_CORRUPT_SA="1")
    r = subprocess.run([sys.executable, "-c", _CORRUPT_SA_SCRIPT % root, out], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rejected 1 on_device 0" in r.stdout, (r.stdout, r.stderr[-2000:])
    m = np.load(out)
    assert (m[0] == so).all() and (m[1] == ho).all()


def test_external_suffix_array_is_accepted(ctx):
    gs = synth.make_genomes(3, 20000, seed=41, d_range=(0.02, 0.1))
    S = gs[1].tobytes() + b"#" + O.revcomp(gs[1].tobytes())
    ctx.set_genomes(gs)
    ctx.set_reference(1, sa=O.suffix_array(S))  # what divsufsort64 would hand over
    ctx.anchor()
    s, h = ctx.compare()
    so, ho = O.Run(gs, 1).process().matrix()
    assert (s == so).all() and (h == ho).all()


def test_sharded_compare_sums_to_full(ctx):
    gs = synth.make_genomes(70, 6000, seed=43, d_range=(0.01, 0.2))
    ctx.set_genomes(gs)
    ctx.set_reference(3)
    ctx.anchor(0, 30)
    ctx.anchor(30, 70)
    s, h = ctx.compare()
    acc_s, acc_h = np.zeros_like(s), np.zeros_like(h)
    for part in range(3):
        ps, ph = ctx.compare(part, 3)
        acc_s += ps
        acc_h += ph
    assert (acc_s == s).all() and (acc_h == h).all()
    so, ho = O.Run(gs, 3).process(threads=4).matrix()
    assert (s == so).all() and (h == ho).all()


# ── golden fixtures: known answers of the compiled reference ──
def test_golden_simple_and_cfg1(ctx, golden_dir):
    known = json.load(open(os.path.join(golden_dir, "known_answers.json")))
    g = [O.read_fasta_genome(os.path.join(golden_dir, f"simple{i}.fasta.gz")) for i in (0, 1)]
    ctx.set_genomes(g)
    s, h = ctx.process(1)
    assert api.format_phylip(["simple0", "simple1"], s, h) == \
        "2\nsimple0  0.0000e+00  9.7004e-02\nsimple1  9.7004e-02  0.0000e+00\n"
    k = known["cfg1"]
    g = [O.read_fasta_genome(os.path.join(golden_dir, f"cfg1_{i}.fasta.gz")) for i in (0, 1)]
    ctx.set_genomes(g)
    s, h = ctx.process(k["ref"])
    assert ctx.threshold == k["threshold"]
    hv = ctx.homologies(0)
    assert len(hv) == k["n_homologies_q0"] and int(hv["length"].sum()) == k["covered_q0"]
    assert [[int(x["index_reference"]), int(x["index_query"]), int(x["length"])] for x in hv[:3]] == \
        k["first_homologies_q0"]
    assert int(s[0, 1]) == k["substitutions"] and int(h[0, 1]) == k["homologs"]
    assert "%.17g" % api.estimate("jc", s[0, 1], h[0, 1]) == k["jc"]
    names = ["cfg1_0", "cfg1_1"]
    assert api.format_phylip(names, s, h, "jc").split()[3] == k["phylip_jc"]
    assert api.format_phylip(names, s, h, "raw").split()[3] == k["phylip_raw"]
    assert api.format_phylip(names, s, h, "ani").split()[3] == k["phylip_ani"]


def test_medium_scale_parity_and_properties(ctx):
    """24 x 400 kbp with structure: oracle-checked; plus size-independent
    properties (symmetry, homologs <= min genome length, self-row coverage)."""
    gs = synth.make_genomes(24, 400000, seed=51, d_range=(0.01, 0.3), indel_per_mbp=100, inv_frac=0.02)
    ctx.set_genomes(gs)
    s, h = ctx.process(11)
    r = O.Run(gs, 11).process(threads=8)
    so, ho = r.matrix()
    assert (s == so).all() and (h == ho).all()
    assert (s == s.T).all() and (h == h.T).all() and (s <= h).all()
    assert (np.diag(h) == 0).all()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("threshold", [16, 17, 18, 24, 33, 40])
def test_thresholds_of_long_references(ctx, threshold):
    """min_anchor_length grows with the reference (13-14 at 1 Mbp, 17 from ~60 Mbp on). Beyond 16 the
    lucky-anchor check no longer fits the 16 bytes of the STEP phase and continues in EXT; a failed
    check there still owes the k-mer lookup (the kernel once had no phase for that and hung)."""
    gs = synth.make_genomes(5, 40000, seed=threshold, d_range=(0.01, 0.15), indel_per_mbp=300, inv_frac=0.06, contigs=2)
    check_process(ctx, gs, 0, threshold=threshold)
    check_process(ctx, gs, 3, chunk=128, threshold=threshold, backend=1)


def test_export_import_roundtrip(ctx):
    gs = synth.make_genomes(9, 15000, seed=61, d_range=(0.01, 0.2), inv_frac=0.05)
    ctx.set_genomes(gs)
    ctx.set_reference(4)
    ctx.anchor()
    s, h = ctx.compare()
    counts, flat = ctx.export_homologies(2, 7)
    assert [int(c) for c in counts] == [len(ctx.homologies(j)) for j in range(2, 7)]
    for j in range(2, 7):
        ctx.set_homologies(j, np.zeros(0, api.PHOM))
    ctx.import_homologies(2, 7, counts, flat)
    s2, h2 = ctx.compare()
    assert (s == s2).all() and (h == h2).all()


def test_device_suffix_array_equals_the_host_one(ctx):
    """The suffix array built on the device (prefix doubling over radix sorts, csrc/sa_kernels.hip) is THE suffix array
    of S — equal, entry by entry, to the host builder's (itself equal to the oracle's, tests/test_abi_cpu.py): random
    sequence, contigs ('!'), inversions, 70 kbp copies, a run of one letter, two-letter periods, tiny inputs."""
    rng = np.random.default_rng(11)
    a = synth.random_base(70000, rng)
    cases = [synth.random_base(300000, rng),
             synth.make_genomes(2, 60000, seed=5, d_range=(0.05, 0.1), inv_frac=0.1, contigs=6)[1],
             np.concatenate([a, synth.random_base(5000, rng), a, synth.random_base(100, rng), a[:30000]]),
             np.frombuffer(b"A" * 40000, np.uint8), np.frombuffer(b"AC" * 20000 + b"!" + b"CA" * 3000, np.uint8),
             np.frombuffer(b"ACGT!ACGT!!ACGT", np.uint8), np.frombuffer(b"A", np.uint8), np.frombuffer(b"TTTTTTTTTTTTTTTTTTTTTTTTTTTTTT", np.uint8),
             np.concatenate([synth.random_base(20, rng)] * 2000)]
    for k, g in enumerate(cases):
        ctx.set_genomes([g, synth.random_base(100, rng)])
        ctx.set_option("sa_builder", 1)
        ctx.set_reference(0)
        st = ctx.stats()
        assert st["ref:sa_on_device"] == 1
        dev = ctx.reference_suffix_array()
        refb = bytes(np.asarray(g, np.uint8))
        want = api.host_suffix_array(refb + b"#" + O.revcomp(refb))
        assert np.array_equal(dev, want), f"case {k}"
        if k in (2, 3, 8):
            assert st["ref:sa_rounds"] >= 5  # the repeats kept the doubling going
        ctx.set_option("sa_builder", 0)
        ctx.set_reference(0)
        assert ctx.stats()["ref:sa_on_device"] == 0 and np.array_equal(ctx.reference_suffix_array(), want)
    # many small strings over two letters (every suffix tied with others for many rounds), with and without contigs
    for seed in range(120):
        r = np.random.default_rng(1000 + seed)
        n = int(r.integers(1, 400))
        g = np.frombuffer(b"AC", np.uint8)[r.integers(0, 2, n)].copy()
        if seed % 3 == 0 and n > 4:
            g[r.integers(0, n, 2)] = ord("!")
        if seed % 5 == 0:
            g = np.tile(g, 3)
        ctx.set_genomes([g, synth.random_base(40, r)])
        ctx.set_option("sa_builder", 1)
        ctx.set_reference(0)
        gb = bytes(g)
        assert ctx.stats()["ref:sa_on_device"] == 1
        assert np.array_equal(ctx.reference_suffix_array(), api.host_suffix_array(gb + b"#" + O.revcomp(gb))), f"seed {seed}"
    ctx.set_option("sa_builder", 1)
    # a byte the packing has no code for: the host builders take over, the array is still the array
    odd = np.frombuffer(b"ACGTNNACGTACGTTTGACA", np.uint8)
    ctx.set_genomes([odd, synth.random_base(50, rng)])
    ctx.set_reference(0)
    ob = bytes(odd)
    assert ctx.stats()["ref:sa_on_device"] == 0
    assert np.array_equal(ctx.reference_suffix_array(), api.host_suffix_array(ob + b"#" + O.revcomp(ob)))


def test_packed_genomes_are_the_same_genomes(ctx):
    """phylo_set_genomes_packed: Q2 copied into place, the byte arena written by the device.  The genomes read back
    byte for byte and both phases give what they give after phylo_set_genomes — for lengths around the 16-base
    words, separators at either end and next to each other, and an empty genome."""
    rng = np.random.default_rng(3)
    gs = [synth.random_base(n, rng) for n in (1, 15, 16, 17, 63, 64, 65, 1000, 4097, 70001)]
    gs += [np.frombuffer(b"!ACGT!!TTGA!", np.uint8), np.zeros(0, np.uint8), np.frombuffer(b"!", np.uint8)]
    big = synth.make_genomes(4, 50000, seed=12, d_range=(0.02, 0.1), inv_frac=0.05, contigs=5)
    gs += big
    ctx.set_genomes_packed([api.pack_genome(g) for g in gs])
    for j, g in enumerate(gs):
        assert np.array_equal(ctx.get_genome(j), np.asarray(g, np.uint8)), j
    ref = len(gs) - 3
    ctx.set_reference(ref)
    ctx.anchor()
    lists = [hom_tuples_gpu(ctx.homologies(j)) for j in range(len(gs))]
    s1, h1 = ctx.compare()
    ctx.set_genomes(gs)
    for j, g in enumerate(gs):
        assert np.array_equal(ctx.get_genome(j), np.asarray(g, np.uint8)), j
    ctx.set_reference(ref)
    ctx.anchor()
    assert lists == [hom_tuples_gpu(ctx.homologies(j)) for j in range(len(gs))]
    s2, h2 = ctx.compare()
    assert (s1 == s2).all() and (h1 == h2).all() and h1[ref, -1] > 30000
    r = O.Run(big, 1).process()  # the four related genomes alone, against the oracle
    ctx.set_genomes_packed([api.pack_genome(g) for g in big])
    ctx.set_reference(1)
    ctx.anchor()
    s3, h3 = ctx.compare()
    so, ho = r.matrix()
    assert (s3 == so).all() and (h3 == ho).all()
    with pytest.raises(api.PhyloniumError, match="separator positions"):
        ctx.set_genomes_packed([(np.zeros(1, np.uint32), 4, np.array([4], np.uint32))])
    # stray code bits behind a genome's end and under a separator are the caller's slip, not the kernels' problem
    dirty = []
    for g in big:
        q2, ln, bad = api.pack_genome(g)
        q2 = q2.copy()
        if ln % 16:
            q2[-1] |= np.uint32((1 << (2 * (16 - ln % 16))) - 1)
        for b in bad:
            q2[b >> 4] |= np.uint32(3 << (30 - 2 * (int(b) & 15)))
        dirty.append((q2, ln, bad))
    ctx.set_genomes_packed(dirty)
    for j, g in enumerate(big):
        assert np.array_equal(ctx.get_genome(j), np.asarray(g, np.uint8))
    ctx.set_reference(1)
    ctx.anchor()
    s4, h4 = ctx.compare()
    assert (s4 == so).all() and (h4 == ho).all()


def test_packed_export_import_is_lossless(ctx):
    gs = synth.make_genomes(9, 15000, seed=62, d_range=(0.01, 0.2), inv_frac=0.1)
    ctx.set_genomes(gs)
    ctx.set_reference(3)
    ctx.anchor()
    before = [np.array(ctx.homologies(j)) for j in range(9)]
    assert any((b["direction"] == 1).any() for b in before)
    counts, flat = ctx.export_packed(0, 9)
    assert flat.dtype.itemsize == 16
    for j in range(9):
        ctx.set_homologies(j, np.zeros(0, api.PHOM))
    ctx.import_packed(0, 9, counts, flat)
    for j in range(9):
        after = np.array(ctx.homologies(j))
        for f in ("index_reference", "index_reference_projected", "index_query", "length", "direction"):
            assert (after[f] == before[j][f]).all(), (j, f)


def test_device_resident_exchange_between_two_contexts():
    """The N>1 data flow without the collectives: two contexts stand for two ranks, each
    anchors its half, the exported device records are laid out as an all-gather would,
    attached to both, and the two window-range parts of phase B add up to the oracle's
    matrix. Lists a context did not compute are read back from the attached buffer on demand."""
    import torch
    gs = synth.make_genomes(10, 25000, seed=83, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.08, contigs=2)
    n, ref = len(gs), 2
    so, ho = O.Run(gs, ref).process().matrix()
    dev = torch.device("cuda", 0)
    bounds = [0, 4, n]
    ctxs = [api.Context(0) for _ in range(2)]
    try:
        counts = np.zeros(n, np.uint64)
        for r, c in enumerate(ctxs):
            c.set_option("filter", 1 + r)  # rank 0 filters on the host, rank 1 on the device (export then copies device to device)
            c.set_genomes(gs)
            c.set_reference(ref)
            c.anchor(bounds[r], bounds[r + 1])
            counts[bounds[r]:bounds[r + 1]] = c.hom_counts(bounds[r], bounds[r + 1])
        sizes = [int(counts[bounds[r]:bounds[r + 1]].sum()) for r in range(2)]
        cap = max(sizes)
        gathered = torch.zeros(2 * cap * 16, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for r, c in enumerate(ctxs):
            got = c.export_packed_device(bounds[r], bounds[r + 1], gathered.data_ptr() + r * cap * 16, cap)
            assert (got == counts[bounds[r]:bounds[r + 1]]).all()
        begin = np.zeros(n, np.uint64)
        for r in range(2):
            cc = counts[bounds[r]:bounds[r + 1]]
            begin[bounds[r]:bounds[r + 1]] = r * cap + np.concatenate(([0], np.cumsum(cc[:-1])))
        total = torch.zeros(2 * n * n, dtype=torch.int64, device=dev)
        for r, c in enumerate(ctxs):
            c.attach_packed_device(gathered.data_ptr(), begin, counts, bounds[r], bounds[r + 1])
            t = torch.empty(2 * n * n, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            c.compare_device(r, 2, t.data_ptr(), t.data_ptr() + n * n * 8)
            total += t
        m = total.cpu().numpy().view(np.uint64).reshape(2, n, n)
        assert (m[0] == so).all() and (m[1] == ho).all()
        # lists of the other half come out of the attached buffer, identical to the owner's
        for j in range(n):
            a, b = np.array(ctxs[0].homologies(j)), np.array(ctxs[1].homologies(j))
            assert len(a) == int(counts[j]) and (a == b).all(), j
        # and the host-side comparison agrees once they are there
        s2, h2 = ctxs[0].compare()
        assert (s2 == so).all() and (h2 == ho).all()
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("blocks", [1, 2, 5, 8, 16])
def test_fold_with_several_blocks_per_query(ctx, blocks):
    """Queries of 1.2 Mbp at low divergence leave tens of thousands of anchors in one window of chunks: the fold's
    iterations are split over `blocks` blocks per query, each finding the carry in front of its part (last anchor, its
    right flag, the open run's start) in the logs; homologies leave through an atomic counter.  Long runs of right
    anchors (indel-free stretches of tens of kbp) make the backward search for a run's start cross several rounds.
    Lists and tallies equal the oracle's, with the device filter and with the host's (which restores query order)."""
    gs = synth.make_genomes(4, 1200000, seed=311, d_range=(0.005, 0.08), indel_per_mbp=15, inv_frac=0.03, contigs=2)
    ctx.set_genomes(gs)
    ctx.set_reference(0)
    r = O.Run(gs, 0).process(threads=4)
    so, ho = r.matrix()
    for filt in (2, 1):
        ctx.set_option("filter", filt)
        ctx.set_option("fold_blocks", blocks)
        for chunk in (0, 2048):
            ctx.set_option("chunk", chunk)
            ctx.anchor()
            for j in range(len(gs)):
                assert hom_tuples_gpu(ctx.homologies(j)) == hom_tuples_orc(r.homologies(j)), (filt, chunk, j)
            s, h = ctx.compare()
            assert (s == so).all() and (h == ho).all()
    ctx.set_option("filter", 0)
    ctx.set_option("fold_blocks", 0)
    ctx.set_option("chunk", 0)


def test_block_exchange_between_three_contexts():
    """The exchange without the host in it (phylo_export_block_device / phylo_attach_blocks_device /
    phylo_compare_triangle_device): three contexts stand for three ranks on torch's current stream; each anchors its
    block of queries and writes its exchange block straight into its slot of the buffer an all-gather would fill, all
    attach that buffer, the three u32 triangles add up to the oracle's matrices (one all-reduce), and lists a context
    did not compute come out of the gathered buffer.  Rank 0 filters on the host (its block is put together there),
    a capacity that is too small is reported by every rank, and round 2's exchange still works beside it."""
    import torch
    gs = synth.make_genomes(13, 25000, seed=85, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.08, contigs=2)
    n, ref = len(gs), 5
    r_orc = O.Run(gs, ref).process()
    so, ho = r_orc.matrix()
    dev = torch.device("cuda", 0)
    bounds = [0, 3, 9, n]
    world = 3
    stream = torch.cuda.current_stream(dev).cuda_stream
    ctxs = [api.Context(0) for _ in range(world)]
    try:
        for r, c in enumerate(ctxs):
            c.set_stream(stream)
            c.set_option("filter", 1 if r == 0 else 0)
            c.set_genomes(gs)
            c.set_reference(ref)
        maxq = 8
        for cap, fits in ((200000, True), (40, False), (200000, True)):
            nbytes = ctxs[0].exchange_block_bytes(maxq, cap)
            gathered = torch.zeros(world * nbytes, dtype=torch.uint8, device=dev)
            nw = ctxs[0].triangle_words()
            total = torch.zeros(nw, dtype=torch.int32, device=dev)
            for r, c in enumerate(ctxs):
                c.anchor(bounds[r], bounds[r + 1])
                c.export_block_device(bounds[r], bounds[r + 1], gathered.data_ptr() + r * nbytes, maxq, cap)
            for r, c in enumerate(ctxs):
                c.attach_blocks_device(gathered.data_ptr(), bounds, maxq, cap, bounds[r], bounds[r + 1])
                t = torch.empty(nw, dtype=torch.int32, device=dev)
                c.compare_triangle_device(r, world, t.data_ptr())  # queued: what the part has to report rides behind its tallies
                total += t
            tail = total[-8:].cpu().numpy()  # the parts' reports: {'!' overflow, unsorted list, block overflow, parts, phase A needs the host, 0, 0, 0}
            assert int(tail[3]) == world and not tail[4:].any()
            if not fits:  # every rank saw every block's overflow mark; whoever reads the summed triangle learns of it
                assert int(tail[2]) == world
                with pytest.raises(api.PhyloniumError, match="overflow"):
                    ctxs[1].triangle_to_matrices(total.data_ptr())
                continue
            s, h = ctxs[1].triangle_to_matrices(total.data_ptr())
            assert (s == so).all() and (h == ho).all()
            ctxs[1].set_option("pairs_kernel", 1)  # the vector-ALU kernels look at their flags themselves
            t = torch.empty(nw, dtype=torch.int32, device=dev)
            ctxs[1].compare_triangle_device(1, world, t.data_ptr())
            ctxs[1].set_option("pairs_kernel", 0)
            t2 = torch.empty(nw, dtype=torch.int32, device=dev)
            ctxs[1].compare_triangle_device(1, world, t2.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(t, t2)
            for j in range(n):
                want = hom_tuples_orc(r_orc.homologies(j))
                for c in ctxs:
                    assert hom_tuples_gpu(c.homologies(j)) == want, j
        s2, h2 = ctxs[2].compare()
        assert (s2 == so).all() and (h2 == ho).all()
    finally:
        for c in ctxs:
            c.close()


def test_queued_rank_pass_and_the_results_shared_home():
    """A rank's pass without a host round trip (round 5): phase A of the rank's queries with its exchange block written
    behind it (phylo_anchor_block_device — nothing waited for), the blocks attached, the parts' triangles summed, and every
    rank's device writing ITS rows of the two matrices into one shared page-locked segment (phylo_result_open /
    phylo_triangle_rows_to_result; three contexts stand for three ranks of a node).  Matrices and lists equal the
    oracle's.  A query whose list has tied starts — only the host's std::sort orders those as the reference does — makes
    its rank's block say so: the summed report tells every rank (word 4), and the pass repeated the long way is right.
    A context that is asked for its lists before any block was attached waits for the queued phase A itself."""
    import torch
    gs = synth.make_genomes(13, 25000, seed=86, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.08, contigs=2)
    # a query that carries stretches of the reference once forward and once reverse-complemented between random flanks: the
    # two homologies of a stretch project onto the same reference interval, and with 20 stretches some pairs begin at the
    # same base (asserted on the oracle's raw list) — equal starts, the host's case
    rng = np.random.default_rng(7)
    clean = gs[5][gs[5] != ord("!")]
    pieces = []
    for x in range(1000, 21000, 1000):
        seg = clean[x:x + 500]
        pieces += [synth.random_base(100, rng), seg, synth.random_base(100, rng), synth.revcomp(seg)]
    dup = np.concatenate(pieces + [synth.random_base(200, rng)])
    starts = [int(x["iproj"]) for x in O.Run(gs + [dup], 5).process(compare=False).homologies(13, filtered=False)]
    assert len(set(starts)) < len(starts)
    dev = torch.device("cuda", 0)
    world, bounds_of = 3, lambda n: [0, 3, 9, n]
    # one stream for the three contexts and torch's own work, as a rank's context shares its stream with the collectives: what
    # orders a rank's queued phase A before the others' reads of its block is the stream (handle 0 would mean "the context's own")
    side = torch.cuda.Stream(device=dev)
    stream = side.cuda_stream
    assert stream != 0
    ctxs = [api.Context(0) for _ in range(world)]
    name = "/phylonium_amd_test_%d" % os.getpid()
    try:
      with torch.cuda.stream(side):
        for tied in (False, True):
                g2 = gs + ([dup] if tied else [])
                n, ref = len(g2), 5
                bounds = bounds_of(n)
                r_orc = O.Run(g2, ref).process()
                so, ho = r_orc.matrix()
                for r, c in enumerate(ctxs):
                    c.set_stream(stream)
                    c.set_genomes(g2)
                    c.set_reference(ref)
                ctxs[0].result_open(name, create=True, ranks=world)
                for c in ctxs[1:]:
                    c.result_open(name, create=False, ranks=world)
                ctxs[0].result_unlink()
                maxq, cap = 8, 100000
                nbytes = ctxs[0].exchange_block_bytes(maxq, cap)
                nw = ctxs[0].triangle_words()
                for attempt in ("queued", "long way"):
                    gathered = torch.zeros(world * nbytes, dtype=torch.uint8, device=dev)
                    total = torch.zeros(nw, dtype=torch.int32, device=dev)
                    for r, c in enumerate(ctxs):
                        if attempt == "queued":
                            c.anchor_block_device(bounds[r], bounds[r + 1], gathered.data_ptr() + r * nbytes, maxq, cap)
                        else:
                            c.anchor(bounds[r], bounds[r + 1])
                            c.export_block_device(bounds[r], bounds[r + 1], gathered.data_ptr() + r * nbytes, maxq, cap)
                    for r, c in enumerate(ctxs):
                        c.attach_blocks_device(gathered.data_ptr(), bounds, maxq, cap, bounds[r], bounds[r + 1])
                        t = torch.empty(nw, dtype=torch.int32, device=dev)
                        c.compare_triangle_device(r, world, t.data_ptr())
                        total += t
                    reps = [c.triangle_rows_to_result(total.data_ptr(), n * r // world, n * (r + 1) // world, r, world if r == world - 1 else 0)
                            for r, c in enumerate(ctxs)]
                    assert all((rep == reps[0]).all() for rep in reps) and int(reps[0][3]) == world
                    if tied and attempt == "queued":  # the last rank's block said its phase A needs the host: every part reports it
                        assert int(reps[0][4]) == world
                        with pytest.raises(api.PhyloniumError, match="needs the host"):
                            ctxs[1].triangle_to_matrices(total.data_ptr())
                        continue
                    assert not reps[0][[0, 1, 2, 4]].any(), (tied, attempt, reps[0])
                    for c in ctxs:  # every context maps the same segment
                        s, h = c.result_matrices()
                        assert (s == so).all() and (h == ho).all(), (tied, attempt)
                    for j in range(n):
                        want = hom_tuples_orc(r_orc.homologies(j))
                        for c in ctxs:
                            assert hom_tuples_gpu(c.homologies(j)) == want, (tied, attempt, j)
                    assert ctxs[0].stat("n:anchor_calls_without_a_wait", 0) >= (1 if attempt == "queued" else 0)
                    if not tied:
                        break
                # asked for its lists with the queued phase A still unread and nothing attached: the context settles it itself
                # (with the tied list: by repeating phase A the long way)
                c = ctxs[2]
                blk = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
                c.anchor_block_device(bounds[2], n, blk.data_ptr(), maxq, cap)
                for j in range(bounds[2], n):
                    assert hom_tuples_gpu(c.homologies(j)) == hom_tuples_orc(r_orc.homologies(j)), (tied, j)
                words = blk.cpu().numpy().view(np.uint32)
                assert int(words[3]) == (1 if tied else 0) and int(words[2]) == n - bounds[2]
                # a private home: phylo_triangle_to_matrices writes the matrices phylo_result_matrices hands out directly
                c = ctxs[1]
                c.anchor()
                c.result_open(None, ranks=1)
                views = c.result_matrices()
                views[0][:] = 7
                tri = torch.empty(nw, dtype=torch.int32, device=dev)
                c.compare_triangle_device(0, 1, tri.data_ptr())
                c.reset_stats()
                s, h = c.triangle_to_matrices(tri.data_ptr(), views)
                assert (s == so).all() and (h == ho).all() and c.stat("ms:triangle_zero_copy") is not None
                for c in ctxs:
                    c.result_close()
    finally:
        for c in ctxs:
            c.close()


_ZERO_COPY_SCRIPT = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %r)
from phylonium_amd import api, synth
with api.Context(0) as ctx:
    ctx.set_option("result_zero_copy", 1)
    for n in (380, 363):
        gs = synth.make_genomes(n, 2500, seed=n, d_range=(0.01, 0.2), indel_per_mbp=400, inv_frac=0.05)
        ctx.set_genomes(gs)
        ctx.set_reference(1)
        ctx.anchor()
        so, ho = ctx.compare()
        assert (so == so.T).all() and int(ho.sum()) > 0
        tri = torch.empty(ctx.triangle_words(), dtype=torch.int32, device="cuda:0")
        ctx.compare_triangle_device(0, 1, tri.data_ptr())
        out = (np.zeros((n, n), np.uint64), np.zeros((n, n), np.uint64))
        ctx.reset_stats()
        for rep in range(3):
            out[0][:] = 7
            out[1][:] = 7
            s, h = ctx.triangle_to_matrices(tri.data_ptr(), out)
            assert (s == so).all() and (h == ho).all(), (n, rep)
        assert ctx.stat("ms:triangle_zero_copy") is not None and ctx.stat("ms:triangle_widen") is not None
        ctx.set_option("result_zero_copy", 0)  # lets go of the matrices before they are freed
        ctx.reset_stats()
        s, h = ctx.triangle_to_matrices(tri.data_ptr(), out)
        assert (s == so).all() and (h == ho).all() and ctx.stat("ms:triangle_zero_copy") is None
        ctx.set_option("result_zero_copy", 1)
print("zero-copy ok")
"""


def test_result_matrices_written_by_the_device():
    """phylo_triangle_to_matrices with option result_zero_copy = 1 and host matrices the caller keeps (a megabyte or more
    each): copied and widened on the host the first time, registered and written by the device itself
    (triangle_to_host_kernel) from the second time on — the same matrices either way, equal to phylo_compare's, for an even
    and an odd number of genomes.  In a process of its own: memory that has been registered with the runtime is nothing the
    rest of the suite should inherit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _ZERO_COPY_SCRIPT % root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "zero-copy ok" in r.stdout, r.stderr[-3000:]


@pytest.mark.parametrize("world", [1, 2, 4, 7])
def test_group_of_ranks_gives_one_contexts_result(world):
    """phylo_group_* (csrc/group.hip): one context and one host thread per rank; the genomes' blocks all-gathered,
    phase A by query block, lists exchanged as device blocks, phase B by window range, triangles all-reduced, every rank's
    device writing its rows of the result.
    Matrices and every list equal the oracle's for any number of ranks — more ranks than GPUs share them (device
    copies instead of RCCL) — over repeated passes (the planned block capacity is reused), after a change of
    reference, and when a later pass outgrows the plan."""
    gs = synth.make_genomes(13, 25000, seed=87, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.08, contigs=2)
    gs.append(gs[3].copy())
    n = len(gs)
    with api.Group(world) as g:
        assert g.backend in ("one rank", "rccl", "device-to-device copies")
        g.set_genomes(gs)
        for ref in (5, 0):
            r = O.Run(gs, ref).process()
            so, ho = r.matrix()
            g.set_reference(ref)
            for rep in range(2):
                s, h = g.process()
                assert (s == so).all() and (h == ho).all(), (world, ref, rep)
            c0 = g.rank_context(0)
            for j in range(n):
                assert hom_tuples_gpu(c0.homologies(j)) == hom_tuples_orc(r.homologies(j)), j
        # a reference change drops the plan (another subject, other lists): against an unrelated reference the lists are
        # all but empty, back on a related one they are long again — no pass overflows
        rng = np.random.default_rng(3)
        big = [synth.random_base(90000, rng)]
        big += [synth.mutate(big[0], 0.2, rng) for _ in range(n - 2)] + [synth.random_base(90000, rng)]
        g.set_genomes(big)
        for ref in (n - 1, 0, n - 1, 0):
            s, h = g.process(ref_idx=ref)
            so, ho = O.Run(big, ref).process(threads=4).matrix()
            assert (s == so).all() and (h == ho).all(), ref
        if world > 1:
            assert g.stat(0, "group:replans") == 0
            # lists that outgrow the blocks (a capacity the host pinned too small): every rank sees the overflow mark, the
            # pass is repeated with a plan made from its own lists, and a caller of the two separate calls gets the error
            g.set_option("exchange_cap", 40)
            s, h = g.process()
            assert (s == so).all() and (h == ho).all()
            assert g.stat(0, "group:replans") == 1
            g.set_option("exchange_cap", 40)
            g.anchor()
            with pytest.raises(api.PhyloniumError, match="overflow"):
                g.compare()
            g.anchor()  # the failed comparison dropped the plan: the next pass fits
            s, h = g.compare()
            assert (s == so).all() and (h == ho).all()


def _tied_query(gs, ref, rng):
    """A query that carries stretches of genome `ref` once forward and once reverse-complemented between random flanks: the two
    homologies of a stretch project onto the same reference interval — equal projected starts, which only the host's std::sort
    orders as the reference does (process.cxx:438)."""
    clean = gs[ref][gs[ref] != ord("!")]
    pieces = []
    for x in range(1000, 21000, 1000):
        seg = clean[x:x + 500]
        pieces += [synth.random_base(100, rng), seg, synth.random_base(100, rng), synth.revcomp(seg)]
    return np.concatenate(pieces + [synth.random_base(200, rng)])


@pytest.mark.parametrize("world", [2, 3, 8])
def test_group_pass_is_one_queue_and_repeats_itself_the_long_way(world):
    """phylo_group_process (csrc/group.hip): a rank's pass as one queue — phylo_anchor_block_device, all-gather in place, attach,
    phylo_compare_triangle_device, all-reduce, the rank's rows into the node's shared home of the result — with one host wait.
    The first pass against a reference has no plan and takes phase A with a wait; the following ones are queued (the
    contexts count them).  A data set with a tied-start list raises the summed report's word 4 on its first queued pass:
    every rank repeats that pass the long way, ONCE — the route is remembered until genomes or reference change.  The
    result can stay in the group's own page-locked home (phylo_group_result_matrices).  All of it equals the oracle."""
    gs = synth.make_genomes(13, 25000, seed=86, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.08, contigs=2)
    dup = _tied_query(gs, 5, np.random.default_rng(7))
    with api.Group(world) as g:
        for tied in (False, True):
            g2 = gs + ([dup] if tied else [])
            n, ref = len(g2), 5
            r = O.Run(g2, ref).process()
            so, ho = r.matrix()
            g.set_genomes(g2)
            g.set_reference(ref)
            before = g.stat(0, "group:passes_repeated")
            for rep in range(4):
                s, h = g.process()
                assert (s == so).all() and (h == ho).all(), (world, tied, rep)
            assert g.stat(0, "group:shared_result") == 1
            assert g.stat(0, "group:passes_repeated") - before == (1 if tied else 0), (world, tied)
            if not tied:  # passes 2-4 were queued: their phase A was not waited for
                assert g.rank_context(0).stat("n:anchor_calls_without_a_wait", 0) >= 3
            vs, vh = g.process_in_place()
            assert (vs == so).all() and (vh == ho).all(), (world, tied)
            c0 = g.rank_context(world - 1)
            for j in range(n):
                assert hom_tuples_gpu(c0.homologies(j)) == hom_tuples_orc(r.homologies(j)), (world, tied, j)
            assert g.stat(0, "group:ms_step") > 0 and g.stat(0, "group:ms_queued") > 0
            # the two separate calls still give the same (phase A with a wait, the lists on every rank in between)
            g.anchor()
            s, h = g.compare()
            assert (s == so).all() and (h == ho).all(), (world, tied)


def _two_rank_worker(rank, world, port, out):
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from phylonium_amd import dist
    gs = synth.make_genomes(11, 30000, seed=71, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.06)
    c = api.Context(0)  # both ranks share the one GPU of the test box; collectives go over gloo
    c.set_genomes(gs)
    s, h = dist.process_sharded(c, 3, rank, world, device=None)
    if rank == 0:
        np.save(out + ".s.npy", s)
        np.save(out + ".h.npy", h)
    c.close()
    td.destroy_process_group()


def test_two_process_sharded_run_on_gpu(tmp_path):
    """The real multi-rank path (query shard → phase A → bulk exchange → window-range
    shard of phase B → matrix sum) with two processes; RCCL is replaced by gloo because
    the test box has a single GPU."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res")
    mp.spawn(_two_rank_worker, args=(2, port, out), nprocs=2, join=True)
    gs = synth.make_genomes(11, 30000, seed=71, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.06)
    so, ho = O.Run(gs, 3).process().matrix()
    assert (np.load(out + ".s.npy") == so).all()
    assert (np.load(out + ".h.npy") == ho).all()


def _queued_ranks_worker(rank, world, port, out):
    """One rank PROCESS of `world` sharing the box's one GPU: dist.process_sharded_device — the code an 8-GPU bench run
    executes — with dist.HostStagedCollectives standing in for RCCL on the same call sites."""
    import torch
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from phylonium_amd import dist
    dist._COLL = dist.HostStagedCollectives()
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))  # the library's kernels and the staged collectives on one stream of their own
    gs = synth.make_genomes(13, 25000, seed=86, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.08, contigs=2)
    dup = _tied_query(gs, 5, np.random.default_rng(7))
    log = {}
    c = api.Context(0)
    for tied in (False, True):
        g2 = gs + ([dup] if tied else [])
        n, ref = len(g2), 5
        # the one-context result, computed by this very process — the ranks in turn: eight processes starting cold on one GPU at
        # the same moment is not what this test is about (and is the one place it was ever seen to fail: profiles/EXPERIMENTS.md)
        for turn in range(world):
            if turn == rank:
                with api.Context(0) as one:
                    one.set_genomes(g2)
                    so, ho = one.process(ref)
            td.barrier()
        c.set_genomes(g2)
        c.set_reference(ref)
        c.reset_stats()
        for step in range(4):  # the first pass plans (phase A with a wait), the others are queued; every rank gets the result
            s, h = dist.process_sharded(c, ref, rank, world, device=dev, set_reference=False)
            assert (s == so).all() and (h == ho).all(), (rank, tied, step)
        plan = c._xplan
        assert plan["shared_result"] and c.stat("ms:result_rows") is not None  # `world` mappers of one segment
        if tied:  # word 4 of the summed report on the first queued pass: every rank repeats it the long way, once
            assert c._route and c._route["slow_anchor"], rank
        else:
            assert c.stat("n:anchor_calls_without_a_wait", 0) >= 3 and not c._route, rank
        # the result on one rank only (bench.py --gpus N): the others get nothing, views of the segment stay valid across a re-plan
        s, h = dist.process_sharded(c, ref, rank, world, device=dev, set_reference=False, result_rank=0, copy=False)
        assert (rank == 0 and (s == so).all() and (h == ho).all()) or (rank != 0 and s is None)
        views = plan["views"]
        # exchange blocks that have become too small: the overflow mark comes back in the summed report — or, with the
        # vector-ALU pair kernels, as the comparison's own error — and every rank plans again and repeats the pass
        for pairs_kernel in (0, 1):
            p = c._xplan
            nbytes = c.exchange_block_bytes(p["maxq"], 16)
            small = dict(p)
            blocks = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
            small.update({"cap": 16, "nbytes": nbytes, "all": blocks, "block": blocks[rank * nbytes:(rank + 1) * nbytes]})
            c._xplan = small
            c.set_option("pairs_kernel", pairs_kernel)
            s, h = dist.process_sharded(c, ref, rank, world, device=dev, set_reference=False)
            c.set_option("pairs_kernel", 0)
            assert (s == so).all() and (h == ho).all() and c._xplan["cap"] > 16, (rank, tied, pairs_kernel)
            assert c._xplan["views"][0].ctypes.data == views[0].ctypes.data  # the same segment, still mapped
        if rank == 0:
            assert (views[0] == so).all() and (views[1] == ho).all()
        td.barrier()
        # every list, on every rank, as the one context has it (the ranks in turn again)
        for turn in range(world):
            if turn == rank:
                with api.Context(0) as one:
                    one.set_genomes(g2)
                    one.set_reference(ref)
                    one.anchor()
                    for j in range(n):
                        assert hom_tuples_gpu(c.homologies(j)) == hom_tuples_gpu(one.homologies(j)), (rank, tied, j)
            td.barrier()
        log[str(tied)] = {"calls_without_a_wait": c.stat("n:anchor_calls_without_a_wait", 0), "route": c._route}
    td.barrier()
    c.close()
    if rank == world - 1:
        import json
        with open(out, "w") as f:
            json.dump(log, f)
    td.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 8])
def test_queued_pass_of_several_rank_processes_on_one_gpu(tmp_path, world):
    """dist.process_sharded_device with MORE THAN ONE RANK, as separate processes: the plan, the
    queued passes (phase A never waited for), the in-place all-gather of a view of its own output, `world` processes
    mapping and registering the one shared result segment and each writing its rows, the result on every rank and on one
    rank only, the repeat protocol — a tied-start list (report word 4), exchange blocks that overflow (word 2, and the
    vector-ALU kernels' own error) — with views of the segment surviving a re-plan.  RCCL refuses two ranks on one
    device, so dist.HostStagedCollectives carries the collectives over gloo on the same call sites and stream order.
    Every rank compares matrices and lists with a one-context run of its own."""
    import json
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "log.json")
    mp.spawn(_queued_ranks_worker, args=(world, port, out), nprocs=world, join=True)
    log = json.load(open(out))
    assert log["False"]["calls_without_a_wait"] >= 3 and log["True"]["route"]["slow_anchor"]


def _shared_home_worker(rank, world, name, out):
    import time
    import torch
    gs = synth.make_genomes(9, 20000, seed=72, d_range=(0.01, 0.2), indel_per_mbp=300, inv_frac=0.05)
    n = len(gs)
    c = api.Context(0)
    c.set_genomes(gs)
    c.set_reference(2)
    c.anchor()
    if rank == 0:
        c.result_open(name, create=True, ranks=world)
    else:
        for _ in range(200):  # (the tests have no collective to wait with: the segment appears when rank 0 has made it)
            try:
                c.result_open(name, create=False, ranks=world)
                break
            except api.PhyloniumError:
                time.sleep(0.05)
        else:
            raise RuntimeError("the shared segment never appeared")
    tri = torch.empty(c.triangle_words(), dtype=torch.int32, device="cuda:0")
    for step in range(3):  # deliveries are counted: the ranks stay in step from pass to pass
        c.compare_triangle_device(0, 1, tri.data_ptr())
        rep = c.triangle_rows_to_result(tri.data_ptr(), n * rank // world, n * (rank + 1) // world, rank, world if rank == 0 else 0)
        assert int(rep[3]) == 1 and not rep[[0, 1, 2, 4]].any()
        if rank == 0:
            s, h = c.result_matrices()
            np.save(out + ".s%d.npy" % step, s)
            np.save(out + ".h%d.npy" % step, h)
    if rank == 0:
        c.result_unlink()
    c.close()


def test_a_rank_that_gives_up_releases_the_ranks_waiting_for_it():
    """phylo_result_abandon: a rank that fails between the all-reduce and its delivery says so in its slot of the shared
    segment's header; the rank waiting for every rank's rows (wait_ranks > 0) returns with an error at once — not after its
    minute's time-out — and the segment can be opened anew."""
    import time
    import torch
    gs = synth.make_genomes(6, 15000, seed=73, d_range=(0.01, 0.2))
    n = len(gs)
    name = "/phylonium_amd_test3_%d" % os.getpid()
    a, b = api.Context(0), api.Context(0)
    try:
        for c in (a, b):
            c.set_genomes(gs)
            c.set_reference(1)
            c.anchor()
        a.result_open(name, create=True, ranks=2)
        b.result_open(name, create=False, ranks=2)
        a.result_unlink()
        tri = torch.empty(a.triangle_words(), dtype=torch.int32, device="cuda:0")
        a.compare_triangle_device(0, 1, tri.data_ptr())
        b.result_abandon(1)
        t0 = time.time()
        with pytest.raises(api.PhyloniumError, match="gave the pass up"):
            a.triangle_rows_to_result(tri.data_ptr(), 0, n // 2, 0, 2)
        assert time.time() - t0 < 5
        for c in (a, b):
            c.result_close()
        a.result_open(name, create=True, ranks=2)  # a new segment under the same name: the old one is gone
        b.result_open(name, create=False, ranks=2)
        a.result_unlink()
        rep = b.triangle_rows_to_result(tri.data_ptr(), n // 2, n, 1, 0)
        rep = a.triangle_rows_to_result(tri.data_ptr(), 0, n // 2, 0, 2)
        s, h = a.result_matrices()
        so, ho = O.Run(gs, 1).process().matrix()
        assert (s == so).all() and (h == ho).all() and int(rep[3]) == 1
    finally:
        a.close()
        b.close()


def test_two_processes_write_their_rows_of_one_shared_result(tmp_path):
    """The result's shared page-locked home across processes (phylo_result_open with a POSIX shared-memory name): two
    processes on the test box's one GPU map and register the same segment, each one's device writes its half of the rows
    of both matrices, rank 0 waits for both deliveries (a counter per rank in the segment's header) and reads the whole
    result — three passes (in the library's own use the collectives of the next pass keep a rank from writing rows the
    result's rank is still reading; here nothing is written between passes that differs)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "res")
    name = "/phylonium_amd_test2_%d" % os.getpid()
    mp.spawn(_shared_home_worker, args=(2, name, out), nprocs=2, join=True)
    gs = synth.make_genomes(9, 20000, seed=72, d_range=(0.01, 0.2), indel_per_mbp=300, inv_frac=0.05)
    so, ho = O.Run(gs, 2).process().matrix()
    for step in range(3):
        assert (np.load(out + ".s%d.npy" % step) == so).all() and (np.load(out + ".h%d.npy" % step) == ho).all(), step
    assert not os.path.exists("/dev/shm" + name)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PHY_FUZZ_SEEDS", "48"))))  # a long sweep: PHY_FUZZ_SEEDS=300
def test_fuzz_small_random_sets(ctx, seed):
    """Randomised shapes: genome count, lengths, divergence, structure, contigs, chunk
    and k-mer sizes all drawn per seed; every tally and homology list must match."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(2, 9))
    length = int(rng.integers(300, 25000))
    d_hi = float(rng.choice([0.02, 0.1, 0.3]))
    gs = synth.make_genomes(n, length, seed=2000 + seed, d_range=(0.001, d_hi), tree=bool(rng.integers(0, 2)),
                            indel_per_mbp=float(rng.choice([0, 500, 3000])), inv_frac=float(rng.choice([0, 0.05, 0.3])),
                            contigs=int(rng.choice([1, 1, 2, 5])) if length > 2000 else 1, inv_len=(50, max(60, length // 10)))
    if rng.random() < 0.3:
        gs.append(gs[int(rng.integers(0, n))].copy())  # an exact duplicate
    if rng.random() < 0.3:
        gs.append(synth.random_base(int(rng.integers(1, 400)), rng))  # an unrelated short one
    ref = int(rng.integers(0, len(gs)))
    chunk = int(rng.choice([0, 64, 128, 192, 448, 512]))
    kmer = int(rng.choice([0, 0, 2, 5]))
    backend = int(rng.integers(0, 2))
    threshold = int(rng.choice([0, 0, 0, 17, 21]))  # 17+: what references beyond ~60 Mbp have
    check_process(ctx, gs, ref, chunk=chunk, kmer=kmer, backend=backend, threshold=threshold)


@pytest.mark.parametrize("world", [2, 3, 5, 6])
def test_fuzz_group_passes(world):
    """The C++ group's pass (csrc/group.hip) over randomised sets — genome count (fewer genomes than ranks among them: ranks
    with an empty block), lengths, divergence, contigs, duplicates, an unrelated short genome, a tied-start query — and
    randomised options: four passes each (the first plans, the others are queued), against the oracle's matrices and lists;
    one group serves all sets (the plan, the route and the result's home follow the changes of genomes and reference)."""
    with api.Group(world) as g:
        for seed in range(8):
            rng = np.random.default_rng(7000 + 31 * world + seed)
            n = int(rng.integers(2, 12))
            length = int(rng.integers(400, 20000))
            gs = synth.make_genomes(n, length, seed=3000 + seed, d_range=(0.001, float(rng.choice([0.02, 0.1, 0.3]))), tree=bool(rng.integers(0, 2)),
                                    indel_per_mbp=float(rng.choice([0, 500, 3000])), inv_frac=float(rng.choice([0, 0.05, 0.3])),
                                    contigs=int(rng.choice([1, 2, 5])) if length > 2000 else 1, inv_len=(50, max(60, length // 10)))
            if rng.random() < 0.4:
                gs.append(gs[int(rng.integers(0, n))].copy())
            if rng.random() < 0.3:
                gs.append(synth.random_base(int(rng.integers(1, 400)), rng))
            ref = int(rng.integers(0, len(gs)))
            if rng.random() < 0.3 and len(gs[ref]) > 3000:  # forward + reverse copies of stretches of the reference: tied projected starts
                clean = gs[ref][gs[ref] != ord("!")]
                pieces = []
                for x in range(200, len(clean) - 600, 700):
                    seg = clean[x:x + 300]
                    pieces += [synth.random_base(60, rng), seg, synth.random_base(60, rng), synth.revcomp(seg)]
                gs.append(np.concatenate(pieces + [synth.random_base(100, rng)]))
            r = O.Run(gs, ref).process()
            so, ho = r.matrix()
            g.set_genomes(gs)
            g.set_option("chunk", int(rng.choice([0, 64, 128, 448])))
            g.set_option("kmer", int(rng.choice([0, 0, 3, 5])))
            g.set_reference(ref)
            for rep in range(4):
                s, h = g.process() if rep != 2 else g.process_in_place()
                assert (s == so).all() and (h == ho).all(), (world, seed, rep)
            c = g.rank_context(int(rng.integers(0, world)))
            for j in range(len(gs)):
                assert hom_tuples_gpu(c.homologies(j)) == hom_tuples_orc(r.homologies(j)), (world, seed, j)
            g.set_option("chunk", 0)
            g.set_option("kmer", 0)


def _nccl_one_rank_worker(rank, world, port, out):
    import torch
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    td.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from phylonium_amd import dist
    dist._FORCE_COLLECTIVES = True
    gs = synth.make_genomes(6, 20000, seed=91, d_range=(0.01, 0.2), inv_frac=0.05)
    c = api.Context(0)
    c.set_genomes(gs)
    s, h = dist.process_sharded(c, 1, rank, world, device=dev)  # block exchange on torch's stream, u32 triangle on the wire
    s, h = s.copy(), h.copy()
    s3, h3 = dist.process_sharded(c, 1, rank, world, device=dev)  # once more: the planned capacity is reused
    assert (s3 == s).all() and (h3 == h).all()
    s4, h4 = dist.process_sharded(c, 1, rank, world, device=dev, result_rank=0)  # a reduce to the result's rank (bench.py --gpus N)
    assert (s4 == s).all() and (h4 == h).all()
    assert c.stat("n:anchor_calls_without_a_wait", 0) >= 2 and c.stat("ms:result_rows") is not None  # queued passes, the shared home
    # exchange blocks that have become too small (the lists grew since the plan was made): the overflow mark of the block comes
    # back in the summed report — or, with the vector-ALU pair kernels, as the comparison's own error — and the pass is
    # planned again and repeated, by every rank alike
    import torch
    for pairs_kernel in (0, 1):
        p = c._xplan
        nbytes = c.exchange_block_bytes(p["maxq"], 16)
        small = dict(p)
        blocks = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
        small.update({"cap": 16, "nbytes": nbytes, "all": blocks, "block": blocks[rank * nbytes:(rank + 1) * nbytes]})
        c._xplan = small
        c.set_option("pairs_kernel", pairs_kernel)
        s6, h6 = dist.process_sharded(c, 1, rank, world, device=dev, result_rank=0)
        c.set_option("pairs_kernel", 0)
        assert (s6 == s).all() and (h6 == h).all() and c._xplan["cap"] > 16, pairs_kernel
    dist._SHARED_RESULT = False  # a node whose ranks cannot share a segment: one rank fetches the result
    c._xplan = None
    for rr in (None, 0):
        s5, h5 = dist.process_sharded(c, 1, rank, world, device=dev, result_rank=rr)
        assert (s5 == s).all() and (h5 == h).all()
    dist._SHARED_RESULT = True
    c._xplan = None
    dist._LEGACY_DEVICE_EXCHANGE = True  # round 2's exchange: counts through the host, u64 matrices on the wire
    s2, h2 = dist.process_sharded(c, 1, rank, world, device=dev)
    assert (s2 == s).all() and (h2 == h).all()
    np.save(out + ".s.npy", s)
    np.save(out + ".h.npy", h)
    c.close()
    td.destroy_process_group()


def test_rccl_collectives_in_a_one_rank_group(tmp_path):
    """bench.py's N>1 path uses torch.distributed with backend nccl (= RCCL). The test
    box has one GPU, so run the same exchange code in a one-rank RCCL group: the
    all_reduce / all_gather_into_tensor calls on CUDA tensors must work and leave the
    result unchanged."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res")
    mp.spawn(_nccl_one_rank_worker, args=(1, port, out), nprocs=1, join=True)
    gs = synth.make_genomes(6, 20000, seed=91, d_range=(0.01, 0.2), inv_frac=0.05)
    so, ho = O.Run(gs, 1).process().matrix()
    assert (np.load(out + ".s.npy") == so).all() and (np.load(out + ".h.npy") == ho).all()


def _long_repeat_set():
    """Two identical 70 kbp copies in the reference: LCP values beyond the 16-bit clip of the SAX
    records (and beyond the device LCP builder's cap), matches that run past 65535 characters."""
    rng = np.random.default_rng(77)
    R = synth.random_base(70000, rng)
    X = [synth.random_base(3000, rng) for _ in range(4)]
    ref = np.concatenate([X[0], R, X[1], R, X[2]])
    q1 = np.concatenate([X[0], R, synth.mutate(X[1], 0.02, rng)])          # unique only beyond the repeat
    q2 = np.concatenate([X[3], R[:68000], synth.random_base(500, rng)])     # ends inside the repeat: never unique
    q3 = synth.mutate(ref, 0.01, rng)
    q4 = np.concatenate([synth.revcomp(R)[:69000], X[3]])                   # the same on the reverse strand
    return [ref, q1, q2, q3, q4]

def test_long_repeat_beyond_the_lcp_clip(ctx):
    gs = _long_repeat_set()
    check_process(ctx, gs, 0)
    check_process(ctx, gs, 0, chunk=1024, backend=1)
    check_process(ctx, gs, 3)


@pytest.mark.parametrize("filt", [1, 2])
def test_sort_filter_on_host_and_on_device(ctx, filt):
    """The same lists whether phase A's sort + chain filter runs on the host cores or on the device;
    lists with equal projected starts (here: a query that is two copies of one segment, so both
    copies anchor at the same reference position) are left to the host's std::sort by the device path."""
    gs = synth.make_genomes(8, 60000, seed=31, d_range=(0.005, 0.25), indel_per_mbp=400, inv_frac=0.08, contigs=2)
    dup = np.concatenate([gs[0][1000:9000], synth.random_base(300, np.random.default_rng(1)), gs[0][1000:9000]])
    gs = gs + [dup]
    check_process(ctx, gs, 0, filt=filt)
    check_process(ctx, gs, 5, chunk=192, filt=filt)
    ctx.set_option("filter", filt)
    ctx.set_genomes(gs)
    ctx.set_reference(0)
    ctx.anchor(2, 7)   # a partial range keeps the other lists
    ctx.anchor(0, 2)
    ctx.anchor(7, 9)
    s, h = ctx.compare()
    so, ho = O.Run(gs, 0).process().matrix()
    assert (s == so).all() and (h == ho).all()
    ctx.set_option("filter", 0)


def test_many_queries_default_options(ctx):
    """140 queries: with default options the sort + chain filter runs on the device (128 queries or
    more), lists stay device-resident and the projection follows phase A directly."""
    gs = synth.make_genomes(140, 12000, seed=57, d_range=(0.005, 0.3), indel_per_mbp=400, inv_frac=0.05)
    ctx.set_option("filter", 0)
    ctx.set_genomes(gs)
    s, h = ctx.process(ref_idx=7)
    assert ctx.stat("n:anchor_filter") is None or True  # (present only when kernels are being timed)
    r = O.Run(gs, 7).process(threads=4)
    so, ho = r.matrix()
    assert (s == so).all() and (h == ho).all()
    for j in (0, 7, 63, 139):  # read back on demand
        assert hom_tuples_gpu(ctx.homologies(j)) == hom_tuples_orc(r.homologies(j))
    # parts of the comparison project their window ranges into the planes phase A's projection left: a whole
    # comparison afterwards must project again (round 5: it took the planes for phase A's still)
    s2, h2 = np.zeros_like(s), np.zeros_like(h)
    for part in range(3):
        a, b = ctx.compare(part, 3)
        s2 += a
        h2 += b
    assert (s2 == so).all() and (h2 == ho).all()
    s3, h3 = ctx.compare()
    assert (s3 == so).all() and (h3 == ho).all()
    # the result's page-locked home as the caller's matrices: the device writes them itself (both phases as one call, and a
    # comparison alone), an odd number of genomes included (the second matrix then starts on a boundary of its own)
    ctx.result_open(None, ranks=1)
    try:
        views = ctx.result_matrices()
        for call in (ctx.anchor_compare, lambda out: ctx.compare(0, 1, out=out)):
            views[0][:] = 7
            views[1][:] = 7
            s4, h4 = call(out=views)
            assert (s4 == so).all() and (h4 == ho).all()
    finally:
        ctx.result_close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("workload", ["c3", "c3tree", "c5s", "c4", "c2like", "c3dup", "c5"])
def test_full_size_properties_and_reference_row(workload, sample_seed):
    """BASELINE configs[2] and [3] at full size (256 and 1024 x 5 Mbp), the tree-shaped variant, the stand-in for
    configs[1] (c2like: 29 x 5 Mbp at d <= 0.03 — the eco29 files are not in the image), a set with byte-identical
    and near-identical genomes (c3dup: phase A's overrun path), and the multi-contig, 10 %-inverted configs[4]
    both scaled to 16 x 20 Mbp and at full size (c5: 64 x 100 Mbp in 100 contigs each; bench.py's generator and
    seed).  The oracle would take
    minutes here, so the checks are the size-independent ones: the matrices are symmetric with an empty
    diagonal, substitutions <= homologs <= the shorter genome; every filtered list is sorted and
    non-overlapping on the reference; and the reference's row is recomputed by another route — the B0
    kernels (seqcmp / revseqcmp semantics over the resident genomes) summed over each query's list — which
    shares nothing with the pileup kernels that made the matrix; and a sub-matrix of three or four genomes,
    at full length, against the oracle.  (bench.py --check compares the first six genomes with the oracle at this size.)"""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n, length, d_range, indel, inv, _ = bench.WORKLOADS[workload]
    dev = torch.device("cuda", 0)
    buf, offs, lens = bench.make_genomes_gpu(torch, n, length, 20260101, dev, d_range, indel, inv,
                                             contigs=bench.CONTIGS.get(workload, 1), tree=workload in bench.TREE,
                                             dup=bench.DUP.get(workload, (0, 0)))
    torch.cuda.synchronize()
    with api.Context(0) as c:
        c.set_genomes_device(buf.data_ptr(), offs, lens)
        s, h = c.process(0)
        L = np.asarray(lens, np.uint64)
        assert (s == s.T).all() and (h == h.T).all() and (np.diag(h) == 0).all() and (np.diag(s) == 0).all()
        assert (s <= h).all() and (h <= np.minimum(L[:, None], L[None, :])).all()
        assert (h[0, 1:] > 0).all()
        for j in sorted({1, 2, n // 6, n // 3 + 1, n // 2, n - 2, n - 1}):
            hom = c.homologies(j)
            assert hom.size > (100 if workload in ("c3", "c3tree", "c4", "c5s", "c5") else 0)
            start, ln = hom["index_reference_projected"], hom["length"]
            assert (start[1:] >= start[:-1] + ln[:-1]).all()                     # sorted, non-overlapping
            assert (hom["index_query"] + ln <= lens[j]).all() and (start + ln <= lens[0]).all()
            assert int(ln.sum()) == int(h[0, j])
            m = hom.size
            sub = c.seqcmp_batch(np.zeros(m, np.uint32), start, np.full(m, j, np.uint32), hom["index_query"], ln,
                                 (hom["direction"] != 0).astype(np.uint8))
            assert int(sub.sum()) == int(s[0, j])
        # a handful of genomes through the oracle at full length: a pair's tallies depend on the reference
        # and the two genomes only, so the small run must reproduce the sub-matrix
        # (every workload: c4 checks a pair kernel tile table of 1024 genomes, c5 — three genomes of 100 Mbp in 100
        # contigs, 10 % inverted — the five-plane kernels and the long-list filter at full length)
        if workload != "c5":  # (c5's 100 Mbp genomes go through the oracle once: the drawn sample below)
            idx = {"c2like": [0, 1, n - 1], "c3dup": [0, 2, 10, n - 1], "c5": [0, 1, n - 1],
                   "c4": [0, 1, n // 2 + 1, n - 1]}.get(workload, [0, 1, n // 2, n - 1])
            gs = [buf[offs[j]:offs[j] + lens[j]].cpu().numpy() for j in idx]
            refb = bytes(gs[0])
            sa = api.host_suffix_array(refb + b"#" + O.revcomp(refb))  # unique: spares the oracle's slow sorter
            so, ho = O.Run(gs, 0).process(sa=sa, threads=8).matrix()
            assert (s[np.ix_(idx, idx)] == so).all() and (h[np.ix_(idx, idx)] == ho).all()
        # the sharded comparison adds up to the same matrices
        s2, h2 = np.zeros_like(s), np.zeros_like(h)
        for part in range(3):
            a, b = c.compare(part, 3)
            s2 += a
            h2 += b
        assert (s2 == s).all() and (h2 == h).all()
        # the whole matrix once more with the vector-ALU pair kernels (five planes, popcounts): a second implementation
        # of the tallies over the same planes — a wrong tile of the matrix-core kernel shows here
        c.set_option("pairs_kernel", 1)
        try:
            sv, hv = c.compare()
        finally:
            c.set_option("pairs_kernel", 0)
        assert (sv == s).all() and (hv == h).all(), "matrix-core and vector-ALU pair kernels disagree: " + \
            str([(int(i), int(j)) for i, j in zip(*np.nonzero((sv != s) | (hv != h)))][:8])
        # Pairs off the reference's row, by routes that share nothing with the pileup.  A sample of genomes drawn anew
        # every run (the run's seed and the workload's name: PHY_SAMPLE_SEED pins it; the test id and every assertion message name it): always the reference, one genome
        # of the last 64-genome tile, and two of one and the same tile; their sub-matrix AND their lists
        #   (i)  against the oracle run on those genomes alone (a pair's tallies depend on the reference and the two
        #        genomes only), and
        #   (ii) against the literal route: the big context's lists installed in a second context (phylo_set_homologies),
        #        compare_backend = 1 — the reference's merge-join restated on the host (pair_segments, process.cxx:566-658)
        #        + the byte kernels (seqcmp_batch_kernel) over the resident bytes.
        import zlib
        seed = sample_seed  # (conftest.py: one value per run, in this test's id and in gpurun_out/sample_seed.txt)
        srng = np.random.default_rng([seed, zlib.crc32(workload.encode())])
        want = {"c5": 4, "c5s": 6}.get(workload, 10)
        ntile = (n + 63) // 64
        pick = {0, int(srng.integers(64 * (ntile - 1), n))}
        t2 = int(srng.integers(0, ntile))
        lo2, hi2 = 64 * t2, min(n, 64 * t2 + 64)
        for v in srng.choice(np.arange(lo2, hi2), size=min(2, hi2 - lo2), replace=False):
            pick.add(int(v))
        while len(pick) < min(want, n):
            pick.add(int(srng.integers(1, n)))
        idx2 = sorted(pick)
        tag = f"[workload {workload}, PHY_SAMPLE_SEED={seed}, genomes {idx2}]"
        gs2 = [buf[offs[j]:offs[j] + lens[j]].cpu().numpy() for j in idx2]
        refb = bytes(gs2[0])
        sa = api.host_suffix_array(refb + b"#" + O.revcomp(refb))
        r2 = O.Run(gs2, 0).process(sa=sa, threads=8)
        so2, ho2 = r2.matrix()
        sub_s, sub_h = s[np.ix_(idx2, idx2)], h[np.ix_(idx2, idx2)]
        assert (sub_h == ho2).all() and (sub_s == so2).all(), "sub-matrix differs from the oracle " + tag
        lists = [c.homologies(j).copy() for j in idx2]
        for t, j in enumerate(idx2):
            assert hom_tuples_gpu(lists[t]) == hom_tuples_orc(r2.homologies(t)), f"list of genome {j} differs from the oracle " + tag
        r2.close()
        with api.Context(0) as c2:
            c2.set_genomes_device(buf.data_ptr(), [offs[j] for j in idx2], [lens[j] for j in idx2])
            c2.set_reference(0)
            for t in range(len(idx2)):
                c2.set_homologies(t, lists[t])
            c2.set_option("compare_backend", 1)
            s3, h3 = c2.compare()
            assert c2.stat("count:segments", 0) > 0
        assert (s3 == sub_s).all() and (h3 == sub_h).all(), "sub-matrix differs from the segment route (merge-join + byte kernels) " + tag


def test_b0_under_the_reference_names_from_eight_threads():
    """libphylonium_amd_b0.so exports seam B0 under the reference's own names and calling convention
    (libs/seqcmp.h:14-25, libs/revseqcmp.h:25-33): pure functions over borrowed host buffers that
    evo_model::account* calls from an OpenMP team (src/evo_model.cxx:53-75).  Eight host threads call
    both at once, each on its own pair of strings; every result equals the oracle's."""
    import ctypes as C
    import threading
    b0 = C.CDLL(os.path.join(os.path.dirname(api.LIB_PATH), "libphylonium_amd_b0.so"))
    for f in (b0.seqcmp, b0.revseqcmp):
        f.restype = C.c_size_t
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(8)
    alpha = np.frombuffer(b"ACGT!", np.uint8)
    jobs = []
    for t in range(8):
        n = int(rng.integers(1, 400000))
        a = alpha[rng.integers(0, 5, n)]
        b = np.where(rng.random(n) < 0.7, a, alpha[rng.integers(0, 5, n)])
        jobs.append((np.ascontiguousarray(a), np.ascontiguousarray(b), n))
    got = [None] * 8

    def work(t):
        a, b, n = jobs[t]
        res = []
        for rep in range(6):
            m = n if rep == 0 else max(0, n - 37 * rep)
            res.append((b0.seqcmp(a.ctypes.data, b.ctypes.data, m), b0.revseqcmp(a.ctypes.data, b.ctypes.data, m)))
        got[t] = res

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for t in range(8):
        a, b, n = jobs[t]
        for rep in range(6):
            m = n if rep == 0 else max(0, n - 37 * rep)
            assert got[t][rep] == (O.seqcmp(a, b, m), O.revseqcmp(a, b, m)), (t, rep)

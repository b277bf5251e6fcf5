"""Phase-A algorithm (k-mer bucket + suffix-array search, speculative chunks,
bridges, fold, O(n log n) chain filter) compiled for the CPU and compared with
the oracle's restatement of the reference (ESA walk, sequential chain, O(n²)
filter).  Bit-exact on the raw homology stream (order included) and on the
sorted + filtered lists."""
import os

import numpy as np
import pytest

import emul_lib as E
import oracle_lib as O
from phylonium_amd import synth


MODES = (1, 3)  # the chain on its packed path, and with every step through its slow resolver


def assert_same(gs, ref, chunk=0, kmer=0, allow_quirk=False, threshold=0, modes=MODES):
    r = O.Run(gs, ref, threshold=threshold).process(compare=False)
    e = None
    for mode in modes:
        e = E.EmulRun(gs, ref, chunk=chunk, kmer=kmer, threshold=threshold, mode=mode)
        assert e.error == 0
        assert e.threshold == r.threshold
        for j in range(len(gs)):
            ro = r.homologies(j, filtered=False)
            re_ = e.raw(j)
            got = [tuple(int(x) for x in row) for row in re_]
            want = [(int(a["iref"]), int(a["iq"]), int(a["len"])) for a in ro]
            assert got == want, f"raw homologies differ for query {j} (mode {mode})"
            fo, fe = r.homologies(j), e.filtered(j)
            got = [(int(b["direction"]), int(b["index_reference"]), int(b["index_reference_projected"]),
                    int(b["index_query"]), int(b["length"])) for b in fe]
            want = [(int(a["rev"]), int(a["iref"]), int(a["iproj"]), int(a["iq"]), int(a["len"])) for a in fo]
            assert got == want, f"filtered homologies differ for query {j} (mode {mode})"
    return r, e


def test_lean_chain_rarely_needs_its_slow_resolver():
    """On ordinary genomes (substitutions, indels, inversions, a few contigs) the packed path answers
    nearly every step; the slow resolver is for windows next to '!' or the query's end, repeats of 16+
    bases and oversized buckets."""
    gs = synth.make_genomes(6, 60000, seed=11, d_range=(0.01, 0.3), indel_per_mbp=300, inv_frac=0.05, contigs=3)
    r, e = assert_same(gs, 0, modes=(1,))
    assert e.steps_spec > 10000
    assert e.slow_steps * 100 < e.steps_spec + e.steps_bridge, (e.slow_steps, e.steps_spec, e.steps_bridge)


@pytest.mark.parametrize("chunk", [0, 64, 128, 192, 1024, 1088])
def test_star_substitutions(chunk):
    gs = synth.make_genomes(5, 20000, seed=3, d_range=(0.01, 0.25))
    assert_same(gs, 0, chunk=chunk)
    assert_same(gs, 3, chunk=chunk)


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_indels_inversions_contigs(seed):
    gs = synth.make_genomes(6, 30000, seed=seed, d_range=(0.01, 0.3), indel_per_mbp=500, inv_frac=0.1,
                            contigs=3, inv_len=(100, 1500))
    for ref in (0, 4):
        for chunk, k in ((256, 0), (320, 0), (64, 3), (64, 1)):
            # (a reference on which the 6-mer cache holds an over-deep entry, esa.cxx:174-199, is flagged — and the
            # reference's answers on it are reproduced: test_cache_quirk_is_reproduced)
            assert E.cache_quirk(gs[ref]) == bool(O.Esa(gs[ref]).cache_quirks())
            assert_same(gs, ref, chunk=chunk, kmer=k)


def test_tree_low_divergence():
    gs = synth.make_genomes(6, 60000, seed=9, d_range=(0.0005, 0.03), tree=True, indel_per_mbp=200, inv_frac=0.05)
    assert_same(gs, 0)
    assert_same(gs, 5, chunk=128)


def test_identical_and_self():
    rng = np.random.default_rng(21)
    a = synth.random_base(9000, rng)
    assert_same([a, a.copy(), a[100:5000].copy()], 0, chunk=128)
    assert_same([a, a.copy()], 1)


def test_reverse_complement_genome():
    rng = np.random.default_rng(22)
    a = synth.random_base(8000, rng)
    b = synth.revcomp(synth.mutate(a, 0.03, rng))
    r, e = assert_same([a, b], 0, chunk=128)
    assert r.homologies(1)["rev"].all()


def test_repeats_and_low_complexity():
    rng = np.random.default_rng(23)
    unit = synth.random_base(700, rng)
    spacer = [synth.random_base(1500, rng) for _ in range(5)]
    a = np.concatenate([spacer[0], unit, spacer[1], unit, spacer[2], synth.revcomp(unit), spacer[3],
                        np.frombuffer(b"A" * 300 + b"AC" * 200 + b"T" * 100, np.uint8), spacer[4]])
    b = synth.mutate(a, 0.02, rng)
    c = synth.mutate(np.concatenate([spacer[2], unit, unit, spacer[0], np.frombuffer(b"A" * 500, np.uint8)]), 0.01, rng)
    for ref in (0, 1, 2):
        assert_same([a, b, c], ref, chunk=64)
        assert_same([a, b, c], ref, chunk=256, kmer=2)


def test_short_and_empty_queries():
    rng = np.random.default_rng(24)
    a = synth.random_base(5000, rng)
    gs = [a, a[:5].copy(), a[10:11].copy(), np.zeros(0, np.uint8), a[200:230].copy(),
          np.frombuffer(b"!!!!", np.uint8).copy(), np.frombuffer(b"ACGT!ACGT", np.uint8).copy()]
    assert_same(gs, 0, chunk=64)


def test_unrelated():
    rng = np.random.default_rng(25)
    gs = [synth.random_base(20000, rng), synth.random_base(20000, rng)]
    r, e = assert_same(gs, 0, chunk=256)
    assert len(r.homologies(1)) == 0


def test_golden_simple(golden_dir):
    g = [O.read_fasta_genome(os.path.join(golden_dir, f"simple{i}.fasta.gz")) for i in (0, 1)]
    r, e = assert_same(g, 1)
    assert len(e.filtered(0)) == 68
    assert_same(g, 1, chunk=512)


@pytest.mark.slow
def test_golden_cfg1(golden_dir):
    g = [O.read_fasta_genome(os.path.join(golden_dir, f"cfg1_{i}.fasta.gz")) for i in (0, 1)]
    r, e = assert_same(g, 1)
    assert e.threshold == 14 and len(e.filtered(0)) == 334
    first = e.filtered(0)[:3]
    assert [(int(x["index_reference"]), int(x["index_query"]), int(x["length"])) for x in first] == \
        [(0, 0, 852), (936, 936, 478), (1518, 1518, 1428)]


def test_suffix_array_and_tables_against_oracle():
    rng = np.random.default_rng(26)
    for n, contigs in ((1, 1), (2, 1), (257, 1), (5000, 1), (4000, 5)):
        s = synth.split_contigs(synth.random_base(n, rng), contigs, rng).tobytes()
        S = s + b"#" + O.revcomp(s)
        sa = E.suffix_array(S)
        assert (sa.astype(np.int64) == O.suffix_array(S)).all()
        lcp = E.lcp(S, sa)
        for r in range(1, len(S)):
            a, b = S[sa[r - 1]:], S[sa[r]:]
            m = 0
            while m < min(len(a), len(b)) and a[m] == b[m]:
                m += 1
            assert lcp[r] == m
            if r > 300:
                break
        for k in (1, 2, 5):
            T = E.kmer_table(S, k)
            assert T[-1] == len(S)
            sufs = sorted(S[i:] for i in range(len(S)))
            import bisect
            for c in range(4 ** k):
                km = bytes(b"ACGT"[(c >> (2 * (k - 1 - t))) & 3] for t in range(k))
                assert T[c] == bisect.bisect_left(sufs, km)
            if n > 1000:
                break
    S = (b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA" * 30)
    assert (E.suffix_array(S).astype(np.int64) == O.suffix_array(S)).all()


def test_suffix_array_on_several_cores():
    """The bucket + per-bucket std::sort construction gives SA-IS's array, on any thread count, and
    gives up (so that SA-IS runs) on text whose repeats would make its comparisons deep."""
    rng = np.random.default_rng(27)
    for n, contigs in ((2, 1), (3, 1), (50, 1), (257, 1), (5000, 1), (4000, 5), (150000, 3)):
        s = synth.split_contigs(synth.random_base(n, rng), contigs, rng).tobytes()
        S = s + b"#" + O.revcomp(s)
        want = E.suffix_array(S)
        for threads in (1, 3, 8):
            ok, sa = E.suffix_array_buckets(S, threads)
            assert ok and (sa == want).all()
    # repeats within the budget: several copies of a 3 kbp unit (rRNA-operon-like)
    unit = synth.random_base(3000, rng).tobytes()
    S = b"".join(synth.random_base(20000, rng).tobytes() + unit for _ in range(7))
    S = S + b"#" + O.revcomp(S)
    ok, sa = E.suffix_array_buckets(S, 4)
    assert ok and (sa == E.suffix_array(S)).all()
    # beyond it: the sort declines, never returns a wrong array
    for S in (b"A" * 3000, b"AC" * 5000 + b"G", synth.random_base(70000, rng).tobytes() * 3):
        ok, sa = E.suffix_array_buckets(S, 4)
        assert not ok or (sa == E.suffix_array(S)).all()
    ok, _ = E.suffix_array_buckets(synth.random_base(70000, rng).tobytes() * 40, 8)
    assert not ok
    # a text with a zero byte cannot use the zero padding as its end marker
    ok, _ = E.suffix_array_buckets(b"ACGT\0ACGT" * 10, 2)
    assert not ok


@pytest.mark.parametrize("threshold", [17, 24, 40])
def test_thresholds_of_long_references(threshold):
    """Thresholds a 60 Mbp+ reference has (forced here): the lucky check needs more than one 16-byte window."""
    gs = synth.make_genomes(4, 30000, seed=threshold, d_range=(0.01, 0.15), indel_per_mbp=300, inv_frac=0.06, contigs=2)
    assert_same(gs, 0, threshold=threshold)
    assert_same(gs, 2, chunk=128, threshold=threshold)


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

def test_long_repeat_beyond_the_lcp_clip():
    gs = _long_repeat_set()
    assert_same(gs, 0)
    assert_same(gs, 3, chunk=1024)


@pytest.mark.parametrize("chunk", [64, 256, 1024, 0])
def test_overruns_of_near_identical_genomes(chunk):
    """Genomes that equal the reference over many chunk lengths: the lean chains cut every speculative
    comparison one chunk length past its chunk's end and hand the true end down the run of chunks
    afterwards (lean_core.h: overruns) — same homologies as the reference's sequential chain."""
    rng = np.random.default_rng(31)
    a = synth.random_base(40000, rng)
    b = a.copy()
    b[[7000, 7001, 23000]] = synth.random_base(3, rng)            # three substitutions in 40 kbp
    c = np.concatenate([a[:15000], a[15010:]])                    # a deletion
    d = np.concatenate([a[:5000], synth.revcomp(a[5000:30000]), a[30000:]])  # a 25 kbp inversion
    e = synth.split_contigs(a.copy(), 3, rng)                     # '!' inside otherwise identical sequence
    gs = [a, a.copy(), b, c, d, e, synth.mutate(a, 0.0003, rng)]
    r, em = assert_same(gs, 0, chunk=chunk, modes=(1,))
    if chunk and chunk <= 1024:
        assert em.overruns > 20
    assert_same(gs, 6, chunk=chunk, modes=(1, 3))


def test_overruns_with_long_repeats():
    """A 6 kbp stretch present three times in the reference, queries near-identical to it: cut comparisons
    whose neighbours in the suffix array share kilobases (the LCP test of may_cut) and runs that end
    because the next chunk's longest match lies on another diagonal."""
    rng = np.random.default_rng(32)
    rep = synth.random_base(6000, rng)
    a = np.concatenate([synth.random_base(9000, rng), rep, synth.random_base(4000, rng), rep, synth.random_base(7000, rng), rep,
                        synth.random_base(3000, rng)])
    b = a.copy()
    b[[12000, 20500, 31000]] = synth.random_base(3, rng)
    for chunk in (128, 512, 2048):
        assert_same([a, b, a.copy()], 0, chunk=chunk, modes=(1,))
        assert_same([a, b, a.copy()], 1, chunk=chunk, modes=(1,))


def test_cache_quirk_detector_agrees_with_the_restated_cache():
    """phylo_reference_cache_quirk's host walk (hostlogic.hpp: esa_cache_quirk) against the oracle's restated
    6-mer cache (esa.cxx:90-228): flagged exactly when the cache holds an entry deeper than its key matches.
    Small multi-contig references (where the bug lives), crafted positives, and larger ones (never)."""
    rng = np.random.default_rng(77)
    seen = {True: 0, False: 0}
    cases = []
    for trial in range(400):
        n = int(rng.integers(8, 400))
        contigs = int(rng.integers(2, 12))
        g = synth.split_contigs(synth.random_base(n, rng), min(contigs, n // 3 + 1), rng)
        cases.append(g)
    # crafted: the only two occurrences of "GT" (and of "AC", on the other strand) stand in front of a contig join
    cases.append(np.frombuffer(b"CCGT!AAAAGT!CCCC", np.uint8))
    assert O.Esa(cases[-1]).cache_quirks() > 0
    # ... and one where a longer shared context ("ACG!") keeps the walk out of the cache's depth: no bug
    cases.append(np.frombuffer(b"TTTTTTTTTTACG!TTTTTTTTTACG!TTTTTT", np.uint8))
    cases.append(synth.split_contigs(synth.random_base(60000, rng), 6, rng))
    for g in cases:
        want = O.Esa(g).cache_quirks() > 0
        got = E.cache_quirk(g)
        assert got == want, bytes(g[:80])
        seen[want] += 1
    assert seen[True] >= 5 and seen[False] >= 5, seen


def test_cache_quirk_is_reproduced():
    """On a subject whose 6-mer cache holds over-deep intervals the reference reports matches that are longer than
    the longest match (get_match_cached continues below the cached interval from its depth, whatever the query
    holds there: esa.cxx:174-199, 542-563).  The chains reproduce those answers (lean_core.h: LeanIndex::quirk):
    raw and filtered lists equal the oracle's on small multi-contig references, crafted and random; without the
    table (EMUL_NO_QUIRK) the true longest matches differ from the reference's on at least one of them."""
    rng = np.random.default_rng(4242)
    crafted = np.frombuffer(b"CCGT!AAAAGT!CCCC", np.uint8)
    sets = []
    # queries that walk into the over-deep keys: the reference's own nucleotides with the joins replaced
    q1 = np.frombuffer(b"CCGTAAAAAGTACCCCGTCAAAGTTCCCC" * 3, np.uint8)
    q2 = np.frombuffer(b"GTAGTCGTGGTTGTAAGTACGTCC" * 4, np.uint8)
    sets.append(([crafted, q1, q2, synth.revcomp(q1)], 0))
    found = 0
    for trial in range(3000):
        if found >= 25:
            break
        n = int(rng.integers(12, 300))
        g = synth.split_contigs(synth.random_base(n, rng), int(rng.integers(2, min(10, n // 3 + 2))), rng)
        if not O.Esa(g).cache_quirks():
            continue
        found += 1
        qs = [synth.mutate(np.where(g == ord("!"), rng.choice(np.frombuffer(b"ACGT", np.uint8), g.size), g), 0.05, rng) for _ in range(3)]
        qs.append(synth.random_base(int(rng.integers(50, 600)), rng))
        sets.append(([g] + qs, 0))
    assert found >= 10, found
    differs = 0
    for gs, ref in sets:
        assert E.cache_quirk(gs[ref])
        assert_same(gs, ref, threshold=int(rng.choice([0, 4, 6, 9])), chunk=int(rng.choice([0, 64, 128])))
        os.environ["EMUL_NO_QUIRK"] = "1"
        try:
            r = O.Run(gs, ref, threshold=4).process(compare=False)
            e = E.EmulRun(gs, ref, threshold=4, mode=1)
            for j in range(len(gs)):
                got = [tuple(int(x) for x in row) for row in e.raw(j)]
                want = [(int(a["iref"]), int(a["iq"]), int(a["len"])) for a in r.homologies(j, filtered=False)]
                differs += got != want
        finally:
            os.environ.pop("EMUL_NO_QUIRK", None)
    assert differs > 0

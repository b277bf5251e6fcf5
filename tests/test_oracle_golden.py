"""Pins the CPU oracle: reference unit-test literals (test/Tprocess.cxx,
test/Tsequence.cxx) and the known answers SURVEY.md §8c recorded from the
compiled reference on simf inputs (committed under tests/golden/)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O


def H(*rows):
    return O.homs(list(rows))


def same(a, b):
    """operator== of test/Tprocess.cxx:9-17: start, end, start_query, end_query."""
    if len(a) != len(b):
        return False
    for x, y in zip(a, b):
        if (x["iproj"], x["iproj"] + x["len"], x["iq"], x["iq"] + x["len"]) != \
           (y["iproj"], y["iproj"] + y["len"], y["iq"], y["iq"] + y["len"]):
            return False
    return True


# ── test/Tprocess.cxx:19-52 ──
def test_homology_basics():
    for A, B, C_ in ((H((0, 0, 10)), H((1, 1, 10)), H((10, 10, 10))),
                     (H((0, 23456, 10)), H((1, 678, 10)), H((10, 987, 10)))):
        assert O.hom_pred(A, B, "starts_left_of")
        assert not O.hom_pred(A, B, "ends_left_of")
        assert O.hom_pred(A, B, "overlaps")
        assert O.hom_pred(A, C_, "starts_left_of")
        assert O.hom_pred(A, C_, "ends_left_of")
        assert not O.hom_pred(A, C_, "overlaps")
    D = O.hom_trim(H((0, 0, 100)), 0, 10)
    assert same(D, H((0, 0, 10)))


# ── test/Tprocess.cxx:54-94 ──
def test_homology_filtering():
    pile = O.sort_filter(H((0, 0, 10), (1, 1, 3)), do_sort=False)
    assert same(pile, H((0, 0, 10)))
    expected = H((0, 0, 10), (10, 10, 20), (40, 40, 5))
    pile = O.sort_filter(H((0, 0, 10), (10, 10, 10), (10, 10, 20), (40, 40, 5)), do_sort=False)
    assert same(pile, expected)
    pile = O.sort_filter(H((0, 0, 10), (10, 10, 10), (10, 10, 20), (40, 40, 5), (42, 42, 2)), do_sort=False)
    assert same(pile, expected)
    pile = O.sort_filter(H((10, 10, 10), (0, 0, 10), (20, 20, 10), (5, 5, 10), (15, 15, 10), (25, 25, 10),
                           (30, 30, 10)), do_sort=True)
    assert same(pile, H((0, 0, 10), (10, 10, 10), (20, 20, 10), (30, 30, 10)))


# ── test/Tprocess.cxx:96-123 ──
def test_complete_deletion():
    hom = [H((10, 10, 10), (110, 110, 20), (220, 220, 10), (260, 260, 10)),
           H((10, 10, 10), (120, 120, 20), (200, 200, 100)),
           H((0, 0, 300), (300, 300, 100))]
    e = H((10, 10, 10), (120, 120, 10), (220, 220, 10), (260, 260, 10))
    got = O.complete_delete(hom)
    assert all(same(g, e) for g in got)
    got2 = O.complete_delete([e, e, e])
    assert all(same(g, e) for g in got2)
    hom = [H((10, 110, 10), (110, 210, 20), (220, 320, 10), (260, 460, 10)),
           H((10, 510, 10), (120, 620, 20), (200, 700, 100)),
           H((0, 0, 300), (300, 300, 100))]
    exp = [H((10, 110, 10), (120, 220, 10), (220, 320, 10), (260, 460, 10)),
           H((10, 510, 10), (120, 620, 10), (220, 720, 10), (260, 760, 10)),
           H((10, 10, 10), (120, 120, 10), (220, 220, 10), (260, 260, 10))]
    got = O.complete_delete(hom)
    assert all(same(g, x) for g, x in zip(got, exp))


# ── test/Tsequence.cxx:14-42 ──
def test_revcomp_literals():
    assert O.revcomp(b"") == b""
    for a, b in ((b"A", b"T"), (b"C", b"G"), (b"G", b"C"), (b"T", b"A"), (b"ACGTACGT", b"ACGTACGT")):
        assert O.revcomp(a) == b
    s = b"TACGATCGATCGAAAGCTAGTTCGCCCCGAGATA"
    assert O.revcomp(s) == b"TATCTCGGGGCGAACTAGCTTTCGATCGATCGTA"
    assert O.revcomp(O.revcomp(s)) == s


def test_filter_nucl_literals():
    for x in (b"", b"A", b"C", b"G", b"T"):
        assert O.filter_nucl(x) == x
    assert O.filter_nucl(b"!") == b""
    s = b"TACGATCGATCGAAAGCTAGTTCGCCCCGAGATA"
    assert O.filter_nucl(s) == s
    assert O.filter_nucl(b"tacgatc!gatc!gaa__agctagttcgcc#ccgagata") == s


# ── SURVEY §8c known answers from the compiled reference ──
@pytest.fixture(scope="module")
def known(golden_dir):
    with open(os.path.join(golden_dir, "known_answers.json")) as f:
        return json.load(f)


def _pair(golden_dir, pre):
    return [O.read_fasta_genome(os.path.join(golden_dir, f"{pre}{i}.fasta.gz")) for i in (0, 1)]


def test_known_answer_simple(golden_dir, known):
    g = _pair(golden_dir, "simple")
    r = O.Run(g, known["simple"]["ref"]).process()
    s, h = r.matrix()
    txt = O.phylip(["simple0", "simple1"], s, h)
    assert txt == "2\nsimple0  0.0000e+00  9.7004e-02\nsimple1  9.7004e-02  0.0000e+00\n"


def test_known_answer_cfg1(golden_dir, known):
    k = known["cfg1"]
    g = _pair(golden_dir, "cfg1_")
    assert len(g[0]) == 1000000 and len(g[1]) == 1000000
    r = O.Run(g, k["ref"]).process()
    assert r.threshold == k["threshold"]
    assert "%.17g" % r.gc == k["gc"]
    hv = r.homologies(0)
    assert len(hv) == k["n_homologies_q0"]
    assert int(hv["len"].sum()) == k["covered_q0"]
    assert all(d == 0 for d in hv["rev"])
    assert [[int(x["iref"]), int(x["iq"]), int(x["len"])] for x in hv[:3]] == k["first_homologies_q0"]
    s, h = r.matrix()
    assert int(s[0, 1]) == k["substitutions"] and int(h[0, 1]) == k["homologs"]
    assert s[1, 0] == s[0, 1] and h[1, 0] == h[0, 1] and s[0, 0] == 0 and h[1, 1] == 0
    assert "%.17g" % O.estimate("jc", s[0, 1], h[0, 1]) == k["jc"]
    names = ["cfg1_0", "cfg1_1"]
    assert O.phylip(names, s, h, "jc").split()[3] == k["phylip_jc"]
    assert O.phylip(names, s, h, "raw").split()[3] == k["phylip_raw"]
    assert O.phylip(names, s, h, "ani").split()[3] == k["phylip_ani"]
    assert O.phylip(names, s, h, "ani").split()[2] == "0"


def test_unrelated_gives_nan():
    rng = np.random.default_rng(5)
    a = rng.choice(np.frombuffer(b"ACGT", np.uint8), 50000)
    b = rng.choice(np.frombuffer(b"ACGT", np.uint8), 50000)
    r = O.Run([a, b], 0).process()
    s, h = r.matrix()
    assert h[0, 1] == 0
    txt = O.phylip(["a", "b"], s, h)
    assert txt.split("\n")[1].split()[2] == "nan"


@pytest.mark.slow
def test_known_answer_big(known):
    simf = os.path.join(O.ORACLE_DIR, "_ref", "simf")
    if not os.path.exists(simf):
        pytest.skip("oracle/_ref/simf not built (reference tree absent)")
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([simf, *known["big"]["simf"], "-p", os.path.join(tmp, "big_")])
        g = [O.read_fasta_genome(os.path.join(tmp, f"big_{i}.fasta")) for i in (0, 1)]
    r = O.Run(g, 1).process()
    s, h = r.matrix()
    assert int(h[0, 1]) == known["big"]["homologs"]
    assert O.phylip(["a", "b"], s, h).split()[3] == known["big"]["phylip_jc"]

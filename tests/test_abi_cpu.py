"""CPU-side checks of the product library: it loads, exports every symbol the
header declares, refuses to run without a GPU (no fallback), and its host-side
helpers agree with the oracle."""
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from phylonium_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(api.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return api.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "phylonium_amd.h")).read()
    declared = set(re.findall(r"\b(phylo_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.SYMBOLS), declared ^ set(api.SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.PhyloniumError):
        api.Context(0)


def test_host_suffix_array(lib):
    rng = np.random.default_rng(1)
    for n in (1, 7, 1000, 30000):
        s = synth.split_contigs(synth.random_base(n, rng), 3 if n > 100 else 1, rng).tobytes()
        S = s + b"#" + O.revcomp(s)
        assert (api.host_suffix_array(S) == O.suffix_array(S)).all()


def test_host_threshold(lib):
    for gc in (0.3, 0.5, 0.65):
        for l in (2001, 200001, 2000001, 10000001):
            assert api.host_min_anchor_length(0.025, gc, l) == O.min_anchor_length(0.025, gc, l)


def test_host_sort_filter_matches_reference_dp(lib):
    """O(n log n) chain filter == the reference's O(n²) DP, ties included."""
    rng = np.random.default_rng(3)
    for trial in range(300):
        n = int(rng.integers(0, 60))
        h = np.zeros(n, api.PHOM)
        o = np.zeros(n, O.HOM_DTYPE)
        # small coordinate range → many overlaps and equal starts / scores
        start = rng.integers(0, 80, n)
        ln = rng.integers(1, 25, n)
        iq = rng.integers(0, 1000, n)
        for t in range(n):
            h[t] = (start[t], start[t], iq[t], ln[t], 0, 0)
            o[t] = (0, start[t], start[t], iq[t], ln[t])
        got = api.host_sort_filter(h)
        want = O.sort_filter(o)
        assert [(int(a["index_reference_projected"]), int(a["index_query"]), int(a["length"])) for a in got] == \
               [(int(a["iproj"]), int(a["iq"]), int(a["len"])) for a in want]


def test_host_sort_filter_packed_path_distinct_starts(lib):
    """Distinct starts take the packed-key path (no std::sort on structs): equal ends and
    equal scores must still pick the reference's predecessor."""
    rng = np.random.default_rng(5)
    for trial in range(400):
        n = int(rng.integers(0, 70))
        start = rng.choice(120, n, replace=False)
        ln = rng.integers(1, 30, n)
        iq = rng.integers(0, 1000, n)
        h = np.zeros(n, api.PHOM)
        o = np.zeros(n, O.HOM_DTYPE)
        for t in range(n):
            h[t] = (start[t], start[t], iq[t], ln[t], 0, 0)
            o[t] = (0, start[t], start[t], iq[t], ln[t])
        got = api.host_sort_filter(h)
        want = O.sort_filter(o)
        assert [(int(a["index_reference_projected"]), int(a["index_query"]), int(a["length"])) for a in got] == \
               [(int(a["iproj"]), int(a["iq"]), int(a["len"])) for a in want]


def test_estimates_and_phylip(lib):
    for s, h in ((0, 0), (0, 10), (5, 100), (89758, 972512), (75, 100), (80, 100)):
        for kind in ("jc", "raw", "ani"):
            a, b = api.estimate(kind, s, h), O.estimate(kind, s, h)
            assert (np.isnan(a) and np.isnan(b)) or a == b  # same libm, same integers ⇒ identical doubles
    S = np.array([[0, 89758, 3], [89758, 0, 0], [3, 0, 0]], np.uint64)
    H = np.array([[0, 972512, 40], [972512, 0, 0], [40, 0, 0]], np.uint64)
    names = ["a", "bb", "c c"]
    for kind in ("jc", "raw", "ani"):
        assert api.format_phylip(names, S, H, kind) == O.phylip(names, S, H, kind)

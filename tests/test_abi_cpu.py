"""CPU-side checks of the product library: it loads, exports every symbol the
header declares, refuses to run without a GPU (no fallback), and its host-side
helpers agree with the oracle."""
import os
import re
import sys

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


def test_b0_library_exports_the_reference_names(lib):
    """Seam B0 under the reference's own names (libs/seqcmp.h:14, libs/revseqcmp.h:25) lives in its own
    small library; it must load and export both (no compute call without a GPU)."""
    import ctypes as C
    b0 = C.CDLL(os.path.join(os.path.dirname(api.LIB_PATH), "libphylonium_amd_b0.so"))
    assert hasattr(b0, "seqcmp") and hasattr(b0, "revseqcmp")


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
    # a matrix large enough for several row blocks (formatted on several threads, joined in order), with nan cells
    rng = np.random.default_rng(2)
    n = 83
    H = rng.integers(0, 5_000_000, (n, n)).astype(np.uint64)
    H[rng.random((n, n)) < 0.05] = 0
    S = (H * rng.random((n, n))).astype(np.uint64)
    names = [f"genome_{i}" for i in range(n)]
    for kind in ("jc", "raw", "ani"):
        assert api.format_phylip(names, S, H, kind) == O.phylip(names, S, H, kind)


def test_host_fasta_reader_and_reference_choice(lib, tmp_path):
    """Host helpers of the drivers (no GPU): FASTA → joined, filtered genomes (sequence.cxx:109-199) and the
    median-length reference choice (phylonium.cxx:360-382)."""
    (tmp_path / "x.fa").write_bytes(b">c1 comment\nACGTacgtNNRY\nGG-TT\n\n>c2\nTTTT\n>empty\n>c4\nA\n")
    (tmp_path / "y.fasta").write_bytes(b">only\nacgt")
    (tmp_path / "bad.fa").write_bytes(b"ACGT\n>late\nAC\n")
    (tmp_path / "none.fa").write_bytes(b"\n\n")
    gs = api.read_fasta([str(tmp_path / "x.fa"), str(tmp_path / "y.fasta")], threads=2)
    assert bytes(gs[0]) == b"ACGTACGTGGTT!TTTT!!A" and bytes(gs[1]) == b"ACGT"
    with pytest.raises(api.PhyloniumError, match="bad.fa: File is not in FASTA format."):
        api.read_fasta([str(tmp_path / "y.fasta"), str(tmp_path / "bad.fa"), str(tmp_path / "nope.fa")])
    with pytest.raises(api.PhyloniumError, match="none.fa: Empty file."):
        api.read_fasta([str(tmp_path / "none.fa")])
    with pytest.raises(api.PhyloniumError, match="nope.fa"):
        api.read_fasta([str(tmp_path / "nope.fa")])
    assert api.genome_name("a/b/x.fasta") == "x" and api.genome_name("x.fas") == "x" and api.genome_name("d.d/x.txt") == "x.txt"
    assert api.host_median_length_index([5, 1, 9]) == 0
    assert api.host_median_length_index([3, 8, 1, 9, 7]) == 4
    lens = [7, 7, 2, 7, 11, 7]
    assert lens[api.host_median_length_index(lens)] == 7
    S = b"ACGTTGCAAC!GGA"
    sa = api.host_reference_suffix_array(S)
    full = S + b"#" + O.revcomp(S)
    assert (sa == O.suffix_array(full)).all()


def _fasta_zoo(tmp_path, rng):
    """FASTA files that exercise the reader: line widths around the 32-byte pieces, lower case, IUPAC codes and
    gaps, CRLF, blank lines, many records, an empty record, no newline at the end."""
    def rnd(n, alphabet):
        return bytes(np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), n)])
    paths = []
    for k, (width, alpha, nrec, crlf) in enumerate([(70, b"ACGT", 1, False), (60, b"ACGTacgt", 3, False), (80, b"ACGTNRYacgtn-", 5, True),
                                                    (1, b"ACGT", 2, False), (31, b"ACGT", 2, False), (32, b"ACGT", 1, False),
                                                    (33, b"ACGTn", 4, False), (100000, b"ACGT", 2, False), (17, b"ACGT", 40, True)]):
        p = tmp_path / f"g{k}.fa"
        with open(p, "wb") as f:
            for r in range(nrec):
                f.write((b">rec%d some text ACGT" % r) + (b"\r\n" if crlf else b"\n"))
                n = int(rng.integers(0, 5000)) if r != 2 else 0
                s = rnd(n, alpha)
                for i in range(0, n, width):
                    f.write(s[i:i + width] + (b"\r\n" if crlf else b"\n"))
                if r == 1:
                    f.write(b"\n\n")
            if k == 5:
                f.seek(-1, 2)
                f.truncate()
        paths.append(str(p))
    return paths


def test_packed_fasta_reader_equals_the_byte_reader(lib, tmp_path):
    """phylo_host_read_fasta_packed (mapped files, 32 bytes at a time, 2-bit codes + separator positions) holds the
    same genomes as phylo_host_read_fasta (sequence.cxx:109-199), with the SIMD path and without it."""
    from phylonium_amd import api
    paths = _fasta_zoo(tmp_path, np.random.default_rng(5))
    ref = api.read_fasta(paths, threads=3)
    pk = api.read_fasta_packed(paths, threads=3)
    for k, (g, (q2, ln, bad)) in enumerate(zip(ref, pk)):
        assert ln == g.size and q2.size == (ln + 15) // 16, k
        assert np.array_equal(np.flatnonzero(g == ord("!")), bad), k
        shifts = np.arange(30, -2, -2, dtype=np.uint32)
        codes = ((q2[:, None] >> shifts[None, :]) & 3).reshape(-1)
        assert not codes[ln:].any() and not codes[bad].any(), k  # codes behind the end and under a separator are 0
        assert np.array_equal(api.unpack_genome(q2, ln, bad), g), k
    # the same through the byte-wise loop only (the switch is read once per process)
    code = ("import sys, pickle; sys.path.insert(0, %r); from phylonium_amd import api; "
            "pickle.dump(api.read_fasta_packed(%r, threads=2), sys.stdout.buffer)" % (ROOT, paths))
    import pickle
    import subprocess
    # (the switch exists in the development build only: `make dev`, -DPHY_DEV_HOOKS)
    dev_lib = os.path.join(ROOT, "phylonium_amd", "libphylonium_amd_dev.so")
    assert os.path.exists(dev_lib), "run __graft_entry__.build() (make dev) first"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True,
                         env=dict(os.environ, PHYLONIUM_AMD_NO_SIMD="1", PHYLONIUM_AMD_LIB=dev_lib), check=True)
    for a, b in zip(pk, pickle.loads(out.stdout)):
        assert a[1] == b[1] and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    # something that cannot be mapped and has no size (a FIFO) still reads: it gets a place of its own in the arena
    import threading
    fifo = str(tmp_path / "pipe.fa")
    os.mkfifo(fifo)
    text = open(paths[2], "rb").read()
    w = threading.Thread(target=lambda: open(fifo, "wb").write(text))
    w.start()
    got = api.read_fasta_packed([paths[0], fifo, paths[1]], threads=1)
    w.join()
    for a, b in zip(got, (pk[0], pk[2], pk[1])):
        assert a[1] == b[1] and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    # errors as the byte reader's
    (tmp_path / "bad.fa").write_bytes(b"hello\n>x\nACGT\n")
    (tmp_path / "none.fa").write_bytes(b"")
    with pytest.raises(api.PhyloniumError, match="bad.fa: File is not in FASTA format."):
        api.read_fasta_packed([paths[0], str(tmp_path / "bad.fa"), str(tmp_path / "nope.fa")])
    with pytest.raises(api.PhyloniumError, match="none.fa: Empty file."):
        api.read_fasta_packed([str(tmp_path / "none.fa")])
    with pytest.raises(api.PhyloniumError, match="nope.fa"):
        api.read_fasta_packed([str(tmp_path / "nope.fa")])


def test_hand_rolled_e4_format_equals_printf(tmp_path):
    """The matrix cells are written by phyfmt::e4 (host/fmt_e4.hpp) instead of printf("%.4e") (src/io.cxx:141-163's
    std::scientific, precision 4): ten million values — raw and Jukes-Cantor distances, powers of two times [0.5, 1.5),
    every decimal tie d.dddd5e+-x with three representable neighbours on either side, the specials — format the same."""
    import subprocess
    exe = str(tmp_path / "fmt_e4_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emul", "fmt_e4_check.cpp")], check=True)
    r = subprocess.run([exe, "3000000"], capture_output=True, text=True)
    assert r.returncode == 0 and int(r.stdout.strip()) > 10_000_000, r.stdout[-500:]

"""The host driver `phylonium-amd` (phylonium's command line over the C ABI):
stdout must be byte-identical to the PHYLIP text the reference prints
(src/io.cxx:141-163) — checked against the survey's known answers and against
the oracle on multi-contig inputs."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from phylonium_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "phylonium_amd", "phylonium-amd")
CLI_DEV = os.path.join(ROOT, "phylonium_amd", "phylonium-amd-dev")  # `make dev`: with the tests' switches (-DPHY_DEV_HOOKS)


def run(args, cwd):
    p = subprocess.run([CLI, *args], cwd=cwd, capture_output=True, text=True)
    return p.returncode, p.stdout, p.stderr


def write_fasta(path, genome, width=70):
    contigs = bytes(genome).split(b"!")
    with open(path, "wb") as f:
        for k, c in enumerate(contigs):
            f.write(b">contig%d some comment\n" % k)
            for i in range(0, len(c), width):
                f.write(c[i:i + width] + b"\n")


def test_known_answers_from_the_compiled_reference(tmp_path, golden_dir):
    for pre in ("simple", "cfg1_"):
        for i in (0, 1):
            with gzip.open(os.path.join(golden_dir, f"{pre}{i}.fasta.gz")) as f:
                open(tmp_path / f"{pre}{i}.fasta", "wb").write(f.read())
    rc, out, err = run(["simple0.fasta", "simple1.fasta"], tmp_path)
    assert rc == 0 and out == "2\nsimple0  0.0000e+00  9.7004e-02\nsimple1  9.7004e-02  0.0000e+00\n"
    rc, out, err = run(["-v", "cfg1_0.fasta", "cfg1_1.fasta"], tmp_path)
    assert rc == 0 and out == "2\ncfg1_0  0.0000e+00  9.8488e-02\ncfg1_1  9.8488e-02  0.0000e+00\n"
    # ref = cfg1_1; `-v`: avg coverage 0.972512, alignment 972512 1000000 0.972512
    assert "chosen reference: cfg1_1" in err and "ref: cfg1_1" in err
    assert "avg coverage:\t0.972512" in err and "alignment:\t972512\t1000000\t0.972512" in err
    rc, out, err = run(["--distance=raw", "cfg1_0.fasta", "cfg1_1.fasta"], tmp_path)
    assert out.split()[3] == "9.2295e-02"
    rc, out, err = run(["--distance=ani", "cfg1_0.fasta", "cfg1_1.fasta"], tmp_path)
    assert out.split()[3] == "90.77" and out.split()[2] == "0"


def test_multi_contig_against_oracle(tmp_path):
    gs = synth.make_genomes(6, 40000, seed=81, d_range=(0.01, 0.2), indel_per_mbp=300, inv_frac=0.08, contigs=4,
                            inv_len=(200, 1500))
    names = [f"g{i}" for i in range(6)]
    for n, g in zip(names, gs):
        write_fasta(tmp_path / f"{n}.fa", g)
    files = [f"{n}.fa" for n in names]
    for ref in (0, 3):
        r = O.Run(gs, ref).process()
        s, h = r.matrix()
        rc, out, err = run(["-r", files[ref], *files], tmp_path)
        assert out == O.phylip(names, s, h)
        assert rc == 0
    r = O.Run(gs, 2).process(complete_deletion=True)
    s, h = r.matrix()
    rc, out, err = run(["--complete-deletion", "-r", files[2], *files], tmp_path)
    assert out == O.phylip(names, s, h)
    rc, out, err = run(["-p", "pos.txt", "-r", files[2], *files], tmp_path)
    assert out == O.phylip(names, s, h)
    # the -p file byte for byte against the restated process.cxx:471-513, 665-723 (forward and reverse blocks)
    want = r.positions_text()
    assert want.count(">part") > 3
    assert (tmp_path / "pos.txt").read_text() == want


def test_packed_and_byte_ingest_print_the_same(tmp_path):
    """The driver's default ingest (2-bit codes made while the mapped files are read, unpacked on the device) and
    --ingest=bytes agree on stdout, stderr and the -p file — lower case, IUPAC codes, CRLF and ragged lines included —
    and the first-pass reference (median length, first genome equal to it) is chosen from the packed form."""
    rng = np.random.default_rng(17)
    gs = synth.make_genomes(5, 30000, seed=9, d_range=(0.01, 0.15), indel_per_mbp=300, inv_frac=0.05, contigs=3, inv_len=(200, 900))
    names = [f"g{i}" for i in range(5)]
    for k, (n, g) in enumerate(zip(names, gs)):
        contigs = bytes(g).split(b"!")
        with open(tmp_path / f"{n}.fa", "wb") as f:
            for r, c in enumerate(contigs):
                f.write(b">c%d\r\n" % r if k == 1 else b">c%d\n" % r)
                if k == 2:
                    c = c.lower()
                width, i = (61, 0)
                while i < len(c):
                    w = int(rng.integers(1, 90)) if k == 3 else width
                    line = c[i:i + w]
                    if k == 4 and i % 7 == 0:
                        line = line[:w // 2] + b"NNRY-" + line[w // 2:]  # dropped by the filter (sequence.cxx:109-146)
                    f.write(line + (b"\r\n" if k == 1 else b"\n"))
                    i += w
    files = [f"{n}.fa" for n in names]
    for extra in ([], ["-v"], ["-r", files[3]], ["--complete-deletion"]):
        a = run([*extra, *files], tmp_path)
        b = run(["--ingest=bytes", *extra, *files], tmp_path)
        assert a == b and a[1].startswith("5\n")
        assert run(["--sa=host", *extra, *files], tmp_path) == a  # the suffix array from the host cores (default: the device)
    a = run(["-p", "pa.txt", *files], tmp_path)
    b = run(["--ingest=bytes", "-p", "pb.txt", *files], tmp_path)
    assert a == b and (tmp_path / "pa.txt").read_bytes() == (tmp_path / "pb.txt").read_bytes()
    assert (tmp_path / "pa.txt").read_text().count(">part") > 3
    # against the oracle on the same genomes (the filter must have removed exactly the inserted codes)
    for ref in range(5):
        sm, hm = O.Run(gs, ref).process().matrix()
        assert run(["-r", files[ref], *files], tmp_path)[1] == O.phylip(names, sm, hm)


def test_bootstrap_matrices_with_a_fixed_seed(tmp_path):
    """-b N: N-1 more matrices whose substitutions are redrawn from Binomial(homologs, s/h) per cell
    (evo_model.cxx:136-147, io.cxx:192-203).  The reference seeds its mt19937 from random_device; with
    PHYLONIUM_AMD_SEED the driver's engine is seeded like the oracle's and the text must be identical."""
    gs = synth.make_genomes(5, 30000, seed=9, d_range=(0.02, 0.2), indel_per_mbp=300, inv_frac=0.05)
    names = [f"b{i}" for i in range(5)]
    for n, g in zip(names, gs):
        write_fasta(tmp_path / f"{n}.fa", g)
    files = [f"{n}.fa" for n in names]
    s, h = O.Run(gs, 1).process().matrix()
    env = dict(os.environ, PHYLONIUM_AMD_SEED="4242")
    r = subprocess.run([CLI, "-b", "4", "-r", files[1], *files], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    boot = O.bootstrap(4242, 3, s, h)
    want = O.phylip(names, s, h) + "".join(O.phylip(names, boot[k], h) for k in range(3))
    assert r.stdout == want
    assert (boot[0] != s).any()  # the resampling does change cells


def test_unrelated_pair_reports_nan_and_fails(tmp_path):
    rng = np.random.default_rng(5)
    write_fasta(tmp_path / "a.fasta", synth.random_base(50000, rng))
    write_fasta(tmp_path / "b.fasta", synth.random_base(50000, rng))
    rc, out, err = run(["a.fasta", "b.fasta"], tmp_path)
    assert rc == 1 and "reported as nan" in err
    assert out.split("\n")[1].split()[2] == "nan"


def test_two_pass_picks_the_central_genome(tmp_path):
    gs = synth.make_genomes(5, 30000, seed=83, d_range=(0.02, 0.15))
    names = [f"t{i}" for i in range(5)]
    for n, g in zip(names, gs):
        write_fasta(tmp_path / f"{n}.fasta", g)
    files = [f"{n}.fasta" for n in names]
    rc, out, err = run(["--2pass", "-v", "-r", files[1], *files], tmp_path)
    refs = [l.split(": ")[1] for l in err.splitlines() if l.startswith("ref: ")]
    assert refs[0] == "t1" and len(refs) == 2
    s1, h1 = O.Run(gs, 1).process().matrix()
    jc = np.array([[O.estimate("jc", s1[i, j], h1[i, j], True) for j in range(5)] for i in range(5)])
    second = int(np.argmin(jc.sum(axis=1)))
    assert refs[1] == names[second]
    s2, h2 = O.Run(gs, second).process().matrix()
    assert out == O.phylip(names, s2, h2)


def test_fasta_reader_semantics(tmp_path):
    """sequence.cxx:109-199 / io.cxx:36-104 as the driver restates them: ACGT kept (lower case
    upper-cased), everything else dropped, records joined by '!', CRLF and blank lines harmless."""
    rng = np.random.default_rng(9)
    a = synth.random_base(30000, rng)
    b = synth.mutate(a, 0.05, rng) if hasattr(synth, "mutate") else a.copy()
    sa, sb = bytes(a), bytes(b)
    # plain files
    write_fasta(tmp_path / "a.fa", a)
    write_fasta(tmp_path / "b.fa", b)
    rc0, want, _ = run(["a.fa", "b.fa"], tmp_path)
    assert rc0 == 0
    # the same sequences dressed up: lower case, CRLF, Ns and digits sprinkled in, leading blank lines
    def dress(seq, path):
        with open(path, "wb") as f:
            f.write(b"\n  \r\n")
            f.write(b">rec one\r\n")
            low = seq.lower()
            for i in range(0, len(low), 61):
                f.write(low[i:i + 30] + b"N7-" + low[i + 30:i + 61] + b"\r\n")
    dress(sa, tmp_path / "a2.fa")
    dress(sb, tmp_path / "b2.fa")
    rc, out, _ = run(["a2.fa", "b2.fa"], tmp_path)
    assert rc == 0 and out.replace("a2", "a").replace("b2", "b") == want
    # two records, the second one empty: the separator is still there (same result as via the oracle)
    with open(tmp_path / "c.fa", "wb") as f:
        f.write(b">x\n" + sa[:15000] + b"\n>y\n" + sa[15000:] + b"\n>empty\n")
    gs = [np.frombuffer(sa[:15000] + b"!" + sa[15000:] + b"!", np.uint8), np.frombuffer(sb, np.uint8)]
    s, h = O.Run(gs, 1).process().matrix()
    rc, out, _ = run(["-r", "b.fa", "c.fa", "b.fa"], tmp_path)
    assert out == O.phylip(["b", "c"], s[::-1, ::-1], h[::-1, ::-1])  # -r sorts the names: b, c


def test_fasta_reader_errors_in_command_line_order(tmp_path):
    rng = np.random.default_rng(10)
    write_fasta(tmp_path / "ok.fa", synth.random_base(5000, rng))
    (tmp_path / "notfasta.fa").write_bytes(b"hello\n>x\nACGT\n")
    (tmp_path / "empty.fa").write_bytes(b"\n\n")
    rc, out, err = run(["ok.fa", "notfasta.fa", "empty.fa"], tmp_path)
    assert rc == 1 and out == "" and "notfasta.fa: File is not in FASTA format." in err and "empty.fa" not in err
    rc, out, err = run(["ok.fa", "empty.fa", "notfasta.fa"], tmp_path)
    assert rc == 1 and "empty.fa: Empty file." in err and "notfasta" not in err
    rc, out, err = run(["ok.fa", "missing.fa"], tmp_path)
    assert rc == 1 and "missing.fa" in err


@pytest.mark.parametrize("backend", ["auto", "copies"])
def test_several_gpus_print_the_same(tmp_path, backend):
    """`phylonium-amd --gpus N` (one host thread and one context per rank, csrc/group.hip: the genomes' blocks
    all-gathered, phase A by query block, the lists exchanged as device blocks, phase B by window range, the u32
    triangles all-reduced, every rank's device writing its rows of the result) prints what one GPU prints — stdout, warnings and exit status — for N = 1, 2, 3, 5,
    with a reference given or chosen, complete deletion, -p and the two-pass mode; on a box with fewer GPUs than
    ranks the ranks share them and the exchange is device-to-device copies."""
    env = dict(os.environ)
    exe = CLI
    if backend == "copies":  # (a switch of the development build)
        env["PHYLONIUM_AMD_GROUP_BACKEND"] = "copies"
        exe = CLI_DEV

    def run_env(args):
        p = subprocess.run([exe, *args], cwd=tmp_path, capture_output=True, text=True, env=env)
        return p.returncode, p.stdout, p.stderr

    gs = synth.make_genomes(11, 30000, seed=83, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.08, contigs=3,
                            inv_len=(200, 1500))
    gs.append(synth.random_base(2000, np.random.default_rng(5)))  # unrelated: nan distances, a warning, exit status 1
    names = [f"m{i:02d}" for i in range(len(gs))]
    for n, g in zip(names, gs):
        write_fasta(tmp_path / f"{n}.fa", g)
    files = [f"{n}.fa" for n in names]
    so, ho = O.Run(gs, 4).process().matrix()
    want = O.phylip(names, so, ho)
    for extra in (["-r", files[4]], [], ["--complete-deletion", "-r", files[2]], ["-2"], ["--sa=host", "-r", files[7]]):
        one = run([*extra, *files], tmp_path)
        if extra == ["-r", files[4]]:
            assert one[1] == want and one[0] == 1
        for n in (1, 2, 3, 5):
            got = run_env(["--gpus", str(n), *extra, *files])
            assert got == one, (extra, n, got[2][-300:])
    one = run(["-p", "p1.txt", "-r", files[2], *files], tmp_path)
    got = run_env(["--gpus", "3", "-p", "p3.txt", "-r", files[2], *files])
    assert got == one and (tmp_path / "p1.txt").read_bytes() == (tmp_path / "p3.txt").read_bytes()
    rc, out, err = run_env(["--gpus", "3", "--timing", "-r", files[4], *files])
    assert out == want and "3 ranks over" in err
    # --bench-steps K: K more passes (a rank's pass as one queue, the result left in the group's home), one JSON line
    import json
    for n in (1, 3, 8):
        rc, out, err = run_env([*(["--gpus", str(n)] if n > 1 else []), "--bench-steps", "4", "-r", files[4], *files])
        line = [l for l in err.splitlines() if l.startswith("bench-steps: ")]
        assert rc == 1 and out == want and len(line) == 1, err[-2000:]
        b = json.loads(line[0][len("bench-steps: "):])
        assert b["steps"] == 4 and b["ranks"] == n and b["identical_to_printed_matrix"] and b["ms_per_step"] > 0 and len(b["per_rank_ms"]) == n
        if n > 1:
            assert b["shared_result"] and b["result_in_place"] and b["passes_repeated"] == 0
            assert all(0 < x["queued"] <= x["step"] for x in b["per_rank_ms"])


@pytest.mark.parametrize("ranks", [2, 8])
def test_verify_ranks_of_the_driver(tmp_path, ranks):
    """`phylonium-amd --gpus N --verify-ranks`: after printing, the reference's row is recomputed through the B0 kernels and
    a sub-matrix of up to 32 genomes (the first genome of every rank's block among them) is compared with a one-GPU run of
    those genomes; one JSON line on stderr carries the verdict and every rank's timings.  Passes for 2 and 8 ranks sharing
    the test box's GPU with the matrix unchanged; a rank that sends a damaged record (PHYLONIUM_AMD_TEST_CORRUPT_RANK, development build) is
    caught — exit status 3, the damaged pairs named."""
    import json
    gs = synth.make_genomes(40, 20000, seed=97, d_range=(0.01, 0.2), indel_per_mbp=300, inv_frac=0.05, contigs=2, inv_len=(200, 1500))
    names = [f"v{i:02d}" for i in range(len(gs))]
    for n, g in zip(names, gs):
        write_fasta(tmp_path / f"{n}.fa", g)
    files = [f"{n}.fa" for n in names]
    plain = run(["--gpus", str(ranks), "-r", files[3], *files], tmp_path)
    rc, out, err = run(["--gpus", str(ranks), "--verify-ranks", "-r", files[3], *files], tmp_path)
    assert (rc, out) == plain[:2], err[-2000:]
    line = [l for l in err.splitlines() if l.startswith("verify-ranks: ")]
    assert len(line) == 1, err[-2000:]
    v = json.loads(line[0][len("verify-ranks: "):])
    assert v["ok"] and v["n_ranks"] == ranks and len(v["ranks"]) == ranks and v["submatrix"]["identical"] and not v["reference_row"]["mismatching"]
    assert v["submatrix"]["genomes"] >= min(32, len(gs))
    env = dict(os.environ, PHYLONIUM_AMD_TEST_CORRUPT_RANK=str(ranks - 1))
    # the shipped driver has no such switch: the matrix is unchanged and the check passes ...
    p = subprocess.run([CLI, "--gpus", str(ranks), "--verify-ranks", "-r", files[3], *files], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert (p.returncode, p.stdout) == plain[:2], p.stderr[-2000:]
    # ... the development build's driver damages the record
    p = subprocess.run([CLI_DEV, "--gpus", str(ranks), "--verify-ranks", "-r", files[3], *files], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert p.returncode == 3, p.stderr[-2000:]
    v = json.loads([l for l in p.stderr.splitlines() if l.startswith("verify-ranks: ")][0][len("verify-ranks: "):])
    assert not v["ok"] and not v["submatrix"]["identical"] and v["submatrix"]["first_mismatching_pairs"]

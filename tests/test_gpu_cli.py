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
    assert out == O.phylip(names, s, h) and (tmp_path / "pos.txt").read_text().startswith(">part1\t(")


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

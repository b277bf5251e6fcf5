#!/usr/bin/env python3
"""Regenerates the committed golden fixtures. Run in the build container only
(needs oracle/_ref/simf, i.e. /root/reference/test/simf.cxx compiled by
oracle/Makefile).  simf's output depends on libstdc++'s default_random_engine
and distribution algorithms, so the FASTA files are committed, not regenerated
on the GPU box.

Outputs (tests/golden/):
  simple{0,1}.fasta.gz   simf -s 7  -l 100000   (test/simple.sh's shape, fixed seed)
  cfg1_{0,1}.fasta.gz    simf -s 42 -l 1000000  (BASELINE.json configs[0])
  known_answers.json     the numbers SURVEY.md §8c recorded from the compiled
                         reference for those inputs (and for simf -s 99 -l 5000000 -d 0.05,
                         whose 10 MB of FASTA is regenerated on demand, not committed)
"""
import gzip, json, os, shutil, subprocess, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SIMF = os.path.join(HERE, "..", "..", "oracle", "_ref", "simf")

KNOWN = {
    "source": "SURVEY.md §8c — reference v1.7 compiled and run in the build container during the survey",
    "cfg1": {"simf": ["-s", "42", "-l", "1000000"], "ref": 1, "threshold": 14, "esa_size": 2000001,
             "gc": "0.50034999999999996", "n_homologies_q0": 334, "covered_q0": 972512,
             "first_homologies_q0": [[0, 0, 852], [936, 936, 478], [1518, 1518, 1428]],
             "substitutions": 89758, "homologs": 972512, "jc": "0.098487533817159911",
             "phylip_jc": "9.8488e-02", "phylip_raw": "9.2295e-02", "phylip_ani": "90.77"},
    "simple": {"simf": ["-s", "7", "-l", "100000"], "ref": 1, "phylip_jc": "9.7004e-02"},
    "big": {"simf": ["-s", "99", "-l", "5000000", "-d", "0.05"], "ref": 1, "phylip_jc": "4.9373e-02",
            "homologs": 4956628, "committed": False},
}


def main():
    tmp = tempfile.mkdtemp()
    try:
        for name, pre in (("cfg1", "cfg1_"), ("simple", "simple")):
            subprocess.check_call([SIMF, *KNOWN[name]["simf"], "-p", os.path.join(tmp, pre)])
            for i in (0, 1):
                with open(os.path.join(tmp, f"{pre}{i}.fasta"), "rb") as f, \
                        gzip.GzipFile(os.path.join(HERE, f"{pre}{i}.fasta.gz"), "wb", mtime=0) as g:
                    g.write(f.read())
        with open(os.path.join(HERE, "known_answers.json"), "w") as f:
            json.dump(KNOWN, f, indent=1)
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()

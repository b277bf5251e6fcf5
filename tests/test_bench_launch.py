"""bench.py's own N-rank launcher (`python bench.py --gpus N` with no torchrun around it)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_launcher_starts_n_ranks_and_reports_their_failure_without_a_gpu():
    """No GPU here: every rank must stop with the 'needs a GPU' message (there is no CPU
    fallback), the launcher must come back with a nonzero status and print no JSON line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu-marked test below")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "small", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert r.stderr.count("needs a GPU") == 2


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_started_by_bench_itself_equal_one_rank(tmp_path):
    """`python bench.py --gpus 2` as the driver runs it (no launcher, WORLD_SIZE unset): two rank
    processes on the one test GPU (collectives over gloo, as the line says), n_gpus == 2, and the
    sharded matrices equal the one-rank ones."""
    outs = {}
    for n in (1, 2):
        dump = str(tmp_path / f"m{n}.npz")
        r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--workload", "small", "--steps", "2", "--warmup", "1",
                            "--cpu-sample", "0", "--dump-matrix", dump], capture_output=True, text=True, timeout=500, env=_env())
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout
        line = json.loads(lines[0])
        assert line["n_gpus"] == n and line["steps"] == 2
        assert line["config"]["workload"].startswith("small")
        if n == 2:
            assert "gloo" in line["config"]["backend"] or "nccl" in line["config"]["backend"]
            # the scaling harness: every rank's own step time, and the same workload on ONE GPU measured in the same job
            assert len(line["ranks_ms_per_step"]) == 2 and line["one_gpu_same_workload_ms"] > 0 and line["speedup_same_workload"] > 0
            assert line["one_gpu_same_workload_identical"] is True
            assert line["rccl_ranks"] in (None, 2)
        else:
            assert line["ranks_ms_per_step"] is None and line["one_gpu_same_workload_ms"] is None
            # seam B0 measured live on the workload's own segments: the reference's row recomputed by the byte kernels
            assert line["roofline_b0"]["row_identical_to_matrix"] is True and 0 < line["roofline_b0"]["frac"] <= 1
        # no fraction of the line exceeds 1: what SURVEY 8d's bytes cannot express (phase B, the whole path) is null, not 42
        def fracs(o, path=""):
            if isinstance(o, dict):
                for k, v in o.items():
                    if "frac" in k and isinstance(v, (int, float)):
                        yield path + "/" + k, v
                    yield from fracs(v, path + "/" + k)
        assert all(0 <= v <= 1.0 for _, v in fracs(line)), [kv for kv in fracs(line) if not 0 <= kv[1] <= 1.0]
        assert line["roofline_path"]["frac"] is None and line["roofline_phase_b"]["hbm_frac_on_alg_bytes"] is None
        assert line["roofline_phase_b"]["bound"] == "mfma"
        outs[n] = np.load(dump)
    assert (outs[1]["subst"] == outs[2]["subst"]).all() and (outs[1]["homologs"] == outs[2]["homologs"]).all()
    assert outs[1]["homologs"].sum() > 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("ranks", [2, 8])
def test_verify_ranks_passes_and_catches_a_damaged_block(ranks):
    """`bench.py --gpus N --verify-ranks`: rank 0 recomputes the reference's row through the B0 kernels and compares a
    sub-matrix of up to 32 genomes (the first genome of every rank's block among them) with a one-context run of those
    genomes; the verdict and every rank's timings are one JSON line on stderr.  It passes for 2 and 8 ranks (sharing the
    test box's GPU), and fails — exit status 3, the damaged pairs named — when a rank sends a damaged record."""
    base = [sys.executable, BENCH, "--gpus", str(ranks), "--workload", "small", "--steps", "2", "--warmup", "1", "--cpu-sample", "0",
            "--no-wallclock", "--verify-ranks"]
    r = subprocess.run(base, capture_output=True, text=True, timeout=800, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert line["verify_ranks"]["ok"] and line["verify_ranks"]["submatrix"]["identical"]
    rep = [l for l in r.stderr.splitlines() if l.startswith("# verify-ranks: ")]
    assert len(rep) == 1
    rep = json.loads(rep[0][len("# verify-ranks: "):])
    assert rep["n_ranks"] == ranks and len(rep["ranks"]) == ranks and all("kernels_ms" in x for x in rep["ranks"])
    r = subprocess.run(base + ["--test-corrupt-rank", str(ranks - 1)], capture_output=True, text=True, timeout=800, env=_env())
    assert r.returncode == 3, (r.returncode, r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    v = line["verify_ranks"]
    assert not v["ok"] and not v["submatrix"]["identical"] and v["submatrix"]["first_mismatching_pairs"]

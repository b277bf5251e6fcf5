#!/usr/bin/env python3
"""bench.py — throughput of the anchor + pairwise-compare path on MI355X.

One "step" = one pass of the hot path (phase A anchor chains + host sort/filter
+ phase B pair grid) over the resident synthetic genome set, i.e. one
`process(subject, queries)` of /root/reference/src/process.cxx:408-556 minus
the suffix-array build (which the north star keeps on the host and the metric
excludes).  Inputs (genomes, reference index) are resident in HBM before the
timed region.  Prints ONE JSON line (see the task contract).

Run:  python bench.py [--gpus N --steps K --warmup W --workload c3]
N>1:  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
      or plain `python bench.py --gpus N`: with WORLD_SIZE unset the script starts the N rank
      processes itself (fresh children, before anything here touches the GPU) and rank 0's
      line is the output.  Rank r runs on GPU r over RCCL; when the box has fewer GPUs than
      ranks (the one-GPU test box) the ranks share them and the same device-resident pass runs
      with its collectives staged through the host over gloo — the line then says so ("backend").
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (genomes, length, d_range, indel_per_mbp, inv_frac, description)
    "c1": (2, 1_000_000, (0.1, 0.1), 0, 0.0, "configs[0]: 2 x 1 Mbp, d=0.1, substitutions only"),
    "c2like": (29, 5_000_000, (0.0002, 0.03), 20, 0.005, "configs[1] stand-in (eco29 data absent): 29 x 5 Mbp, d in [0.0002,0.03]"),
    "c3": (256, 5_000_000, (0.01, 0.3), 100, 0.02, "configs[2]: 256 x 5 Mbp, d in [0.01,0.3] from the reference, 100 indels/Mbp, 2% inverted"),
    "c4": (1024, 5_000_000, (0.01, 0.3), 100, 0.02, "configs[3]: 1024 x 5 Mbp, same distribution as c3"),
    "c5s": (16, 20_000_000, (0.005, 0.1), 100, 0.10, "configs[4] scaled down: 16 x 20 Mbp, 50 contigs each, 10% inverted"),
    "c5": (64, 100_000_000, (0.005, 0.1), 100, 0.10, "configs[4]: 64 x 100 Mbp, 100 contigs each, 10% inverted"),
    "c3tree": (256, 5_000_000, (0.004, 0.04), 30, 0.005, "configs[2] on a tree (SURVEY 8d): 256 x 5 Mbp, genome k descends from genome "
               "(k-1)//2 with branch length U(0.004,0.04): distances to the reference (the root) up to ~0.2, between leaves up to ~0.35"),
    "small": (32, 1_000_000, (0.01, 0.3), 100, 0.02, "dev: 32 x 1 Mbp"),
    "c3dup": (64, 5_000_000, (0.01, 0.3), 100, 0.02, "dev: 64 x 5 Mbp as c3, but genomes 1-8 are byte-identical to the reference and "
              "genomes 9-16 differ from it in ~1 base per 10 kbp (matches far longer than a chunk: phase A's overrun path)"),
}
DUP = {"c3dup": (8, 8)}  # (byte-identical copies of the reference, near-identical genomes at d = 1e-4)
CONTIGS = {"c5s": 50, "c5": 100}
TREE = {"c3tree"}  # genomes mutated from their parent in a binary tree instead of all from genome 0
MULTI_GPU_WORKLOAD = "c4"  # the 1024-genome set the several-GPU target is quoted on (fits one GPU: 5.1 GB of genomes)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def make_genomes_gpu(torch, n, length, seed, device, d_range, indel_per_mbp, inv_frac, inv_len=(1000, 5000), contigs=1,
                     tree=False, dup=(0, 0)):
    """Synthetic genome set on the GPU. Genome 0 is the unmutated base (the reference,
    like simf's S0, test/simf.cxx:32); genome g>0 = base at JC distance d_g ~ U(d_range),
    plus indel events and inverted blocks; with `tree`, genome g descends from genome (g-1)//2
    instead (a binary tree rooted at genome 0), so pair distances span a range. Returns (buffer, offsets, lengths)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    base = torch.randint(0, 4, (length,), dtype=torch.uint8, generator=g, device=device)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    codes = []
    for k in range(n):
        if k == 0 or 1 <= k <= dup[0]:
            code = base
        else:
            d = float(rng.uniform(*d_range))
            if dup[0] < k <= dup[0] + dup[1]:
                d = 1e-4
            p = 0.75 - 0.75 * math.exp(-(4.0 / 3.0) * d)
            if tree:
                base = codes[(k - 1) // 2]
                length = int(base.numel())
            hit = torch.rand(length, generator=g, device=device) < p
            shift = torch.randint(1, 4, (length,), dtype=torch.uint8, generator=g, device=device)
            code = (base + hit.to(torch.uint8) * shift) & 3
            n_indel = int(round(indel_per_mbp * length / 1e6))
            n_inv = int(round(inv_frac * length / (0.5 * (inv_len[0] + inv_len[1]))))
            if dup[0] < k <= dup[0] + dup[1]:
                n_indel, n_inv = 2, 1
            if n_indel or n_inv:
                ev = [(int(x), 0) for x in rng.integers(0, length, n_indel)] + \
                     [(int(x), 1) for x in rng.integers(0, length, n_inv)]
                ev.sort()
                pieces, pos = [], 0
                for at, kind in ev:
                    if at < pos:
                        continue
                    pieces.append(code[pos:at])
                    if kind == 0:
                        ln = int(rng.integers(1, 51))
                        if rng.random() < 0.5:
                            pos = min(length, at + ln)
                        else:
                            pieces.append(torch.randint(0, 4, (ln,), dtype=torch.uint8, generator=g, device=device))
                            pos = at
                    else:
                        ln = int(rng.integers(inv_len[0], inv_len[1] + 1))
                        end = min(length, at + ln)
                        pieces.append(3 - torch.flip(code[at:end], dims=(0,)))  # reverse complement
                        pos = end
                pieces.append(code[pos:])
                code = torch.cat(pieces)
        codes.append(code)
    lens = [int(c.numel()) for c in codes]
    offs, tot = [], 64
    for l in lens:
        offs.append(tot)
        tot += (l + 63) // 64 * 64 + 64
    buf = torch.zeros(tot + 256, dtype=torch.uint8, device=device)  # +256: kernels prefetch 128-byte query windows
    for c, o, l in zip(codes, offs, lens):
        buf[o:o + l] = lut[c.long()]
        if contigs > 1:  # contig breaks: '!' replaces a base (src/sequence.cxx:171-199 joins contigs with '!')
            cuts = np.unique(rng.integers(1000, l - 1000, size=contigs - 1))  # (rng.choice without replacement shuffles all l positions)
            cuts = torch.from_numpy(cuts).to(device)
            buf[o + cuts] = ord("!")
    del codes
    return buf, offs, lens


def cpu_baseline(torch, buf, offs, lens, ref_idx, sa_ref, sample_queries, threads, reps=3):
    """The oracle (CPU port of the reference path, byte kernels bound as the reference's ifunc
    resolver would bind them on this CPU: libs/seqcmp.c:32-60) on a bounded sample of the same
    workload: the reference genome plus `sample_queries` others, anchor + compare timed, ESA
    construction excluded (as the suffix-array build is on the GPU side).  `reps` repetitions at
    the full thread count (median reported) and on one thread (a sixth of the sample)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    idx = [ref_idx] + [j for j in range(len(lens)) if j != ref_idx][:sample_queries]
    gs = [buf[offs[j]:offs[j] + lens[j]].cpu().numpy() for j in idx]
    variants = O.resolved_variants()

    def timed(genomes, nthreads):
        r = O.Run(genomes, 0)
        rates, last = [], None
        bases = float(sum(len(g) for g in genomes))
        for _ in range(reps):
            r.process(sa=sa_ref, threads=nthreads)
            t_esa, t_a, t_b = r.times()
            rates.append(bases / (t_a + t_b) / 1e9)
            last = (t_esa, t_a, t_b)
        r.close()
        return sorted(rates), last, bases

    rates, (t_esa, t_a, t_b), bases = timed(gs, threads)
    one_n = max(2, min(len(gs), 1 + max(1, sample_queries // 6)))
    rates1, (_, t_a1, t_b1), bases1 = timed(gs[:one_n], 1)
    med = rates[len(rates) // 2]
    return {"value": med, "unit": "Gbp/s", "cores": threads, "kind": "port",
            "one_thread": rates1[len(rates1) // 2], "runs": [round(x, 4) for x in rates], "runs_one_thread": [round(x, 4) for x in rates1],
            "seqcmp_variant": variants[0], "revseqcmp_variant": variants[1],
            "sample": f"oracle (C++ restatement of process.cxx/esa.cxx; seqcmp -> {variants[0]}, revseqcmp -> {variants[1]} "
                      f"as libs/seqcmp.c:32-60 / libs/revseqcmp.c:34-51 resolve on this CPU; OpenMP over queries and pair rows) on "
                      f"{len(gs)} of the workload's genomes ({bases / 1e6:.0f} Mbp), median of {reps} runs: last run anchor {t_a:.2f}s + compare "
                      f"{t_b:.2f}s; one thread on {one_n} genomes ({bases1 / 1e6:.0f} Mbp): anchor {t_a1:.2f}s + compare {t_b1:.2f}s; "
                      f"ESA build {t_esa:.1f}s excluded"}


def wallclock_leg(torch, holder, offs, lens, contigs_sep, want_text, n_gpus, runs=3, before_run=None):
    """BASELINE.json's second metric: wall-clock FASTA files -> PHYLIP text, through the C++ host driver
    (phylonium_amd/phylonium-amd, FASTA in / matrix out as src/phylonium.cxx:89-299) started as a fresh process.
    The workload's genomes are written as FASTA files (70 columns) to /dev/shm — outside every timed region — the
    driver runs `runs` times, and the median of the child's wall-clock (start of the process to its exit, as this
    parent sees it) is reported with the driver's own --timing split of the median run.  The text it prints must be
    the text made from this bench's result matrices."""
    import re
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "phylonium_amd", "phylonium-amd")
    if not os.path.exists(exe):
        return {"wallclock_s": None, "note": "phylonium_amd/phylonium-amd has not been built"}
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="phylonium_amd_bench_", dir=base)
    buf = holder[0]  # the genome buffer; `holder` is the caller's only reference to it
    try:
        t0 = time.time()
        files = []
        nl = torch.tensor([10], dtype=torch.uint8, device=buf.device)
        for j, (o, l) in enumerate(zip(offs, lens)):
            path = os.path.join(d, f"g{j:04d}.fasta")
            files.append(path)
            g = buf[o:o + l]
            with open(path, "wb") as f:
                if contigs_sep:
                    parts = bytes(g.cpu().numpy()).split(b"!")
                else:
                    parts = [g]
                for k, part in enumerate(parts):
                    f.write(b">contig%d\n" % k)
                    if not torch.is_tensor(part):  # a contig cut out on the host: plain bytes, 70 to a line
                        f.write(b"".join(part[i:i + 70] + b"\n" for i in range(0, len(part), 70)))
                        continue
                    a = part
                    full = a.numel() // 70 * 70
                    if full:
                        lines = torch.empty((full // 70, 71), dtype=torch.uint8, device=buf.device)
                        lines[:, :70] = a[:full].view(-1, 70)
                        lines[:, 70] = 10
                        f.write(lines.cpu().numpy().tobytes())
                    if full < a.numel():
                        f.write(torch.cat([a[full:], nl]).cpu().numpy().tobytes())
        t_write = time.time() - t0
        fasta_bytes = sum(os.path.getsize(f) for f in files)
        # this process lets go of its device memory before the driver starts (it keeps only an idle HIP context)
        del g, buf, parts, part
        a = None
        holder.clear()
        torch.cuda.empty_cache()
        if before_run:  # several ranks: the others have let go of their device memory too
            before_run()
        cmd = [exe, "--timing", "-r", files[0]] + (["--gpus", str(n_gpus)] if n_gpus > 1 else []) + files
        res = []
        for _ in range(runs):
            # (untimed: the driver clears the device memory the last process — this one, or the run before — gave back, and
            # a hipMalloc issued before it is done waits for it; somebody who runs the program once does not see that)
            time.sleep(2.0)
            t0 = time.perf_counter()
            pr = subprocess.run(cmd, capture_output=True)
            wall = time.perf_counter() - t0
            err = pr.stderr.decode(errors="replace")
            m = re.search(r"timing: (genomes .*)", err)
            res.append({"wall_s": round(wall, 4), "exit": pr.returncode, "timing": m.group(1) if m else err[-300:],
                        "matrix_identical": pr.stdout.decode(errors="replace") == want_text})
        # like for like with earlier rounds (and with somebody who runs the program in a loop): no pause, and the runtime's
        # orderly teardown instead of _exit — two runs back to back, the second one right behind the first one's exit
        b2b = []
        for _ in range(2):
            t0 = time.perf_counter()
            pr = subprocess.run(cmd[:1] + ["--teardown"] + cmd[1:], capture_output=True)
            b2b.append(round(time.perf_counter() - t0, 4))
            if pr.stdout.decode(errors="replace") != want_text:
                b2b[-1] = None
        med = sorted(res, key=lambda r: r["wall_s"])[len(res) // 2]
        split = {}
        for key, pat in (("inside_main", r"total ([0-9.]+) s"), ("read", r"\| read ([0-9.]+)"), ("wait_for_device", r"wait-for-device ([0-9.]+)"),
                         ("upload", r"upload ([0-9.]+)"), ("process_and_print", r"process\+print ([0-9.]+)"),
                         ("suffix_array", r"\[suffix array ([0-9.]+)"), ("rest_of_index", r"rest of the index ([0-9.]+)"),
                         ("anchor", r"anchor ([0-9.]+)"), ("compare", r"compare ([0-9.]+)\]")):
            mm = re.search(pat, med["timing"])
            if mm:
                split[key] = float(mm.group(1))
        return {"wallclock_s": med["wall_s"], "unit": "s", "n_gpus": n_gpus, "what": "FASTA files in /dev/shm -> PHYLIP text on stdout, "
                "`phylonium-amd --timing -r g0000.fasta <files>` as a fresh process, start to exit; median of %d runs" % runs,
                "runs_s": [r["wall_s"] for r in res], "matrix_identical": all(r["matrix_identical"] for r in res),
                "exit_status": med["exit"], "split_s": split,
                "pause_before_each_run_s": 2.0, "exit_mode": "_exit once the matrix is out (the driver's default)",
                "back_to_back_with_teardown_s": b2b, "fasta_bytes": fasta_bytes, "fasta_write_s": round(t_write, 2),
                "note": "exit status 1 is the reference's soft-warning status (src/io.cxx:106-139: a pair with < 20 % homology); "
                        "split_s is the driver's own --timing of the median run: process start-up and exit are the difference to wallclock_s"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def one_gpu_leg(torch, api, workload, seed, device, local, steps):
    """One GPU's step (both phases as one call, phylo_anchor_compare) on another workload's genomes: `scaling_baseline`."""
    n, length, d_range, indel, inv, desc = WORKLOADS[workload]
    buf, offs, lens = make_genomes_gpu(torch, n, length, seed, device, d_range, indel, inv, contigs=CONTIGS.get(workload, 1),
                                       tree=workload in TREE, dup=DUP.get(workload, (0, 0)))
    torch.cuda.synchronize()
    with api.Context(local) as c:
        c.set_genomes_device(buf.data_ptr(), offs, lens)
        c.set_reference(0)
        c.result_open(None, ranks=1)
        o = c.result_matrices()
        for _ in range(2):
            c.anchor_compare(out=o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            c.anchor_compare(out=o)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    bases = float(sum(lens))
    del buf
    torch.cuda.empty_cache()
    return {"workload": f"{workload}: {desc}", "n_gpus": 1, "steps": steps, "ms_per_step": round(ms, 3),
            "value": round(bases / (ms * 1e-3) / 1e9, 4), "unit": "Gbp/s",
            "note": "the workload every `--gpus N > 1` line runs, on this one GPU (no HIP events around any kernel): the one-GPU point "
                    "of the scaling curve; N > 1 lines carry the same measurement made in their own job (one_gpu_same_workload_ms)"}


def b0_leg(ctx, s, h, lens, ref_idx, max_queries=256, reps=5):
    """Seam B0 on the workload's own segments (the kernel north_star asks an HBM fraction for): the reference's row as
    evo_model::account / account_rev would call it (src/evo_model.cxx:53-75) — one seqcmp / revseqcmp per homology of up to
    `max_queries` queries, reference bytes against query bytes — through phylo_seqcmp_batch over the resident genomes.
    The kernels' time is the library's HIP-event span around them, measured live; 2 bytes per compared site over it is
    `achieved`; the tallies must add up to the matrix's row."""
    n = len(lens)
    qs = [j for j in range(n) if j != ref_idx][:max_queries]
    ga, oa, gb, ob, ln, rv, cuts = [], [], [], [], [], [], [0]
    for j in qs:
        hom = ctx.homologies(j)
        m = hom.size
        ga.append(np.full(m, ref_idx, np.uint32))
        oa.append(hom["index_reference_projected"].astype(np.uint64))
        gb.append(np.full(m, j, np.uint32))
        ob.append(hom["index_query"].astype(np.uint64))
        ln.append(hom["length"].astype(np.uint64))
        rv.append((hom["direction"] != 0).astype(np.uint8))
        cuts.append(cuts[-1] + m)
    ga, oa, gb, ob, ln, rv = (np.concatenate(x) for x in (ga, oa, gb, ob, ln, rv))
    ctx.set_option("profile", 1)
    for _ in range(2):
        sub = ctx.seqcmp_batch(ga, oa, gb, ob, ln, rv)
    ctx.reset_stats()
    for _ in range(reps):
        sub = ctx.seqcmp_batch(ga, oa, gb, ob, ln, rv)
    ms = ctx.stat("ms:seqcmp_batch") / max(1.0, ctx.stat("n:seqcmp_batch") or 1.0)
    ok = all(int(sub[cuts[t]:cuts[t + 1]].sum()) == int(s[ref_idx, j]) and int(ln[cuts[t]:cuts[t + 1]].sum()) == int(h[ref_idx, j])
             for t, j in enumerate(qs))
    sites = float(ln.sum())
    return {"bound": "hbm", "kernel": "seqcmp_rounds + seqcmp_pass (csrc/seqcmp_kernels.hip)", "achieved": round(2.0 * sites / (ms * 1e-3) / 1e9, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(2.0 * sites / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms, 4),
            "segments": int(ln.size), "mean_length": round(sites / max(1, ln.size), 1), "reverse_share": round(float(rv.mean()), 4) if rv.size else 0.0,
            "alg_bytes_per_launch": 2.0 * sites, "queries": len(qs), "row_identical_to_matrix": bool(ok),
            "note": "off the default path (phase B is the pileup): the reference's row of the result recomputed by the byte kernels over "
                    "each query's homologies, 2 B per compared site (SURVEY 8d) over the kernels' HIP-event span, measured in this run.  "
                    "Of the two strings of a segment the reference's is re-read by every query and stays in the L2 / Infinity Cache "
                    "(5 MB at c3), so the HBM interface carries about half of these bytes; segments between two queries at random "
                    "places, where nothing is re-read, run at 0.67 (profiles/r06_seqcmp_bw.json; PMC traffic 1.06 x the algorithmic "
                    "bytes: profiles/r06_seqcmp_pmc.txt)"}


def verify_ranks(torch, api, dist, ctx, s, h, buf, offs, lens, ref_idx, world, local):
    """The result of an N-rank run against two routes that share nothing with the exchange between the ranks:
    (1) the reference's row from rank 0's lists through seam B0 (seqcmp / revseqcmp over the resident genomes, summed over
        each query's list) — the lists every rank sent are what the tallies were made from;
    (2) a sub-matrix of up to 32 genomes — the reference, the first genome of every rank's block, the rest evenly spread —
        against a fresh one-context run of those genomes alone (a pair's tallies depend on the reference and the two
        genomes only): lists damaged or mixed up on their way between the ranks change it."""
    import hashlib
    n = len(lens)
    bounds = [dist.query_shard(n, r, world, lens)[0] for r in range(world)] + [n]
    row_sample = sorted({j for j in [1, n // 3, n // 2, n - 1] + bounds[:-1] if 0 <= j < n and j != ref_idx})[:12]
    row_bad = []
    for j in row_sample:
        hom = ctx.homologies(j)
        m = hom.size
        ln = hom["length"]
        sub = ctx.seqcmp_batch(np.full(m, ref_idx, np.uint32), hom["index_reference_projected"], np.full(m, j, np.uint32),
                               hom["index_query"], ln, (hom["direction"] != 0).astype(np.uint8))
        if int(sub.sum()) != int(s[ref_idx, j]) or int(ln.sum()) != int(h[ref_idx, j]):
            row_bad.append(j)
    want = min(n, 32)
    idx = {ref_idx} | {b for b in bounds[:-1] if b < n}
    for t in range(want):
        if len(idx) >= want:
            break
        idx.add(min(n - 1, (t * n) // want))
    idx = sorted(idx)[:max(want, 1)]
    if ref_idx not in idx:
        idx = sorted(idx[:-1] + [ref_idx])
    with api.Context(local) as c2:
        c2.set_genomes_device(buf.data_ptr(), [offs[j] for j in idx], [lens[j] for j in idx])
        s2, h2 = c2.process(idx.index(ref_idx))
    sub_s, sub_h = np.ascontiguousarray(s[np.ix_(idx, idx)]), np.ascontiguousarray(h[np.ix_(idx, idx)])
    digest = lambda a, b: hashlib.sha256(a.tobytes() + b.tobytes()).hexdigest()[:16]
    sub_ok = bool((sub_s == s2).all() and (sub_h == h2).all())
    bad_pairs = [] if sub_ok else [(idx[i], idx[j]) for i, j in zip(*np.nonzero((sub_s != s2) | (sub_h != h2)))][:8]
    return {"ok": not row_bad and sub_ok, "reference_row": {"genomes": row_sample, "mismatching": row_bad},
            "submatrix": {"genomes": len(idx), "sha256_n_ranks": digest(sub_s, sub_h), "sha256_one_context": digest(s2, h2),
                          "identical": sub_ok, "first_mismatching_pairs": [[int(a), int(b)] for a, b in bad_pairs]}}


def usable_cpus():
    """CPUs this process may actually burn: the cgroup CPU-time quota (cpu.max) when there is
    one — the GPU boxes show 256 CPUs but meter a job to a fraction of them — else cpu_count."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, min(n, int(int(q) / int(p)))), f"cgroup cpu.max = {int(q) / int(p):g} of {n} CPUs"
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, min(n, q // p)), f"cgroup cfs quota = {q / p:g} of {n} CPUs"
    except Exception:
        pass
    return n, f"{n} CPUs, no quota"


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (the same
    command line, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) as fresh
    children.  Nothing in this process has touched the GPU (torch is not even imported), the
    children are started with subprocess (no re-exec of a process that holds a device), rank 0
    inherits stdout so its one JSON line is this command's output, and the exit status is the
    worst of the ranks'."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    # all ranks are watched together: one that dies alone (out of memory, an import error) would leave the others
    # waiting inside a collective until its timeout; the first non-zero exit ends the rest, and so does the deadline
    rc, deadline = 0, time.time() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "3000"))
    try:
        live = list(procs)
        while live:
            for p in list(live):
                st = p.poll()
                if st is not None:
                    live.remove(p)
                    rc = max(rc, abs(st))
            if rc or time.time() > deadline:
                rc = rc or 124
                break
            if live:
                time.sleep(0.05)
        # (a rank has failed: the others a moment to fail the same way and say so themselves — all of them without a GPU,
        # say — before the ones stuck in a collective are ended)
        t_grace = time.time() + 2.0
        while rc != 124 and any(p.poll() is None for p in procs) and time.time() < t_grace:
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5
        for p in procs:
            while p.poll() is None and time.time() < t_end:
                time.sleep(0.05)
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c3 on one GPU (the configuration the 1-GPU metric is quoted on), c4 on several "
                         "(the 1024-genome set the multi-GPU target is quoted on)")
    ap.add_argument("--dump-matrix", default="", help="dev/tests: rank 0 saves the two N x N result matrices here (.npz)")
    ap.add_argument("--genomes", type=int, default=0, help="override genome count")
    ap.add_argument("--length", type=int, default=0, help="override genome length")
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--cpu-sample", type=int, default=63, help="queries in the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not time kernels with HIP events")
    ap.add_argument("--no-scaling-baseline", action="store_true", help="one rank on the default workload: skip the extra leg that times the "
                    "several-GPU workload (c4) on this one GPU (`scaling_baseline` in the line)")
    ap.add_argument("--no-b0", action="store_true", help="skip the seam-B0 leg (the byte kernels over the reference row's own segments: `roofline_b0`)")
    ap.add_argument("--no-wallclock", action="store_true", help="skip the FASTA -> PHYLIP wall-clock leg (the C++ host driver as a child process)")
    ap.add_argument("--check", action="store_true", help="verify a sample of the result against the oracle")
    ap.add_argument("--verify-ranks", action="store_true", help="after the timed steps rank 0 checks the N-rank result by two other routes — "
                    "the reference's row recomputed through the B0 kernels, and a sub-matrix of up to 32 genomes (the first genome of every "
                    "rank's block among them) against a one-context run of those genomes alone — and prints the verdict with every rank's "
                    "timings as one JSON line on stderr; exit status 3 when a check fails")
    ap.add_argument("--test-corrupt-rank", type=int, default=-1, help="tests of --verify-ranks: this rank sends one damaged record — its first "
                    "homology's query position moved by one base, a list as valid as any — into the exchange")
    ap.add_argument("--chunk", type=int, default=0, help="dev: force the phase-A chunk length")
    ap.add_argument("--filter", type=int, default=0, help="dev: sort + chain filter 1 on the host, 2 on the device (0: the library chooses)")
    ap.add_argument("--host-threads", type=int, default=0, help="dev: size of the library's host worker pool")
    ap.add_argument("--kmer", type=int, default=0, help="dev: force the bucket k of the reference index")
    ap.add_argument("--sa-builder", type=int, default=-1, help="dev: who builds the reference's suffix array: 1 the device, 0 the host cores (library default when < 0)")
    ap.add_argument("--fold-blocks", type=int, default=-1, help="dev: blocks per query of the fold kernel (0: the library chooses)")
    ap.add_argument("--pairs-wchunk", type=int, default=0, help="dev: windows per chunk of the pair kernel (library's choice when 0)")
    ap.add_argument("--d-range", default="", help="dev: lo,hi — override the workload's divergence range")
    ap.add_argument("--emulate-rank", default="", help="dev: R/N — time rank R of N's share of the work on this one GPU "
                    "(no collectives; the printed value is NOT a bench result)")
    ap.add_argument("--emulate-exchange", action="store_true", help="dev, with --emulate-rank: also pay the host side of "
                    "the two exchanges (export, one-rank RCCL collectives, import of the other ranks' lists, matrix all-reduce)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    default_workload = args.workload is None
    if args.workload is None:
        # (one workload for the whole scaling curve: every line of a several-GPU run carries the same workload's one-GPU step
        # measured in the same job — one_gpu_same_workload_ms, speedup_same_workload — and the one-GPU default run carries the
        # several-GPU workload's step on this GPU as `scaling_baseline`: a 1/2/4/8 curve is never c4 at N over c3 at 1)
        args.workload = "c3" if world_env == 1 else MULTI_GPU_WORKLOAD

    # stdout carries exactly one JSON line.  Libraries below (RCCL prints a version banner
    # through C stdio) write to fd 1 whenever they like: lend them stderr until the result
    # is ready, then flush C stdio and take fd 1 back.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as td
    from phylonium_amd import api, dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # one rank per GPU over RCCL; with fewer GPUs than ranks (the one-GPU test box) the ranks share
    # them and the exchange goes through the host over gloo (RCCL refuses two ranks on one device)
    shared = world > ndev
    backend = "gloo" if shared else "nccl"
    local = local % ndev
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 or args.emulate_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if shared:
            td.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            td.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)
    coll_dev = torch.device("cpu") if shared else device  # where the barrier / timing tensors of the collectives live
    if shared:
        # ranks sharing a GPU (RCCL refuses that): the same device-resident pass, its collectives staged through the host over
        # gloo on the same call sites and stream order (dist.HostStagedCollectives: a stand-in for tests, not a bench result)
        dist._COLL = dist.HostStagedCollectives()
    if world > 1 or args.emulate_exchange:
        # the library's kernels and torch's collectives on ONE stream of their own (the context is lent torch's current stream:
        # dist.process_sharded_device; handle 0 — the legacy default stream — would mean "the context's own stream", ordered with
        # torch's only by the default stream's implicit synchronisation)
        torch.cuda.set_stream(torch.cuda.Stream(device=device))

    n, length, d_range, indel, inv, desc = WORKLOADS[args.workload]
    n = args.genomes or n
    length = args.length or length
    if args.d_range:
        d_range = tuple(float(x) for x in args.d_range.split(","))
    t_gen = time.time()
    buf, offs, lens = make_genomes_gpu(torch, n, length, args.seed, device, d_range, indel, inv,
                                       contigs=CONTIGS.get(args.workload, 1), tree=args.workload in TREE,
                                       dup=DUP.get(args.workload, (0, 0)))
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    ref_idx = 0

    ctx = api.Context(local)
    # the timed steps carry HIP events around the kernel the roofline is reported for (the chain kernel: option profile = 2);
    # every other kernel's time comes from a pass of its own with events around all of them (~4 us each: 0.07 ms a step)
    ctx.set_option("profile", 0 if args.no_profile else 2)
    if args.chunk:
        ctx.set_option("chunk", args.chunk)
    if args.fold_blocks >= 0:
        ctx.set_option("fold_blocks", args.fold_blocks)
    if args.kmer:
        ctx.set_option("kmer", args.kmer)
    if args.pairs_wchunk > 0:
        ctx.set_option("pairs_wchunk", args.pairs_wchunk)
    if args.sa_builder >= 0:
        ctx.set_option("sa_builder", args.sa_builder)
    if args.host_threads:
        ctx.set_option("host_threads", args.host_threads)
    if args.filter:
        ctx.set_option("filter", args.filter)
    ctx.set_genomes_device(buf.data_ptr(), offs, lens)
    print(f"# genomes generated in {t_gen:.1f} s", file=sys.stderr, flush=True)
    if world > 1:  # every rank must hold the same genomes (same seed, same generator): compare a checksum
        chk = (buf[::4097].to(torch.int64).sum() * 1000003 + int(sum(lens))).to(coll_dev)
        lo, hi = chk.clone(), chk.clone()
        td.all_reduce(lo, op=td.ReduceOp.MIN)
        td.all_reduce(hi, op=td.ReduceOp.MAX)
        if int(lo.item()) != int(hi.item()):
            raise SystemExit("bench.py: the ranks generated different genomes")
    t_ref = time.time()
    if shared:  # (ranks sharing a GPU, tests only: their first — cold — calls of the library in turn, profiles/EXPERIMENTS.md round 6)
        for turn in range(world):
            if turn == rank:
                ctx.set_reference(ref_idx)
            td.barrier()
    else:
        ctx.set_reference(ref_idx)  # suffix array + tables: outside the metric
    t_ref = time.time() - t_ref
    ref_stats = {k: ctx.stat(k) for k in ("ms:ref_suffix_array", "ms:ref_lcp_table", "ms:ref_total", "ms:ref_fetch", "ref:sa_on_device", "ref:sa_rounds")}
    if os.environ.get("BENCH_REF_STATS"):
        print("# " + json.dumps({k: v for k, v in ctx.stats().items() if "ref" in k}), file=sys.stderr, flush=True)
    print(f"# reference index built in {t_ref:.3f} s (suffix array {(ref_stats['ms:ref_suffix_array'] or 0) / 1e3:.3f} s, "
          f"{'device, %d doubling rounds' % ref_stats['ref:sa_rounds'] if ref_stats['ref:sa_on_device'] else 'host cores'})",
          file=sys.stderr, flush=True)
    total_bases = float(sum(lens))

    emu = None
    if args.emulate_rank:
        er, en = (int(x) for x in args.emulate_rank.split("/"))
        emu = (er, en)

    emu_state = {}
    # one rank: the two N x N result matrices are the library's own page-locked home of the result (phylo_result_open,
    # private to this context): the device writes them itself, nothing is staged or widened on the host (the N-rank path
    # has the node's shared segment)
    out_mats = None
    if world == 1 and not emu:
        try:
            ctx.result_open(None, ranks=1)
            out_mats = ctx.result_matrices()
        except Exception as e:  # (no page-locked memory to be had: the caller's own matrices, staged and widened by the library)
            print(f"# result home unavailable ({e}): plain host matrices", file=sys.stderr)
    if out_mats is None:
        out_mats = (np.empty((n, n), np.uint64), np.empty((n, n), np.uint64))

    seg = {}
    def lap(name, t_prev):
        t = time.perf_counter()
        seg[name] = seg.get(name, 0.0) + (t - t_prev)
        return t

    tamper = {}
    if args.test_corrupt_rank == rank:  # tests of --verify-ranks (the record's query position is word 1 of the 16-byte record)
        def damage_block(block, maxq):
            words = block.view(torch.int32)
            words[4 + maxq + 1] += (words[0] > 0).to(torch.int32)  # (block: 4 header words, the lengths, the records)

        def damage_records(flat):
            if flat.size:
                flat = flat.copy()
                flat["index_query"][0] += 1
            return flat
        tamper = {"on_block": damage_block, "on_records": damage_records}

    def step():
        if emu:
            bounds = emu_state["bounds"]
            qb, qe = bounds[emu[0]], bounds[emu[0] + 1]
            tl = time.perf_counter()
            if args.emulate_exchange:
                # what rank emu[0] of emu[1] does (dist.process_sharded_device), with a one-rank RCCL group standing in for
                # the collectives and the other ranks' blocks prepared beforehand: phase A with its block behind it, the
                # all-gather, the attach, the rank's windows of phase B, the all-reduce, the rank's rows of the result
                W = emu[1]
                pl = emu_state["plan"]
                own = pl["all"][emu[0] * pl["nbytes"]:(emu[0] + 1) * pl["nbytes"]]
                ctx.anchor_block_device(qb, qe, own.data_ptr(), pl["maxq"], pl["cap"])
                tl = lap("anchor + export (queued)", tl)
                td.all_gather_into_tensor(own, own)
                tl = lap("all_gather (queued)", tl)
                ctx.attach_blocks_device(pl["all"].data_ptr(), bounds, pl["maxq"], pl["cap"], qb, qe)
                tl = lap("attach (queued)", tl)
                ctx.compare_triangle_device(emu[0], W, pl["tri"].data_ptr())
                tl = lap("compare (queued)", tl)
                td.all_reduce(pl["tri"])
                rep = ctx.triangle_rows_to_result(pl["tri"].data_ptr(), n * emu[0] // W, n * (emu[0] + 1) // W, emu[0], 0)
                if rep[0] or rep[1] or rep[2] or rep[4]:
                    raise SystemExit(f"emulated rank: the pass reported {rep.tolist()}")
                tl = lap("all_reduce + rows of the result + wait", tl)
                return emu_state["views"]
            ctx.anchor(qb, qe)
            tl = lap("anchor", tl)
            return ctx.compare(emu[0], emu[1])
        return dist.process_sharded(ctx, ref_idx, rank, world, device=device, lengths=lens,
                                    set_reference=False, out=out_mats if world == 1 else None, copy=False,
                                    result_rank=0 if world > 1 else None, **tamper)

    if emu:  # the other ranks' lists must exist for the projection: compute them once, untimed
        emu_state["bounds"] = [dist.query_shard(n, r, emu[1], lens)[0] for r in range(emu[1])] + [n]
        if args.emulate_exchange:
            W = emu[1]
            ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
            bounds = emu_state["bounds"]
            tot = []
            for r in range(W):
                ctx.anchor(bounds[r], bounds[r + 1])
                tot.append(int(ctx.hom_counts(bounds[r], bounds[r + 1]).sum()))
            cap = max(tot) + max(tot) // 4 + 64
            maxq = (max(bounds[r + 1] - bounds[r] for r in range(W)) + 3) // 4 * 4
            nbytes = ctx.exchange_block_bytes(maxq, cap)
            allb = torch.zeros(W * nbytes, dtype=torch.uint8, device=device)
            for r in range(W):
                ctx.anchor(bounds[r], bounds[r + 1])
                ctx.export_block_device(bounds[r], bounds[r + 1], allb.data_ptr() + r * nbytes, maxq, cap)
            torch.cuda.synchronize()
            emu_state["plan"] = {"maxq": maxq, "cap": cap, "nbytes": nbytes, "all": allb,
                                 "tri": torch.empty(ctx.triangle_words(n), dtype=torch.int32, device=device)}
            ctx.result_open(None, ranks=W)  # (one process: the result's home is private; the rank writes its rows of it)
            emu_state["views"] = ctx.result_matrices()
            print(f"# emulated exchange: blocks of {nbytes / 1e6:.2f} MB x {W} ranks, triangle {n * (n - 1) * 4 / 1e6:.2f} MB", file=sys.stderr)
        else:
            ctx.anchor(0, n)

    for _ in range(args.warmup):
        s, h = step()
    seg.clear()
    ctx.reset_stats()
    if world > 1:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s, h = step()
    torch.cuda.synchronize()
    if world > 1:
        td.barrier()
    dt = time.perf_counter() - t0
    ranks_ms, rccl_ranks = None, None
    if world > 1:
        own = torch.tensor([dt / args.steps * 1e3], dtype=torch.float64, device=coll_dev)
        every = torch.empty(world, dtype=torch.float64, device=coll_dev)
        td.all_gather_into_tensor(every, own)
        ranks_ms = [round(float(x), 3) for x in every.cpu().tolist()]  # every rank's own clock around the K steps
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
        if not shared:  # the ranks RCCL itself counts: a 1 from every rank of the communicator, summed on the devices
            ones = torch.ones(1, dtype=torch.int32, device=device)
            td.all_reduce(ones, op=td.ReduceOp.SUM)
            rccl_ranks = int(ones.item())
    # the same K steps twice more: with HIP events around every kernel (the `kernels` table, roofline_mfma, the held clock),
    # and without any (reported beside the timed figure)
    dt_plain = None
    dt_all = None
    if not args.no_profile:
        stats_timed = ctx.stats()
        ctx.set_option("profile", 1)
        ctx.reset_stats()
        seg_keep = dict(seg)
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            s, h = step()
        torch.cuda.synchronize()
        dt_all = time.perf_counter() - t1
        seg.clear()
        seg.update(seg_keep)
        # (the timed loop's own span of the chain kernel is the one the roofline uses)
        stats_all = ctx.stats()
        stats_all["ms:anchor_spec"], stats_all["n:anchor_spec"] = stats_timed.get("ms:anchor_spec", 0.0), stats_timed.get("n:anchor_spec", 0.0)
        for k2, v2 in stats_timed.items():
            if not (k2.startswith("ms:") or k2.startswith("n:") or k2.startswith("clock:")):
                stats_all[k2] = v2
        for k2 in ("ms:anchor_total", "ms:anchor_gpu", "ms:anchor_setup", "ms:compare_total", "n:anchor_calls", "n:anchor_calls_without_a_wait"):
            if k2 in stats_timed:
                stats_all[k2] = stats_timed[k2]
    if not args.no_profile and not emu:
        stats_keep = stats_all
        ctx.set_option("profile", 0)
        for _ in range(min(2, args.warmup)):
            step()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            s, h = step()
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        dt_plain = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt_plain], dtype=torch.float64, device=coll_dev)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            dt_plain = float(t.item())
        ctx.set_option("profile", 2)
    else:
        stats_keep = stats_all if not args.no_profile else None

    stats = stats_keep if stats_keep is not None else ctx.stats()
    # several ranks: the same workload's step on ONE GPU, measured in this job — rank 0 alone runs both phases over all genomes
    # (phylo_anchor_compare, what `--gpus 1 --workload <this one>` times), the other ranks idle at the barrier
    one_gpu_ms, one_gpu_same = None, None
    if world > 1 and not emu:
        if rank == 0:  # (a context of its own: the ranks' lists in `ctx` stay what the exchange left, for --verify-ranks)
            with api.Context(local) as c1:
                c1.set_genomes_device(buf.data_ptr(), offs, lens)
                c1.set_reference(ref_idx)
                c1.result_open(None, ranks=1)
                o1 = c1.result_matrices()
                k1 = max(3, min(args.steps, 10))
                for _ in range(2):
                    c1.anchor_compare(out=o1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(k1):
                    c1.anchor_compare(out=o1)
                torch.cuda.synchronize()
                one_gpu_ms = (time.perf_counter() - t1) / k1 * 1e3
                one_gpu_same = bool((np.asarray(o1[0]) == np.asarray(s)).all() and (np.asarray(o1[1]) == np.asarray(h)).all())
        td.barrier()
    if seg and rank == 0:
        print("# emulated rank, ms per step: " + "  ".join(f"{k} {v / args.steps * 1e3:.3f}" for k, v in seg.items()),
              file=sys.stderr, flush=True)
    verdict = None
    rank_report = None
    if args.verify_ranks:  # every rank's view of the timed steps, gathered on rank 0 (one JSON line on stderr, below)
        mine = {"rank": rank, "device": local, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "phases_ms": {k[3:]: round(v / args.steps, 3) for k, v in stats.items()
                              if k in ("ms:anchor_total", "ms:anchor_gpu", "ms:compare_total", "ms:triangle_zero_copy", "ms:triangle_copy", "ms:triangle_widen")},
                "kernels_ms": {k[3:]: round(v / max(1.0, stats.get("n:" + k[3:], 1.0)), 4) for k, v in stats.items()
                               if k.startswith("ms:") and ("n:" + k[3:]) in stats}}
        if world > 1:
            rank_report = [None] * world
            td.all_gather_object(rank_report, mine)
        else:
            rank_report = [mine]
    if rank == 0:
        K = args.steps
        P = n * (n - 1) // 2
        ns = 2 * lens[ref_idx] + 1
        sites = float(np.triu(h.astype(np.float64), 1).sum())
        bytes_a = total_bases + 26.0 * ns          # each query byte once + one pass over the reference-layout ESA
        bytes_b = 2.0 * sites + 16.0 * P           # two 1-byte nucleotides per compared site + one 16 B tally per pair
        shard = emu[1] if emu else world  # the fraction of the work one rank's kernels carry
        kern = {k[3:]: v for k, v in stats.items() if k.startswith("ms:") and ("n:" + k[3:]) in stats}
        launches = {k: stats["n:" + k] for k in kern}
        roof = None
        kernels = {}
        if kern:
            alg = {"anchor_spec": bytes_a / shard, "anchor_bridge": 0.0, "anchor_fold": 0.0, "anchor_compact": 0.0,
                   "pileup_project": total_bases, "pileup_project5": total_bases, "seqcmp_batch": bytes_b / shard}
            # (the pair kernels have no algorithmic bytes of their own: SURVEY 8d's bytes_B is what the byte kernels of seam B0
            # would move, the pileup's pair kernels read bit planes — their bound is roofline_mfma / roofline_valu)
            for k in kern:
                avg_ms = kern[k] / launches[k]
                kernels[k] = {"avg_ms": round(avg_ms, 4), "launches_per_step": launches[k] / K,
                              "alg_GBps": round(alg[k] / (avg_ms * 1e-3) / 1e9, 2) if alg.get(k) else None}
            # the kernel the HBM roofline is reported for: phase A's chain kernel, which carries ~15x the HBM bytes of
            # any other kernel and is the longest one at the metric's configuration; at N = 1024 the pair kernel takes
            # about as long, but it is bound by the vector ALUs (see `roofline_valu`), not by memory
            dom = "anchor_spec" if "anchor_spec" in kern else max(kern, key=lambda k: kern[k])
            avg_ms = kern[dom] / launches[dom]
            achieved = alg.get(dom, 0.0) / (avg_ms * 1e-3) / 1e9
            traffic, traffic_current = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    pj = json.load(open(pmc))
                    traffic = pj.get(args.workload, {}).get(dom)
                    # were the counters taken on the kernels this run executes?  (sha256 over the chain and phase-B kernels' sources, written by
                    # tools/tools_pmc_traffic.py when the profile was made)
                    import glob, hashlib
                    hh = hashlib.sha256()
                    cs = os.path.join(ROOT, "phylonium_amd", "csrc")
                    for f in ("lean_kernels.hip", "lean_core.h", "anchor_core.h", "pileup_kernels.hip"):  # the kernels the traffic is reported for
                        hh.update(open(os.path.join(cs, f), "rb").read())
                    traffic_current = pj.get("kernels_sha256_" + args.workload) == hh.hexdigest() if traffic else None
                except Exception:
                    traffic = None
            rq_dom = pj.get("read_requests_" + args.workload, {}).get(dom) if traffic else None
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "alg_bytes_per_launch": alg.get(dom, 0.0), "avg_launch_ms": round(avg_ms, 4),
                    # what the HBM interface actually carried (PMC bytes of an earlier profile of this workload over
                    # this run's launch time): the request-granular view of the same kernel
                    "traffic_GBps": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None,
                    "traffic_frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                    "traffic_of_these_kernels": traffic_current,
                    # the line-granularity cost as one number: what the memory carried over SURVEY 8d's algorithmic bytes, and the
                    # bytes the kernel asked for — 16 B per memory-side read request (a slot or a record; 1.11e8 on c3 = one per
                    # chain step) + the queries' 2-bit codes (|Q| / 4), each fetched as a 128-byte line
                    "wasted_traffic_ratio": round(traffic / alg[dom], 2) if traffic and alg.get(dom) else None,
                    "useful_bytes": (16.0 * rq_dom + total_bases / 4.0 / shard) if rq_dom else None,
                    "useful_frac_of_traffic": round((16.0 * rq_dom + total_bases / 4.0 / shard) / traffic, 4) if rq_dom and traffic else None,
                    "traffic_source": "profiles/pmc_traffic.json (rocprofv3 PMC passes of this workload; its source_<workload> "
                                      "entry names the profile; traffic_of_these_kernels says whether the chain and phase-B kernels' sources are the ones "
                                      "the profile was taken on (sha256 recorded by tools/tools_pmc_traffic.py); regenerate "
                                      "with tools/tools_prof.sh + tools/tools_pmc_traffic.py when the kernels change)",
                    "note": "anchor_spec fetches one random 16-byte k-mer slot per chain step, and every such fetch is a 128-byte "
                            "line at the memory whatever the load's cache policy (profiles/r04_gather_bench.jsonl: ~55 G random rows/s "
                            "= 7 TB/s of lines, the same out of the Infinity Cache): `traffic` (PMC) over the launch time — "
                            "`traffic_GBps`, `traffic_frac` — is what the HBM interface carried, `requests` the same by count; "
                            "`frac` prices SURVEY 8d's algorithmic bytes (query bytes + one pass over the reference-layout ESA)",
                    # the request-granular view: memory-side read requests per launch (PMC, the same profile) over this run's launch
                    # time, against the random-row rate the chip delivers to a pure gather
                    "requests": (lambda rq: {"per_launch": rq, "achieved_G_per_s": round(rq / (avg_ms * 1e-3) / 1e9, 2), "peak_G_per_s": 55.0,
                                             "frac": round(rq / (avg_ms * 1e-3) / 1e9 / 55.0, 4),
                                             "peak_source": "profiles/r04_gather_bench.jsonl: 16-byte rows, 128 MB - 1 GB tables, 55-57 G rows/s"}
                                 if rq else None)(rq_dom)}
        # the pair kernel against the vector ALUs: 16 x 6 instructions per window and wavefront for its 16 x 64 pairs
        # (2 xor, and, bitop3, 2 popcount-accumulate) + 8 of loop and address work, one wavefront per tile of
        # 16 x 64 genomes holding a pair i < j; peak = CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction
        roof_valu = None
        pk = next((k for k in ("pileup_pairs", "pileup_pairs_bang") if k in kern), None)
        if pk:
            ntiles = sum(1 for ig in range((n + 15) // 16) for jt in range((n + 63) // 64) if ig * 16 < jt * 64 + 63)
            windows = (lens[ref_idx] + 31) // 32 / shard
            per_win = 104 if pk == "pileup_pairs" else 135  # VALU instructions of the loop body in the ISA (hipcc -S)
            inst = ntiles * windows * per_win
            t_ms = kern[pk] / launches[pk]
            peak = 650.0  # wave64 integer instructions per ns the chip issues in register loops (tools/microbench/valu.hip: 520-650 G/s; the
            # nominal 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles = 614 is exceeded by the counted SQ_INSTS_VALU of this kernel at C4: 635 G/s)
            roof_valu = {"kernel": pk, "bound": "valu", "achieved": round(inst / (t_ms * 1e-3) / 1e9, 1), "peak": peak,
                         "unit": "G wave-instructions/s", "frac": round(inst / (t_ms * 1e-3) / 1e9 / peak, 4),
                         "wave_instructions_per_launch": inst, "avg_launch_ms": round(t_ms, 4),
                         "note": "instruction count from the kernel's shape (tiles x windows x instructions per window; the counted "
                                 "SQ_INSTS_VALU of C3 in profiles/r02_rocprof_c3_summary.json is 2 % below it); peak = one wave64 "
                                 "instruction per 4 cycles and SIMD at the nominal 2.4 GHz — plain register loops of fma, and_or or "
                                 "xor+popcount reach 520-600 G/s on this chip, so a fraction around 1 means the kernel issues as fast "
                                 "as such loops do, not that anything ran beyond the hardware"}
        # the pair kernel on the matrix cores (the default when no projected position holds '!'): four FP4 channels of
        # {-1, 0, 1} per reference position and pair, one multiply-add each — against the dense FP4 peak
        roof_mfma = None
        if "pileup_pairs_mfma" in kern:
            t_ms = kern["pileup_pairs_mfma"] / launches["pileup_pairs_mfma"]
            positions = lens[ref_idx] / shard
            flop_alg = 2.0 * 4.0 * P * positions
            nt = (n + 63) // 64
            flop_issued = 2.0 * 4.0 * (nt * (nt + 1) // 2 * 4096 - nt * 1024) * ((lens[ref_idx] + 31) // 32 * 32) / shard
            peak = 10000.0  # MI355X_MICROARCH.md: FP4 MFMA ~10 PFLOP/s dense
            roof_mfma = {"kernel": "pileup_pairs_mfma", "bound": "mfma", "achieved": round(flop_alg / (t_ms * 1e-3) / 1e12, 1), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(flop_alg / (t_ms * 1e-3) / 1e12 / peak, 4), "avg_launch_ms": round(t_ms, 4),
                         "issued_TFLOPs": round(flop_issued / (t_ms * 1e-3) / 1e12, 1),
                         # the clock the chip held under this kernel (its wavefronts' shader cycles over their 100 MHz ticks, stamped
                         # inside the kernel on profiled launches), and the peak at that clock: instruction-mix loss apart from DVFS
                         "clock_ghz": round(stats.get("clock:pairs_mfma_mhz", 0.0) / 1e3, 3) or None,
                         "frac_at_held_clock": (round(flop_alg / (t_ms * 1e-3) / 1e12 / (peak * stats["clock:pairs_mfma_mhz"] / 2400.0), 4)
                                                if stats.get("clock:pairs_mfma_mhz") else None),
                         "note": "algorithmic work: pairs x reference positions x 4 channels x 2 (v_mfma_f32_32x32x64_f8f6f4, both operands "
                                 "FP4, f32 accumulators exact below 2^24); `issued` adds the halves of the diagonal 32 x 32 sub-tiles "
                                 "and the padding of the last window.  The operands are expanded from the bit planes in registers "
                                 "(~7 vector instructions per matrix instruction): the kernel runs at the sum of the two, the chip "
                                 "holding ~1.4-2.1 GHz under it (`clock_ghz`: measured in this run; the peak is quoted at 2.4 GHz; tools/microbench/mfma_pairs.hip, DESIGN section 4)"}
        phase_b_traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(args.workload, {})
            phase_b_traffic = sum(v for k2, v in tr.items() if k2.startswith("pileup_")) or None
        except Exception:
            phase_b_traffic = None
        cpu = None
        if world == 1 and args.cpu_sample > 0:
            try:
                sa_ref = None
                S = None
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib as O
                refb = buf[offs[ref_idx]:offs[ref_idx] + lens[ref_idx]].cpu().numpy().tobytes()
                S = refb + b"#" + O.revcomp(refb)
                sa_ref = api.host_suffix_array(S)  # the suffix array is unique; saves the oracle's slow sorter
                ncpu, cpu_note = usable_cpus()
                threads = min(ncpu, 64, 1 + min(args.cpu_sample, n - 1))
                cpu = cpu_baseline(torch, buf, offs, lens, ref_idx, sa_ref, min(args.cpu_sample, n - 1), threads)
                cpu["sample"] += f"; {threads} threads ({cpu_note})"
            except Exception as e:  # the baseline is a report, never a reason to lose the bench line
                cpu = {"value": None, "unit": "Gbp/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
        if args.verify_ranks:
            verdict = verify_ranks(torch, api, dist, ctx, s, h, buf, offs, lens, ref_idx, world, local)
        roof_b0 = None
        if world == 1 and not emu and not args.no_b0:
            try:
                roof_b0 = b0_leg(ctx, s, h, lens, ref_idx)
            except Exception as e:  # a report beside the metric, never a reason to lose the bench line
                roof_b0 = {"bound": "hbm", "frac": None, "note": f"failed: {e!r}"}
        if args.check:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            m = min(n, 6)
            gs = [buf[offs[j]:offs[j] + lens[j]].cpu().numpy() for j in range(m)]
            refb = bytes(gs[0])
            sa_chk = api.host_suffix_array(refb + b"#" + O.revcomp(refb))  # unique; spares the oracle's slow sorter
            so, ho = O.Run(gs, 0).process(sa=sa_chk, threads=usable_cpus()[0]).matrix()
            ok = bool((s[:m, :m] == so).all() and (h[:m, :m] == ho).all())
            print(f"# check vs oracle on the first {m} genomes: {'OK' if ok else 'MISMATCH'}", file=sys.stderr)
            if not ok:
                raise SystemExit(2)
        out = {
            "metric": "Gbp/s through anchor+seqcmp", "value": round(total_bases * K / dt / 1e9, 4), "unit": "Gbp/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "genomes": n, "genome_length": length,
                       "query_bases": total_bases, "pairs": P, "reference": "genome 0 (unmutated base)",
                       "threshold": ctx.threshold, "seed": args.seed,
                       "parallelism": f"queries (phase A) and reference-window ranges (phase B) sharded over {world} rank(s)",
                       "backend": ("none (one rank)" if world == 1 else
                                   "gloo-staged: %d ranks share %d GPU(s); the device-resident pass with its collectives staged through the "
                                   "host (a stand-in for tests: not a bench result)" % (world, ndev) if shared else
                                   "nccl (RCCL), one rank per GPU, device-resident exchange")},
            # several ranks: what RCCL counts, every rank's own time, and the same workload on ONE GPU measured in this job
            "rccl_ranks": rccl_ranks, "ranks_ms_per_step": ranks_ms,
            "one_gpu_same_workload_ms": round(one_gpu_ms, 3) if one_gpu_ms else None,
            "speedup_same_workload": round(one_gpu_ms / (dt / K * 1e3), 3) if one_gpu_ms else None,
            "one_gpu_same_workload_identical": one_gpu_same,  # the N-rank matrices against that one-GPU run's, whole
            "ms_per_step_noprofile": round(dt_plain / K * 1e3, 3) if dt_plain else None,
            "ms_per_step_all_kernels_timed": round(dt_all / K * 1e3, 3) if dt_all else None,
            "timing_note": "value / ms_per_step: the K timed steps, HIP events around the chain kernel only (the roofline's kernel); "
                           "`kernels`, roofline_mfma and the held clock come from K more steps with events around every kernel "
                           "(ms_per_step_all_kernels_timed), ms_per_step_noprofile from K steps without any",
            "roofline": roof, "roofline_valu": roof_valu, "roofline_mfma": roof_mfma, "roofline_b0": roof_b0, "cpu_baseline": cpu,
            "phases_note": ("one rank queues both phases as one call (phylo_anchor_compare): anchor_total is the host's part of "
                            "phase A, compare_total ends with the one wait for both phases' kernels; the kernels' own times "
                            "are under `kernels`") if stats.get("n:anchor_calls_without_a_wait") else None,
            "phases_ms_per_step": {k[3:]: round(v / K, 3) for k, v in stats.items()
                                   if k in ("ms:anchor_total", "ms:anchor_gpu", "ms:anchor_setup", "ms:anchor_copyback",
                                            "ms:host_sort_filter", "ms:compare_total")},
            "extra_ms": {k[3:]: round(v / K, 3) for k, v in stats.items()
                         if k in ("ms:compare_project_phase", "ms:compare_pairs_phase", "ms:compare_symmetrise",
                                  "ms:compare_hom_flatten", "ms:compare_hom_upload", "ms:stage_send_done", "ms:stage_all")},
            "kernels": kernels,
            "compared_sites": sites, "alg_bytes": {"anchor": bytes_a, "compare": bytes_b},
            # SURVEY 8d's whole-path and phase-B-only HBM fractions are not reported (null): phase B does not move the
            # reference layout's 2 B per compared site — the pileup reads 3 bits per genome and reference position once and
            # contracts them on the matrix cores — so bytes_B over any time is not a bandwidth.  Phase B's bound is
            # roofline_mfma (matrix cores + operand expansion); what its kernels do carry at the memory is `traffic`.
            "roofline_path": {"bound": "mixed", "frac": None, "alg_bytes": bytes_a + bytes_b,
                              "note": "no single roofline: phase A is bound by the memory's line rate (roofline: frac on SURVEY 8d's bytes, "
                                      "traffic_frac on the bytes carried), phase B by the matrix cores (roofline_mfma); bytes_B of SURVEY "
                                      "8d is not moved by this design, so (bytes_A + bytes_B) / time is not a bandwidth and is not printed"},
            "roofline_phase_b": (lambda tb, tr: {"bound": "mfma", "frac": roof_mfma["frac"] if roof_mfma else None,
                                                 "frac_at_held_clock": roof_mfma.get("frac_at_held_clock") if roof_mfma else None,
                                                 "hbm_frac_on_alg_bytes": None, "alg_bytes_not_moved": bytes_b / shard,
                                                 "ms": round(tb, 3), "kernels": "pileup_project* + pileup_pairs*",
                                                 "traffic": tr,
                                                 "traffic_GBps": round(tr / (tb * 1e-3) / 1e9, 1) if tr else None,
                                                 "traffic_frac": round(tr / (tb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr else None,
                                                 "note": "frac = roofline_mfma's (the pair kernel's multiply-adds against the dense FP4 peak); "
                                                         "traffic = HBM-side bytes of these kernels from the PMC profile over this run's time"}
                                 if tb > 0 else None)(
                sum(kern[k] for k in kern if k.startswith("pileup_")) / K, phase_b_traffic),
            "phase_a_plan": {"chunk": stats.get("anchor:chunk"), "chunks": (stats.get("count:chunks") or 0) / max(1.0, stats.get("n:anchor_calls") or 1.0)},
            "setup_s": {"generate": round(t_gen, 2), "reference_index": round(t_ref, 2),
                        "suffix_array": round((ref_stats["ms:ref_suffix_array"] or 0) / 1e3, 3),
                        "suffix_array_builder": "device" if ref_stats["ref:sa_on_device"] else "host"},
        }
        if emu:  # an emulation is not a bench result: no value, and the line says what it is
            out["emulated_rank"] = "%d/%d" % emu
            out["emulated_ms_per_step"] = out["ms_per_step"]
            out["value"] = None
            out["n_gpus"] = 1
            out["note"] = ("rank %d of %d's share of every kernel, timed alone on ONE GPU%s: not a throughput figure of any job"
                           % (emu[0], emu[1], " with a one-rank RCCL group standing in for the wire" if args.emulate_exchange else ""))
        if args.dump_matrix:
            np.savez(args.dump_matrix, subst=np.asarray(s), homologs=np.asarray(h))
        if verdict is not None:
            out["verify_ranks"] = verdict
            print("# verify-ranks: " + json.dumps({"n_ranks": world, "backend": out["config"]["backend"], **verdict, "ranks": rank_report}),
                  file=sys.stderr, flush=True)
    else:
        out = None
    # BASELINE.json's second metric, wall-clock FASTA -> PHYLIP: the C++ host driver as a fresh process on the same
    # genomes (with N ranks: `phylonium-amd --gpus N`, one host thread and one context per GPU, RCCL between them)
    do_wall = not args.no_wallclock and not emu
    want_text = None
    if out is not None and do_wall:
        want_text = api.format_phylip([f"g{j:04d}" for j in range(n)], np.asarray(s), np.asarray(h))
    ctx.close()
    # one rank on the default workload: the several-GPU workload's step on THIS GPU — the baseline a 1/2/4/8 curve of
    # `--gpus N` lines (which all run that workload) is to be divided by
    if out is not None and world == 1 and not emu and default_workload and not args.no_scaling_baseline and args.workload != MULTI_GPU_WORKLOAD:
        try:
            out["scaling_baseline"] = one_gpu_leg(torch, api, MULTI_GPU_WORKLOAD, args.seed, device, local, max(3, min(args.steps, 20)))
        except Exception as e:  # a report beside the metric, never a reason to lose the bench line
            out["scaling_baseline"] = {"workload": MULTI_GPU_WORKLOAD, "ms_per_step": None, "note": f"failed: {e!r}"}
    if do_wall:
        holder = [buf]
        del buf
        sync = (lambda: td.barrier()) if world > 1 else None
        if rank == 0:
            try:
                out["wallclock"] = wallclock_leg(torch, holder, offs, lens, CONTIGS.get(args.workload, 1) > 1, want_text,
                                                 world if not shared else 1, before_run=sync)
            except Exception as e:  # a report beside the metric, never a reason to lose the bench line
                out["wallclock"] = {"wallclock_s": None, "note": f"failed: {e!r}"}
                if sync and not holder == []:
                    sync()
        else:
            holder.clear()
            torch.cuda.empty_cache()
            sync()
    if world > 1 or args.emulate_exchange:
        td.destroy_process_group()
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if out is not None:
        print(json.dumps(out), flush=True)
        if out.get("verify_ranks") and not out["verify_ranks"]["ok"]:
            raise SystemExit(3)
        if out.get("one_gpu_same_workload_identical") is False:  # (with --test-corrupt-rank: expected, and reported above)
            print("bench.py: the %d-rank result differs from one GPU's on the same genomes" % world, file=sys.stderr)
            raise SystemExit(3)


if __name__ == "__main__":
    main()

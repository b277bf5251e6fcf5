#!/usr/bin/env python3
"""Wall-clock FASTA -> PHYLIP of the host driver (phylonium_amd/phylonium-amd) on a synthetic
workload, with the split the driver's --timing prints.  Run on the GPU box:
    python tools/tools_wallclock.py [--workload c3] [--genomes N] [--out gpurun_out/wallclock.json]
Writes the genomes as FASTA files (70 columns) under /tmp, runs the driver twice (the
second run has the files in the page cache), and checks the matrix text against the
library called through the Python mirror on the same genomes."""
import argparse, json, os, re, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--genomes", type=int, default=0)
    ap.add_argument("--gpus", default="", help="also run `phylonium-amd --gpus N` (the C++ host with one thread and one context per "
                    "rank, csrc/group.hip): comma list of rank counts, e.g. 1,2,8 (more ranks than GPUs: they share them)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "wallclock.json"))
    ap.add_argument("--prepare", default="", help="internal: write the FASTA files and the expected text, then exit")
    args = ap.parse_args()
    n, length, d_range, indel, inv, desc = bench.WORKLOADS[args.workload]
    n = args.genomes or n
    d = f"/tmp/wallclock_{args.workload}_{n}"
    files = [os.path.join(d, f"g{j:04d}.fasta") for j in range(n)]
    if not args.prepare:
        # The files and the expected matrix come from a child process that is gone before the drivers run: a process
        # that still holds a GPU context makes the drivers' large hipMallocs stall, which is not what
        # someone running the driver on its own would see.
        meta_path = os.path.join(d, "meta.json")
        os.makedirs(d, exist_ok=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--genomes", str(n),
                        "--prepare", meta_path], check=True)
        meta = json.load(open(meta_path))
        want = open(os.path.join(d, "want.txt")).read()
        return drive(args, n, desc, files, want, meta)
    import torch
    from phylonium_amd import api
    dev = torch.device("cuda", 0)
    buf, offs, lens = bench.make_genomes_gpu(torch, n, length, 20260101, dev, d_range, indel, inv,
                                             contigs=bench.CONTIGS.get(args.workload, 1))
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    host = []
    for j in range(n):
        g = buf[offs[j]:offs[j] + lens[j]].cpu().numpy()
        host.append(g)
        path = files[j]
        with open(path, "wb") as f:
            for k, contig in enumerate(bytes(g).split(b"!")):
                f.write(b">contig%d\n" % k)
                a = np.frombuffer(contig, np.uint8)
                full = len(a) // 70 * 70
                lines = np.empty((full // 70, 71), np.uint8)
                lines[:, :70] = a[:full].reshape(-1, 70)
                lines[:, 70] = 10
                f.write(lines.tobytes())
                if full < len(a):
                    f.write(a[full:].tobytes() + b"\n")
    t_write = time.time() - t0
    # expected text through the library's Python mirror (reference = genome 0, as bench.py)
    with api.Context(0) as ctx:
        ctx.set_genomes(host)
        s, h = ctx.process(ref_idx=0)
    names = [f"g{j:04d}" for j in range(n)]
    open(os.path.join(d, "want.txt"), "w").write(api.format_phylip(names, s, h))
    json.dump({"bases": float(sum(lens)), "fasta_write_s": round(t_write, 2), "devices": torch.cuda.device_count()}, open(args.prepare, "w"))


# Between two runs: the driver clears the device memory a process gives back and hands none of it out before it has — a
# run started right behind another one's exit (5-50 GB each) waits for that inside its first hipMalloc, up to a second
# (tools/microbench/alloc.hip shows it without any of this code).  Somebody who runs the program once does not see it.
PAUSE_S = 2.0


def floor_runs():
    """The floor: tools/microbench/startup.hip (hipInit, a stream, hipMalloc of 1 MB, an empty kernel) start to exit on this
    box, with the runtime's orderly teardown and leaving through _exit."""
    exe = os.path.join(ROOT, "build", "startup")
    if not os.path.exists(exe):
        return None
    out = {}
    for mode in ("orderly", "quick"):
        walls, notes = [], []
        for _ in range(5):
            time.sleep(0.5)
            t0 = time.perf_counter()
            p = subprocess.run([exe] + (["quick"] if mode == "quick" else []), capture_output=True)
            walls.append(round(time.perf_counter() - t0, 4))
            notes.append(p.stderr.decode().strip()[-160:])
        out[mode] = {"wall_s": sorted(walls), "median_s": sorted(walls)[2], "inside": notes[2]}
    t0 = time.perf_counter()
    subprocess.run(["/bin/true"])
    out["spawn_of_an_empty_process_s"] = round(time.perf_counter() - t0, 4)
    return out


def drive(args, n, desc, files, want, meta):
    exe = os.path.join(ROOT, "phylonium_amd", "phylonium-amd")
    runs = []
    for label, extra in (("first run", []), ("files in page cache", []), ("files in page cache, --ingest=bytes", ["--ingest=bytes"]),
                         ("files in page cache, again", []), ("files in page cache, a third time", [])):
        time.sleep(PAUSE_S)
        t0 = time.time()
        p = subprocess.run([exe, "--timing", "-r", files[0]] + extra + files, capture_output=True)
        wall = time.time() - t0
        err = p.stderr.decode()
        m = re.search(r"timing: (.*)", err)
        runs.append({"label": label, "wall_s_including_exec": round(wall, 3), "timing": m.group(1) if m else err[-400:],
                     "matrix_identical": p.stdout.decode() == want, "exit": p.returncode})
    for ranks in [int(x) for x in args.gpus.split(",") if x]:
        for rep in range(2):
            time.sleep(PAUSE_S)
            t0 = time.time()
            p = subprocess.run([exe, "--timing", "--gpus", str(ranks), "-r", files[0]] + files, capture_output=True)
            wall = time.time() - t0
            err = p.stderr.decode()
            m = re.search(r"timing: (genomes .*)", err)
            m2 = re.search(r"timing: (\d+ ranks over .*)", err)
            runs.append({"label": f"phylonium-amd --gpus {ranks} ({meta['devices']} GPU(s) in the box), run {rep + 1}",
                         "wall_s_including_exec": round(wall, 3), "timing": m.group(1) if m else err[-400:],
                         "ranks": m2.group(1) if m2 else None, "matrix_identical": p.stdout.decode() == want, "exit": p.returncode})
    out = {"workload": f"{args.workload}: {desc}", "genomes": n, "bases": meta["bases"],
           "fasta_bytes": sum(os.path.getsize(f) for f in files), "fasta_write_s": meta["fasta_write_s"], "runs": runs,
           "floor": floor_runs(), "pause_between_runs_s": PAUSE_S,
           "note": "exit 1 is the reference's soft-warning status (io.cxx:106-139): this workload has pairs with less than "
                   "20 % homology; the matrix is printed all the same"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out, indent=1))
    for f in files:
        os.remove(f)


if __name__ == "__main__":
    main()

"""FASTA files in, PHYLIP matrix out on the N GPUs of one node — one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m phylonium_amd.mgpu [-r REF.fasta] [--raw | --ani] [--timing] FILE FILE ...

What phylonium's `main` does around process() (/root/reference/src/phylonium.cxx:89-299:
name clean-up, reference choice, one pass, print_matrix of src/io.cxx:106-163 with its
warnings and exit status), laid out for N ranks:

  * the files are split into contiguous blocks balanced by size; a rank reads only its block —
    mapped files straight to 2-bit codes (library host helper, several threads) — and uploads the
    codes into its block of the buffer; one in-place all-gather (RCCL over xGMI) makes every genome
    resident on every GPU, a quarter of the bytes of the byte form
    — the layout of each block is what `phylo_set_genomes_packed_device` wants, and the byte form
    the other kernels read is written by each GPU for itself;
  * every rank's GPU sorts the reference's suffixes for itself (12 ms at 10 M, `csrc/sa_kernels.hip`); with
    `--sa host` the rank that read the reference builds the suffix array on the host cores (north star)
    while the genomes travel, and broadcasts it;
  * `dist.process_sharded`: phase A on the rank's block of queries, homology lists
    all-gathered device to device, phase B on the rank's pair tiles, matrices all-reduced;
  * rank 0 prints.  Exit status of rank 0 is the reference's (1 after a soft warning such as
    "less than 20% homology"); the other ranks exit 0.

The single-GPU driver with the full command line (`--2pass`, `-b`, `-p`,
`--complete-deletion`) is the C++ one, `phylonium_amd/phylonium-amd`.  There is no CPU
fallback: without a GPU or the library this module fails.
"""
import argparse
import os
import sys
import threading
import time

import numpy as np


def block_layout(lens, bounds):
    """Byte offsets of every genome when block r (genomes bounds[r]:bounds[r+1]) is packed the way
    `phylo_set_genomes_device` asks — offsets multiples of 64 and >= 64, 64 zero bytes after each
    genome, 256 readable bytes at the end — and the blocks sit `cap` bytes apart.
    Returns (cap, offsets relative to the start of the gathered buffer)."""
    world = len(bounds) - 1
    rel = np.zeros(len(lens), np.int64)
    size = np.zeros(world, np.int64)
    for r in range(world):
        at = 64
        for j in range(bounds[r], bounds[r + 1]):
            rel[j] = at
            at += (int(lens[j]) + 63) // 64 * 64 + 64
        size[r] = at + 256
    cap = int((size.max() + 4095) // 4096 * 4096)
    offs = rel.copy()
    for r in range(world):
        offs[bounds[r]:bounds[r + 1]] += r * cap
    return cap, offs


def warnings_and_status(names, lens, s, h, kind, api, err=sys.stderr):
    """The checks print_matrix makes before printing (io.cxx:106-139); returns the exit status."""
    n = len(names)
    status = 0
    hf = h.astype(np.float64)
    ln = np.asarray(lens, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        raw = s.astype(np.float64) / hf
        c1 = hf / ln[:, None]
        c2 = hf / ln[None, :]
    lower = np.tril(np.ones((n, n), bool), -1)
    if kind == "jc":  # -0.75 * log(1 - 4/3 * raw): nan when the argument is negative (and when h = 0)
        with np.errstate(invalid="ignore"):
            failed = (h == 0) | (1.0 - (4.0 / 3.0) * raw < 0.0)
    else:
        failed = h == 0
    thin = ~failed & ((c1 < 0.2) | (c2 < 0.2))
    lines = []
    for i, j in zip(*np.nonzero(lower & (failed | thin))):  # row-major: i ascending, j ascending below it
        if failed[i, j]:
            lines.append(f"phylonium-amd: For the two sequences '{names[i]}' and '{names[j]}' the distance computation "
                         "failed and is reported as nan.")
        else:
            lines.append(f"phylonium-amd: For the two sequences '{names[i]}' and '{names[j]}' less than 20% homology "
                         f"were found ({c1[i, j]:f} and {c2[i, j]:f}, respectively).")
    if lines:
        print("\n".join(lines), file=err)
        status = 1
    return status


def main(argv=None):
    ap = argparse.ArgumentParser(prog="phylonium_amd.mgpu", description=__doc__.split("\n\n")[0])
    ap.add_argument("files", nargs="+")
    ap.add_argument("-r", "--reference", default="")
    ap.add_argument("--raw", action="store_true", help="uncorrected distances")
    ap.add_argument("--ani", action="store_true", help="average nucleotide identity")
    ap.add_argument("-t", "--threads", type=int, default=0, help="host threads per rank")
    ap.add_argument("--timing", action="store_true", help="where the wall-clock went (stderr, rank 0)")
    ap.add_argument("--sa", default="device", choices=["device", "host"],
                    help="who sorts the reference's suffixes: every rank's GPU for itself (default), or the host cores of the "
                         "rank that read the reference, broadcast to the others (the north star's placement)")
    ap.add_argument("--backend", default=os.environ.get("PHYLO_DIST_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="gloo: collectives through host memory (tests on a box with fewer GPUs than ranks)")
    args = ap.parse_args(argv)
    t_start = time.perf_counter()

    files = list(args.files)
    if args.reference:  # cleanup_names, phylonium.cxx:384-391
        files = sorted(set(files + [args.reference]))
    n = len(files)
    if n < 2:
        ap.error("at least two genomes are needed")
    kind = "raw" if args.raw else "ani" if args.ani else "jc"

    # The communication libraries print banners on stdout; the matrix is the only thing that may go there.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as td
    from . import api, dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("phylonium_amd.mgpu needs a GPU (no CPU fallback)")
    on_rccl = args.backend == "nccl"
    ordinal = local if on_rccl else local % torch.cuda.device_count()
    torch.cuda.set_device(ordinal)
    device = torch.device("cuda", ordinal)
    cdev = device if on_rccl else torch.device("cpu")  # where tensors handed to collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29578")
        if on_rccl:
            td.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            td.init_process_group("gloo", rank=rank, world_size=world)
    threads = args.threads or max(1, min(16, (os.cpu_count() or 1) // max(world, 1)))
    t_up = time.perf_counter()

    # the device context starts (HIP initialisation) while the files are read
    box = {}
    ctx_thread = threading.Thread(target=lambda: box.setdefault("ctx", api.Context(ordinal)))
    ctx_thread.start()

    # ── read the own block ──
    sizes = []
    for f in files:
        try:
            sizes.append(max(os.path.getsize(f), 1))
        except OSError:
            sizes.append(1)
    bounds = [dist.query_shard(n, r, world, sizes)[0] for r in range(world)] + [n]
    b0, b1 = bounds[rank], bounds[rank + 1]
    failed = n
    message = ""
    try:
        # mapped files -> 2-bit codes + separator positions (a quarter of the bytes to upload and to gather)
        mine = api.read_fasta_packed(files[b0:b1], threads) if b1 > b0 else []
    except api.PhyloniumError as e:
        mine, message = [], str(e)
        failed = b0 + next((k for k, f in enumerate(files[b0:b1]) if message.startswith(f + ":")), 0)
    t_read = time.perf_counter()
    names = [api.genome_name(f) for f in files]
    lens_t = torch.zeros(n + 1, dtype=torch.int64)
    for k, g in enumerate(mine):
        lens_t[b0 + k] = g[1]
    lens_t[n] = -failed  # the first bad file in command-line order wins, as when they are read in order
    if world > 1:
        fail_t = lens_t[n:].clone().to(cdev)
        td.all_reduce(fail_t, op=td.ReduceOp.MAX)
        first_bad = -int(fail_t.item())
    else:
        first_bad = failed
    if first_bad < n:
        if failed == first_bad:
            print(f"phylonium-amd: {message}", file=sys.stderr)
        ctx_thread.join()
        if world > 1:
            td.destroy_process_group()
        return 1
    if world > 1:
        lt = lens_t[:n].clone().to(cdev)
        td.all_reduce(lt, op=td.ReduceOp.SUM)
        lens = [int(x) for x in lt.cpu().tolist()]
    else:
        lens = [int(x) for x in lens_t[:n].tolist()]

    # ── reference (phylonium.cxx:360-382), its suffix array on the rank that holds it ──
    if args.reference:
        ref_idx = files.index(args.reference)
    else:
        ref_idx = api.host_median_length_index(lens)
    owner = next(r for r in range(world) if bounds[r] <= ref_idx < bounds[r + 1])

    def all_ok(ok, what):
        """Every rank learns whether any rank failed at `what` before the next collective, and all leave together
        (a rank that raised alone would leave the others waiting in the collective until it times out)."""
        if world > 1:
            t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=cdev)
            td.all_reduce(t, op=td.ReduceOp.SUM)
            ok = int(t.item()) == 0
        if not ok and rank == 0:
            print(f"phylonium-amd: {what}", file=sys.stderr)
        return ok

    if lens[ref_idx] == 0:  # the same lengths on every rank: all take this branch
        if rank == 0:
            print("phylonium-amd: the reference genome is empty", file=sys.stderr)
        ctx_thread.join()
        if world > 1:
            td.destroy_process_group()
        return 1
    ns = 2 * lens[ref_idx] + 1
    sa_box = {}
    sa_thread = None
    if rank == owner and args.sa == "host":
        def build_sa():
            try:
                sa_box["sa"] = api.host_reference_suffix_array(api.unpack_genome(*mine[ref_idx - b0]))
            except Exception as e:  # reported through all_ok below
                sa_box["error"] = e
        sa_thread = threading.Thread(target=build_sa)
        sa_thread.start()  # ctypes releases the GIL: this runs beside the staging and the gather

    # ── stage, upload, gather ──
    cap, offs = block_layout(lens, bounds)
    ctx_thread.join()
    if not all_ok("ctx" in box, "a rank could not create its device context"):
        if world > 1:
            td.destroy_process_group()
        return 1
    ctx = box["ctx"]
    t_ctx = time.perf_counter()
    capw = cap // 16  # block_layout's offsets are multiples of 64 bytes: words of 16 codes line up with them
    if world == 1 or on_rccl:
        # the own genomes' codes go straight from the reader's buffers into the own block of the buffer every
        # rank will hold (no staging copy on the host), and the all-gather fills the other blocks in place
        q2_all = torch.zeros(world * capw + 64, dtype=torch.int32, device=device)
        for k, (q2, ln, bad) in enumerate(mine):
            if q2.size:
                o = int(offs[b0 + k]) // 16
                q2_all[o:o + q2.size].copy_(torch.from_numpy(q2.view(np.int32)))
        if world > 1:
            td.all_gather_into_tensor(q2_all[:world * capw], q2_all[rank * capw:(rank + 1) * capw])
    else:
        stage = torch.zeros(capw, dtype=torch.int32)
        sv = stage.numpy()
        for k, (q2, ln, bad) in enumerate(mine):
            o = (int(offs[b0 + k]) - rank * cap) // 16
            sv[o:o + q2.size] = q2.view(np.int32)
        gathered = torch.empty(world * capw, dtype=torch.int32)
        td.all_gather_into_tensor(gathered, stage)
        q2_all = torch.zeros(world * capw + 64, dtype=torch.int32, device=device)
        q2_all[:world * capw].copy_(gathered)
    # the separator positions (a few per genome) travel as objects
    bad_mine = [g[2] for g in mine]
    if world > 1:
        parts = [None] * world
        td.all_gather_object(parts, bad_mine)
        bad_all = [b for part in parts for b in part]
    else:
        bad_all = bad_mine
    torch.cuda.synchronize(device)
    if args.threads:
        ctx.set_option("host_threads", args.threads)
    # a failure on one rank (device memory, a refused layout) must end all of them: the others would wait in the
    # next collective otherwise
    err = None
    try:
        ctx.set_genomes_packed_device(q2_all.data_ptr(), [int(x) for x in offs], lens, bad_all)
    except api.PhyloniumError as e:
        err = e
    if not all_ok(err is None, f"a rank could not install the genomes ({err})"):
        ctx.close()
        if world > 1:
            td.destroy_process_group()
        return 1
    if not args.reference:
        # "the first genome equal to the chosen one" (phylonium.cxx:372-378): only a same-named copy of the
        # same length can precede it; its suffix array is the same array, so the thread above is not redone
        at = lambda j: q2_all[int(offs[j]) // 16:int(offs[j]) // 16 + (lens[j] + 15) // 16]
        for i in range(ref_idx):
            if (names[i] == names[ref_idx] and lens[i] == lens[ref_idx] and np.array_equal(bad_all[i], bad_all[ref_idx])
                    and torch.equal(at(i), at(ref_idx))):
                ref_idx = i
                break
    del q2_all  # copied by the context
    t_upload = time.perf_counter()

    # ── suffix array to every rank ──
    sa = None
    if sa_thread is not None:
        sa_thread.join()
        sa = sa_box.get("sa")
    if not all_ok(args.sa != "host" or rank != owner or sa is not None, "the reference's suffix array could not be built"):
        ctx.close()
        if world > 1:
            td.destroy_process_group()
        return 1
    if world > 1 and args.sa == "host":
        sa_t = torch.from_numpy(sa.astype(np.int32)).to(cdev) if rank == owner else torch.empty(ns, dtype=torch.int32, device=cdev)
        td.broadcast(sa_t, src=owner)
        if rank != owner:
            sa = sa_t.cpu().numpy().astype(np.int64)
    t_sa = time.perf_counter()
    ctx.set_option("sa_builder", 0 if args.sa == "host" else 1)
    err = None
    try:
        ctx.set_reference(ref_idx, sa=sa)
    except api.PhyloniumError as e:
        err = e
    del sa
    if not all_ok(err is None, f"a rank could not build the reference's index ({err})"):
        ctx.close()
        if world > 1:
            td.destroy_process_group()
        return 1
    t_index = time.perf_counter()

    # ── the path ──
    s, h = dist.process_sharded(ctx, ref_idx, rank, world, device=device if on_rccl else None, lengths=lens,
                                set_reference=False)
    s, h = np.array(s, np.uint64), np.array(h, np.uint64)
    t_path = time.perf_counter()

    status = 0
    if rank == 0:
        status = warnings_and_status(names, lens, s, h, kind, api)
        sys.stdout.flush()
        text = api.format_phylip(names, s, h, kind).encode()
        while text:
            text = text[os.write(real_stdout, text):]
        t_done = time.perf_counter()
        if args.timing:
            print(f"timing: ranks {world}  genomes {n}  bases {sum(lens)}  total {t_done - t_start:.3f} s | "
                  f"imports + process group {t_up - t_start:.3f}  read own block {t_read - t_up:.3f} ({threads} threads, {b1 - b0} files)  "
                  f"lengths + wait-for-device {t_ctx - t_read:.3f}  upload + all-gather {t_upload - t_ctx:.3f}  "
                  f"wait-for-suffix-array + broadcast {t_sa - t_upload:.3f} ({'built on rank %d' % owner if args.sa == 'host' else 'none: every GPU builds its own with the index'})  "
                  f"index on device {t_index - t_sa:.3f}  anchor + exchange + compare {t_path - t_index:.3f}  "
                  f"print {t_done - t_path:.3f}", file=sys.stderr)
    ctx.close()
    if world > 1:
        td.barrier()
        td.destroy_process_group()
    return status


if __name__ == "__main__":
    sys.exit(main())

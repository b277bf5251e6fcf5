"""Multi-GPU sharding of the path (one process per GPU, torch.distributed).

The path shards without a data-path collective inside either phase
(SURVEY §8e): phase A is independent per query, phase B per pair tile.  Two
exchanges assemble the result: the filtered homology lists after phase A (every
rank needs all of them for its pair tiles) and the tally matrix after phase B.
Works on any backend: `nccl` (= RCCL over xGMI) with CUDA tensors on the GPU
box, `gloo` with CPU tensors in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as td

from .api import PACKED


# tests: run the collectives even in a one-rank group (exercises the RCCL code path on a single GPU)
_FORCE_COLLECTIVES = False
# tests: round 2's exchange (counts through the host, u64 matrices on the wire) instead of the block exchange
_LEGACY_DEVICE_EXCHANGE = False
# tests: without the node's shared home of the result (one rank fetches it, as before round 5)
_SHARED_RESULT = True


# The collectives of the device-resident pass go through this indirection: torch.distributed itself (None), or — tests on
# a box with one GPU, where RCCL refuses two ranks on one device — HostStagedCollectives below.
_COLL = None


def _coll():
    return _COLL if _COLL is not None else td


class HostStagedCollectives:
    """TEST-ONLY stand-in for RCCL when several rank PROCESSES share one GPU: the same calls on the same call sites of
    process_sharded_device, each carried out as a copy to the host ON THE CURRENT STREAM (so everything the library queued
    before it on that stream is complete, as before a real collective), the collective over gloo, and a copy back on the
    current stream (so everything queued behind it sees the result).  Never used by bench.py's RCCL runs."""

    @staticmethod
    def _staged(t, fn):
        if not t.is_cuda:
            return fn(t)
        host = t.detach().cpu()  # (waits for the current stream)
        fn(host)
        t.copy_(host)

    def all_reduce(self, t, op=td.ReduceOp.SUM):
        self._staged(t, lambda h: td.all_reduce(h, op=op))

    def reduce(self, t, dst, op=td.ReduceOp.SUM):
        self._staged(t, lambda h: td.reduce(h, dst=dst, op=op))

    def all_gather_into_tensor(self, out, inp):
        world = td.get_world_size()
        host_in = inp.detach().cpu().contiguous()
        parts = [torch.empty_like(host_in) for _ in range(world)]
        td.all_gather(parts, host_in)
        out.copy_(torch.cat([p.reshape(-1) for p in parts]).view(out.dtype).reshape(out.shape))

    def broadcast_object_list(self, objs, src=0):
        td.broadcast_object_list(objs, src=src)


def query_shard(n, rank, world, lengths=None):
    """Contiguous block of queries for this rank, balanced by total length."""
    if lengths is None:
        lengths = [1] * n
    tot = float(sum(lengths)) or 1.0
    bounds, acc, r = [0], 0.0, 1
    for j, l in enumerate(lengths):
        acc += l
        while r < world and acc >= tot * r / world:
            bounds.append(j + 1)
            r += 1
    while len(bounds) < world + 1:
        bounds.append(n)
    bounds[-1] = n
    return bounds[rank], bounds[rank + 1]


def _dev(backend_device):
    return backend_device if backend_device is not None else torch.device("cpu")


def exchange_homologies(ctx, n, rank, world, bounds, device=None, _n_pad=1, on_records=None):
    """After phase A every rank holds the lists of its own query block; afterwards
    every rank holds all of them.  Two collectives: the per-genome counts
    (all-reduce of a vector that is zero outside the own block) and the flat
    lists (all-gather of byte tensors padded to the longest block)."""
    dev = _dev(device)
    qb, qe = bounds[rank], bounds[rank + 1]
    counts, flat = ctx.export_packed(qb, qe)  # 16-byte wire records
    if on_records is not None:  # (the self-check's tests damage a record here: bench.py --test-corrupt-rank)
        flat = on_records(flat)
    call = np.zeros(n, np.int64)
    call[qb:qe] = counts.astype(np.int64)
    ct = torch.from_numpy(call).to(dev)
    td.all_reduce(ct, op=td.ReduceOp.SUM)
    call = ct.cpu().numpy()
    item = PACKED.itemsize
    sizes = [int(call[bounds[r]:bounds[r + 1]].sum()) * item for r in range(world)]
    cap = max(max(sizes), 1)
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if flat.size:
        mine[:flat.size * item] = torch.from_numpy(flat.view(np.uint8)).to(dev)
    gathered = torch.empty(world * cap * _n_pad, dtype=torch.uint8, device=dev)  # _n_pad: bench emulation of a wider gather
    td.all_gather_into_tensor(gathered[:world * cap], mine)
    g = gathered.cpu().numpy()
    for r in range(world):
        if r == rank or bounds[r] == bounds[r + 1]:
            continue
        part = g[r * cap:r * cap + sizes[r]].view(PACKED)
        ctx.import_packed(bounds[r], bounds[r + 1], call[bounds[r]:bounds[r + 1]].astype(np.uint64), part)


def allreduce_matrix(subst, homologs, device=None):
    """Sum the per-rank partial tally matrices (each pair is owned by one rank)."""
    dev = _dev(device)
    t = torch.from_numpy(np.stack([subst, homologs]).astype(np.int64)).to(dev)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    t = t.cpu().numpy().astype(np.uint64)
    return t[0], t[1]


def exchange_homologies_device(ctx, n, rank, world, bounds, device):
    """The same exchange with the records staying in device memory: every rank puts its
    own lists (16-byte records) into a device buffer, one all-gather assembles all of
    them on every GPU, and the context projects straight from that buffer."""
    qb, qe = bounds[rank], bounds[rank + 1]
    call = np.zeros(n, np.int64)
    call[qb:qe] = ctx.hom_counts(qb, qe).astype(np.int64)
    ct = torch.from_numpy(call).to(device)
    td.all_reduce(ct, op=td.ReduceOp.SUM)
    call = ct.cpu().numpy()
    sizes = [int(call[bounds[r]:bounds[r + 1]].sum()) for r in range(world)]
    cap = max(max(sizes), 1)
    item = PACKED.itemsize
    mine = torch.empty(cap * item, dtype=torch.uint8, device=device)
    gathered = torch.empty(world * cap * item, dtype=torch.uint8, device=device)
    torch.cuda.current_stream(device).synchronize()  # the library writes on its own stream
    ctx.export_packed_device(qb, qe, mine.data_ptr(), cap)
    td.all_gather_into_tensor(gathered, mine)
    torch.cuda.current_stream(device).synchronize()
    begin = np.zeros(n, np.uint64)
    for r in range(world):
        b0, b1 = bounds[r], bounds[r + 1]
        if b1 > b0:
            c = call[b0:b1]
            begin[b0:b1] = r * cap + np.concatenate(([0], np.cumsum(c[:-1])))
    ctx.attach_packed_device(gathered.data_ptr(), begin, call.astype(np.uint64), qb, qe)
    return gathered  # borrowed by the context: the caller keeps it alive until the comparison is done


def _exchange_plan(ctx, n, rank, world, bounds, device):
    """Shape of the exchange blocks (list lengths per rank, record capacity), from this step's counts: the one
    place where the ranks wait for each other's numbers on the host.  Kept on the context and reused while the
    lists fit (their sizes repeat from step to step; an overflow is detected on the device and re-plans).  With it the
    node's shared home of the result (phylo_result_open): every rank's device writes its rows of the two matrices over
    its own PCIe link; when a rank cannot map or register the segment, all fall back to one rank fetching the result."""
    qb, qe = bounds[rank], bounds[rank + 1]
    own = int(ctx.hom_counts(qb, qe).sum())
    t = torch.tensor([own], dtype=torch.int64, device=device)
    _coll().all_reduce(t, op=td.ReduceOp.MAX)
    cap = int(t.item())
    cap = cap + cap // 4 + 64
    maxq = max(bounds[r + 1] - bounds[r] for r in range(world))
    maxq = (maxq + 3) // 4 * 4 or 4
    nbytes = ctx.exchange_block_bytes(maxq, cap)
    shared = False
    seg = getattr(ctx, "_result_segment", None)
    if _SHARED_RESULT and seg == (ctx.n, world):
        shared = True  # a plan made again (lists that outgrew their blocks): the segment's size depends on n and world only,
        # it stays mapped — views handed out with copy=False stay valid
    elif _SHARED_RESULT:
        import os
        import time
        name = [None]
        ok = 1
        if rank == 0:
            name[0] = "/phylonium_amd_%d_%x" % (os.getpid(), int(time.time() * 1e6) & 0xffffffff)
            try:
                ctx.result_open(name[0], create=True, ranks=world)
            except Exception:
                ok = 0
        if world > 1:
            _coll().broadcast_object_list(name, src=0)
        if rank != 0:
            try:
                ctx.result_open(name[0], create=False, ranks=world)
            except Exception:
                ok = 0
        t = torch.tensor([ok], dtype=torch.int64, device=device)
        _coll().all_reduce(t, op=td.ReduceOp.MIN)
        if rank == 0 and ok:
            ctx.result_unlink()
        shared = bool(int(t.item()))
        if not shared:
            ctx.result_close()
        ctx._result_segment = (ctx.n, world) if shared else None
    all_blocks = torch.empty(world * nbytes, dtype=torch.uint8, device=device)
    return {"maxq": maxq, "cap": cap, "bounds": tuple(bounds), "nbytes": nbytes, "shared_result": shared,
            "views": ctx.result_matrices() if shared else None,
            # a rank's own block lies in its place of the gathered buffer: the all-gather fills the rest in place
            "block": all_blocks[rank * nbytes:(rank + 1) * nbytes], "all": all_blocks,
            "tri": torch.empty(ctx.triangle_words(n), dtype=torch.int32, device=device)}


class _Repeat(Exception):
    """A pass that every rank repeats (all of them read the same summed report): `how` says in which way."""

    def __init__(self, how, msg):
        super().__init__(msg)
        self.how = how


def _check_report(rep):
    if int(rep[2]):
        raise _Repeat("plan", "the lists gathered from the ranks overflowed their blocks' capacity")
    if int(rep[4]):
        raise _Repeat("anchor", "a rank's phase A needs the host (a list with tied starts, or scratch that overflowed)")
    if int(rep[0]):
        raise _Repeat("pairs", "more '!' inside homologies than the genomes hold separators: the vector-ALU pair kernels take the pass")
    if int(rep[1]):
        raise RuntimeError("a gathered list is not sorted by projected start, disjoint and inside the reference")


def process_sharded_device(ctx, rank, world, bounds, device, out=None, result_rank=None, on_block=None, copy=True):
    """One step with the exchange left to the device: the context works on torch's current stream, so the library's
    kernels and the collectives are ordered by the stream and the host waits ONCE, for the result.  Phase A of the rank's
    queries is queued with its exchange block behind it (phylo_anchor_block_device); lists travel as fixed-shape blocks
    (one all-gather); tallies as a u32 upper triangle (one all-reduce, a quarter of the bytes of the two u64 matrices);
    every rank's device writes its rows of the result into the node's shared page-locked segment
    (phylo_triangle_rows_to_result).  What a rank would have learnt at a wait of its own — a list that needs the host's
    std::sort, an overflowing block, a '!' list beyond its capacity — comes back in the summed triangle's report, which
    every rank reads: the pass is then repeated, by all of them alike, the long way."""
    n = ctx.n
    qb, qe = bounds[rank], bounds[rank + 1]
    stream = torch.cuda.current_stream(device).cuda_stream
    if getattr(ctx, "_on_stream", None) != stream:
        ctx.set_stream(stream)
        ctx._on_stream = stream
    # the route this data set needs is kept with the context (api.Context._new_inputs clears it): a set with a tied-start
    # list, or more '!' than the lists hold, raises its report on every step — later steps start on the route that worked
    route = getattr(ctx, "_route", None) or {}
    slow_anchor, valu_pairs = bool(route.get("slow_anchor")), bool(route.get("valu_pairs"))
    for attempt in range(4):
        plan = getattr(ctx, "_xplan", None)
        usable = plan is not None and plan["bounds"] == tuple(bounds) and plan["tri"].numel() == ctx.triangle_words(n)
        try:
            if valu_pairs:
                ctx.set_option("pairs_kernel", 1)
            # everything from here to the result is queued on the stream: phase A with the block export behind it, the
            # all-gather, the attach, the comparison (its kernels and what it has to report ride in the triangle's last
            # words), the all-reduce and the rows of the result
            if usable and not slow_anchor:
                ctx.anchor_block_device(qb, qe, plan["block"].data_ptr(), plan["maxq"], plan["cap"])
            else:  # no plan yet (the blocks are sized from this pass's own lists), or a pass repeated the long way
                ctx.anchor(qb, qe)
                if not usable:
                    plan = ctx._xplan = _exchange_plan(ctx, n, rank, world, bounds, device)
                ctx.export_block_device(qb, qe, plan["block"].data_ptr(), plan["maxq"], plan["cap"])
            if on_block is not None:  # (the self-check's tests damage a record here: bench.py --test-corrupt-rank)
                on_block(plan["block"], plan["maxq"])
            _coll().all_gather_into_tensor(plan["all"], plan["block"])
            ctx.attach_blocks_device(plan["all"].data_ptr(), bounds, plan["maxq"], plan["cap"], qb, qe)
            ctx._attached_records = plan["all"]  # still the source of the other ranks' lists should the caller ask for them
            # (the vector-ALU pair kernel, option pairs_kernel = 1, waits for its flags itself: a block that overflowed
            # surfaces here, on every rank alike — all of them hold all blocks — and is planned again like the others)
            ctx.compare_triangle_device(rank, world, plan["tri"].data_ptr())
            wants = result_rank is None or rank == result_rank
            if plan["shared_result"]:
                _coll().all_reduce(plan["tri"], op=td.ReduceOp.SUM)
                rep = ctx.triangle_rows_to_result(plan["tri"].data_ptr(), n * rank // world, n * (rank + 1) // world, rank,
                                                  world if wants else 0)
                _check_report(rep)
                if not wants:
                    return None, None
                s, h = plan["views"]
                if out is not None:
                    np.copyto(out[0], s)
                    np.copyto(out[1], h)
                    return out[0], out[1]
                return (s.copy(), h.copy()) if copy else (s, h)
            if result_rank is None:  # every rank gets the matrices
                _coll().all_reduce(plan["tri"], op=td.ReduceOp.SUM)
                return ctx.triangle_to_matrices(plan["tri"].data_ptr(), out)
            # the job's one result, on one rank (as the reference prints one matrix): a reduce instead of the all-reduce, and
            # the other ranks neither copy 2 N^2 words home nor widen them on host cores the result's rank could use; the
            # parts' reports are all-reduced beside it (32 bytes), so that every rank learns of a pass to repeat
            report = plan["tri"][-8:].clone()
            _coll().reduce(plan["tri"], dst=result_rank, op=td.ReduceOp.SUM)
            _coll().all_reduce(report, op=td.ReduceOp.SUM)
            _check_report(report.cpu().numpy())
            if rank == result_rank:
                return ctx.triangle_to_matrices(plan["tri"].data_ptr(), out)
            return None, None
        except _Repeat as e:
            how = e.how
        except Exception as e:  # the library's own words for the same verdicts (the paths that wait for their flags themselves)
            msg = str(e)
            how = "plan" if "overflow" in msg and "scratch" not in msg else "anchor" if "needs the host" in msg else \
                "pairs" if "pairs_kernel = 1" in msg else None
            if how is None:
                if plan is not None and plan.get("shared_result"):
                    ctx.result_abandon(rank)  # (the ranks that wait for this one's rows return at once, not after their time-out)
                    ctx._result_segment = None
                raise
        finally:
            if valu_pairs:
                ctx.set_option("pairs_kernel", 0)
        if how == "plan":
            if getattr(ctx, "_xplan", None) is None:
                raise RuntimeError("process_sharded_device: the exchange blocks overflowed twice")
            ctx._xplan = None
        elif how == "anchor":
            if slow_anchor:
                raise RuntimeError("process_sharded_device: phase A failed on the host's route as well")
            slow_anchor = True
            ctx._route = dict(route, slow_anchor=True, valu_pairs=valu_pairs)
        else:
            if valu_pairs:
                raise RuntimeError("process_sharded_device: the vector-ALU pair kernels reported a '!' list overflow")
            valu_pairs = True
            ctx._route = dict(route, slow_anchor=slow_anchor, valu_pairs=True)
    raise RuntimeError("process_sharded_device: the pass was repeated three times without a result")


def shard_bounds(ctx, world, lengths=None):
    """bounds[r] .. bounds[r + 1]: rank r's block of queries (kept on the context: a thousand genomes' lengths are not
    walked through again step after step)."""
    lens = lengths or getattr(ctx, "lengths", None)
    key = (ctx.n, world, id(lens), len(lens) if lens is not None else 0)
    cached = getattr(ctx, "_shard_bounds", None)
    if cached is None or cached[0] != key:
        cached = ctx._shard_bounds = (key, [query_shard(ctx.n, r, world, lens)[0] for r in range(world)] + [ctx.n])
    return cached[1]


def process_sharded(ctx, ref_idx, rank, world, device=None, lengths=None, set_reference=True, out=None, copy=True, result_rank=None,
                    on_block=None, on_records=None):
    """process() with queries and pair tiles sharded over `world` ranks.
    ctx: an api.Context (or any object with the same methods) holding all genomes.
    Returns (subst, homologs), two N x N uint64 arrays.  With `out` = (subst, homologs) the result is written
    there and those arrays are returned; without it the arrays are the caller's own (fresh copies) on every
    path — the device-resident path keeps a pinned staging buffer of its own that the next call overwrites;
    copy=False hands out views of that buffer instead (valid until the next call: a timing loop that looks at
    the last result only).  result_rank = r (device-side exchange only): the tallies are reduced to rank r alone, which
    returns the matrices; the other ranks return (None, None)."""
    if set_reference:
        ctx.set_reference(ref_idx)
    bounds = shard_bounds(ctx, world, lengths)
    qb, qe = bounds[rank], bounds[rank + 1]
    if world == 1 and not _FORCE_COLLECTIVES and hasattr(ctx, "anchor_compare"):
        return ctx.anchor_compare(out=out)  # one rank: both phases as the one call they are in the reference
    on_gpu = device is not None and torch.device(device).type == "cuda" and hasattr(ctx, "attach_packed_device")
    if on_gpu and (world > 1 or _FORCE_COLLECTIVES) and hasattr(ctx, "export_block_device") and not _LEGACY_DEVICE_EXCHANGE:
        # (phase A is this path's own first step — on the caller's stream, and again should the exchange blocks overflow)
        return process_sharded_device(ctx, rank, world, bounds, device, out=out, result_rank=result_rank, on_block=on_block, copy=copy)
    ctx.anchor(qb, qe)
    if on_gpu and (world > 1 or _FORCE_COLLECTIVES):
        # device-resident: records and tallies never visit the host between the ranks
        keep = exchange_homologies_device(ctx, ctx.n, rank, world, bounds, device)
        n = ctx.n
        t = torch.empty(2 * n * n, dtype=torch.int64, device=device)
        torch.cuda.current_stream(device).synchronize()
        ctx.compare_device(rank, world, t.data_ptr(), t.data_ptr() + n * n * 8)
        ctx._attached_records = keep  # still the source of the other ranks' lists should the caller ask for them
        td.all_reduce(t, op=td.ReduceOp.SUM)
        pin = getattr(ctx, "_pinned_matrix", None)  # a pageable D2H of 2 x N x N x 8 B costs milliseconds at N = 1024
        if pin is None or pin.numel() != t.numel():
            pin = torch.empty(t.numel(), dtype=torch.int64, pin_memory=True)
            ctx._pinned_matrix = pin
        pin.copy_(t, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
        m = pin.numpy().view(np.uint64).reshape(2, n, n)  # the pinned staging buffer: overwritten by the next call
        if out is not None:
            np.copyto(out[0], m[0])
            np.copyto(out[1], m[1])
            return out[0], out[1]
        return (m[0].copy(), m[1].copy()) if copy else (m[0], m[1])
    if world > 1 or _FORCE_COLLECTIVES:
        exchange_homologies(ctx, ctx.n, rank, world, bounds, device, on_records=on_records)
    s, h = ctx.compare(rank, world, out=out) if out is not None else ctx.compare(rank, world)
    if world > 1 or _FORCE_COLLECTIVES:
        s, h = allreduce_matrix(s, h, device)
    return s, h

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

from .api import PHOM


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


def allgather_homologies(local, n, device=None):
    """local: {genome index: PHOM array} for this rank's queries → list of n arrays."""
    world = td.get_world_size()
    dev = _dev(device)
    idx = sorted(local)
    counts = np.zeros(n, np.int64)
    for j in idx:
        counts[j] = len(local[j])
    ct = torch.from_numpy(counts).to(dev)
    td.all_reduce(ct, op=td.ReduceOp.SUM)
    counts_all = ct.cpu().numpy()
    flat = np.concatenate([np.ascontiguousarray(local[j], PHOM) for j in idx]) if idx else np.zeros(0, PHOM)
    mine = torch.from_numpy(flat.view(np.uint8).copy()).to(dev)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    sizes[td.get_rank()] = mine.numel()
    td.all_reduce(sizes, op=td.ReduceOp.SUM)
    sizes = sizes.cpu().numpy()
    cap = int(sizes.max())
    pad = torch.zeros(max(cap, 1), dtype=torch.uint8, device=dev)
    pad[:mine.numel()] = mine
    parts = [torch.zeros_like(pad) for _ in range(world)]
    td.all_gather(parts, pad)
    owner = torch.full((n,), -1, dtype=torch.int64, device=dev)
    for j in idx:
        owner[j] = td.get_rank()
    td.all_reduce(owner, op=td.ReduceOp.MAX)
    owner = owner.cpu().numpy()
    out = [np.zeros(0, PHOM)] * n
    cursor = [0] * world
    for j in range(n):
        r = int(owner[j])
        if r < 0:
            continue
        nb = int(counts_all[j]) * PHOM.itemsize
        buf = parts[r][cursor[r]:cursor[r] + nb].cpu().numpy()
        cursor[r] += nb
        out[j] = buf.view(PHOM).copy()
    return out


def allreduce_matrix(subst, homologs, device=None):
    """Sum the per-rank partial tally matrices (each pair is owned by one rank)."""
    dev = _dev(device)
    t = torch.from_numpy(np.stack([subst, homologs]).astype(np.int64)).to(dev)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    t = t.cpu().numpy().astype(np.uint64)
    return t[0], t[1]


def process_sharded(ctx, ref_idx, rank, world, device=None, lengths=None, set_reference=True):
    """process() with queries and pair tiles sharded over `world` ranks.
    ctx: an api.Context (or any object with the same methods) holding all genomes."""
    if set_reference:
        ctx.set_reference(ref_idx)
    qb, qe = query_shard(ctx.n, rank, world, lengths or getattr(ctx, "lengths", None))
    ctx.anchor(qb, qe)
    if world > 1:
        local = {j: ctx.homologies(j) for j in range(qb, qe)}
        allh = allgather_homologies(local, ctx.n, device)
        for j in range(ctx.n):
            if not (qb <= j < qe):
                ctx.set_homologies(j, allh[j])
    s, h = ctx.compare(rank, world)
    if world > 1:
        s, h = allreduce_matrix(s, h, device)
    return s, h

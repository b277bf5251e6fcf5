"""world_size-2 gloo test of the multi-GPU plumbing (phylonium_amd/dist.py):
query sharding, homology all-gather, pair-tile sharding and matrix all-reduce.
No GPU here, so the per-rank compute is stood in by the oracle (test-only);
what is under test is that the sharded assembly reproduces the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


class OracleCtx:
    """Same methods as api.Context, computing with the CPU oracle (test stand-in)."""

    def __init__(self, genomes):
        import oracle_lib as O
        self.O = O
        self.genomes = genomes
        self.n = len(genomes)
        self.lengths = [len(g) for g in genomes]
        self.h = [None] * self.n

    def set_reference(self, ref_idx):
        self.ref_idx = ref_idx
        self.esa = self.O.Esa(self.genomes[ref_idx])
        self.thr = self.O.min_anchor_length(0.025, self.O.gc_content(bytes(self.genomes[ref_idx])), self.esa.size)

    def anchor(self, qb, qe):
        from phylonium_amd.api import PHOM
        L = self.lengths[self.ref_idx]
        for j in range(qb, qe):
            raw = self.esa.anchor(self.thr, self.genomes[j])
            for r in raw:  # reverseEh
                if r["iref"] >= L:
                    r["iproj"] = 2 * L + 1 - r["len"] - r["iref"]
                    r["rev"] = 1
            f = self.O.sort_filter(raw)
            out = np.zeros(len(f), PHOM)
            out["index_reference"], out["index_reference_projected"] = f["iref"], f["iproj"]
            out["index_query"], out["length"], out["direction"] = f["iq"], f["len"], f["rev"]
            self.h[j] = out

    def homologies(self, j):
        return self.h[j]

    def set_homologies(self, j, h):
        self.h[j] = np.array(h)

    def export_packed(self, qb, qe):
        from phylonium_amd.api import PACKED, PHOM
        counts = np.array([len(self.h[j]) for j in range(qb, qe)], np.uint64)
        full = np.concatenate([self.h[j] for j in range(qb, qe)]) if qe > qb else np.zeros(0, PHOM)
        flat = np.zeros(len(full), PACKED)
        flat["start"], flat["index_query"] = full["index_reference_projected"], full["index_query"]
        flat["length"], flat["direction"] = full["length"], full["direction"]
        return counts, flat

    def import_packed(self, qb, qe, counts, flat):
        from phylonium_amd.api import PHOM
        L = self.lengths[self.ref_idx]
        o = 0
        for k, j in enumerate(range(qb, qe)):
            c = int(counts[k])
            p = flat[o:o + c]
            h = np.zeros(c, PHOM)
            h["index_reference_projected"], h["index_query"] = p["start"], p["index_query"]
            h["length"], h["direction"] = p["length"], p["direction"]
            h["index_reference"] = np.where(p["direction"] == 1, 2 * L + 1 - p["length"].astype(np.int64) - p["start"], p["start"])
            self.h[j] = h
            o += c

    def compare(self, part, nparts):
        O = self.O
        s = np.zeros((self.n, self.n), np.uint64)
        h = np.zeros((self.n, self.n), np.uint64)
        pid = 0
        for i in range(self.n):
            for j in range(i + 1, self.n):
                if pid % nparts == part:
                    a, b = self.h[i], self.h[j]
                    def conv(x):
                        o = np.zeros(len(x), O.HOM_DTYPE)
                        o["rev"], o["iref"], o["iproj"] = x["direction"], x["index_reference"], x["index_reference_projected"]
                        o["iq"], o["len"] = x["index_query"], x["length"]
                        return o
                    ss, hh = O.compare_lists(self.genomes[i], conv(a), self.genomes[j], conv(b))
                    s[i, j] = s[j, i] = ss
                    h[i, j] = h[j, i] = hh
                pid += 1
        return s, h


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from phylonium_amd import dist, synth
    gs = synth.make_genomes(7, 12000, seed=17, d_range=(0.01, 0.2), indel_per_mbp=400, inv_frac=0.08)
    ctx = OracleCtx(gs)
    s, h = dist.process_sharded(ctx, 2, rank, world, device=None)
    if rank == 0:
        np.save(out + ".s.npy", s)
        np.save(out + ".h.npy", h)
    td.destroy_process_group()


def test_query_shard_partitions():
    from phylonium_amd import dist
    for n, world in ((1, 2), (7, 2), (256, 8), (5, 8)):
        lens = list(np.random.default_rng(n).integers(1, 100, n))
        seen = []
        for r in range(world):
            b, e = dist.query_shard(n, r, world, lens)
            assert 0 <= b <= e <= n
            seen += list(range(b, e))
        assert seen == list(range(n))


def test_two_rank_sharded_process_matches_unsharded(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    import oracle_lib as O
    from phylonium_amd import synth
    gs = synth.make_genomes(7, 12000, seed=17, d_range=(0.01, 0.2), indel_per_mbp=400, inv_frac=0.08)
    so, ho = O.Run(gs, 2).process().matrix()
    assert (np.load(out + ".s.npy") == so).all()
    assert (np.load(out + ".h.npy") == ho).all()


def _staged_worker(rank, world, port, out):
    import torch
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from phylonium_amd import dist
    c = dist.HostStagedCollectives()
    # the in-place all-gather of process_sharded_device: a rank's own block is a view of the gathered buffer
    allb = torch.zeros(world * 5, dtype=torch.uint8)
    own = allb[rank * 5:(rank + 1) * 5]
    own[:] = torch.arange(5, dtype=torch.uint8) + 10 * rank
    c.all_gather_into_tensor(allb, own)
    assert allb.tolist() == [v + 10 * r for r in range(world) for v in range(5)]
    t = torch.tensor([rank + 1, 7], dtype=torch.int32)
    c.all_reduce(t, op=td.ReduceOp.SUM)
    assert t.tolist() == [world * (world + 1) // 2, 7 * world]
    m = torch.tensor([rank], dtype=torch.int64)
    c.all_reduce(m, op=td.ReduceOp.MAX)
    assert int(m) == world - 1
    r = torch.tensor([1], dtype=torch.int32)
    c.reduce(r, dst=0, op=td.ReduceOp.SUM)
    assert rank != 0 or int(r) == world
    name = ["x%d" % rank]
    c.broadcast_object_list(name, src=0)
    assert name == ["x0"]
    if rank == 0:
        open(out, "w").write("ok")
    td.destroy_process_group()


def test_staged_collectives_keep_torch_distributeds_semantics(tmp_path):
    """dist.HostStagedCollectives — the stand-in that lets several rank processes share one GPU in the -m gpu tests of
    dist.process_sharded_device — behaves like the torch.distributed calls it replaces (CPU tensors here: the staging copy is
    the identity): the in-place all-gather of a view of its own output, all-reduce (sum, max), reduce, object broadcast."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "ok")
    mp.spawn(_staged_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"

import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from phylonium_amd import api, synth
gs = synth.make_genomes(13, 25000, seed=86, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.08, contigs=2)
dup = np.concatenate([gs[5][1000:9000], synth.random_base(300, np.random.default_rng(1)), gs[5][1000:9000]])
g2 = gs + [dup]
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    c = api.Context(0)
    c.set_stream(side.cuda_stream)
    c.set_genomes(g2)
    c.set_reference(5)
    c.anchor(9, 14)
    print("stats after anchor(9,14):", {k: v for k, v in c.stats().items() if k.startswith("count:")})
    maxq, cap = 8, 100000
    nbytes = c.exchange_block_bytes(maxq, cap)
    blk = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    c.reset_stats()
    c.anchor_block_device(9, 14, blk.data_ptr(), maxq, cap)
    side.synchronize()
    w = blk.cpu().numpy().view(np.uint32)
    print("block header", w[:4], "lengths", w[4:12])
    h = c.homologies(13)
    print("list 13", h)
    print("stats:", {k: v for k, v in c.stats().items() if k.startswith("count:") or k.startswith("n:")})

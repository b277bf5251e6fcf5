import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from phylonium_amd import api, synth
gs = synth.make_genomes(13, 25000, seed=86, d_range=(0.01, 0.25), indel_per_mbp=300, inv_frac=0.08, contigs=2)
with api.Context(0) as one:
    one.set_genomes(gs)
    print("=== process", file=sys.stderr, flush=True)
    so, ho = one.process(5)
print("done", int(ho.sum()))

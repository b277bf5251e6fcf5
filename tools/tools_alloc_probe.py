#!/usr/bin/env python3
"""dev: one cold one-context process() of thirteen 25 kbp genomes — with the development build and PHY_DEBUG_ALLOC=1 every
device / page-locked allocation of that call on stderr (addresses, sizes); PHY_DEBUG_POISON=0xff fills new buffers:
    PHYLONIUM_AMD_LIB=phylonium_amd/libphylonium_amd_dev.so PHY_DEBUG_ALLOC=1 python tools/tools_alloc_probe.py"""
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

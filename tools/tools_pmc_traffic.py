#!/usr/bin/env python3
"""usage: tools_pmc_traffic.py <summary.json from tools_prof.sh> <workload> <source label>
Updates profiles/pmc_traffic.json: HBM-side bytes per launch for every kernel, from the
rocprofv3 PMC passes (reads = TCC_EA0_RDREQ_128B*128 + _64B*64 + _32B*32; FETCH_SIZE
tallies every request at 64 B, so it shows half the bytes of 128-B requests; writes =
WRITE_SIZE KB)."""
import json, os, sys

summary, wl, label = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles", "pmc_traffic.json")
cur = json.load(open(out)) if os.path.exists(out) else {}
pmc = json.load(open(summary))["pmc_per_dispatch_avg"]
tot, det = {}, {}
for k, v in pmc.items():
    rd = v.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + v.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + v.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
    wr = v.get("WRITE_SIZE", 0) * 1024
    tot[k] = rd + wr
    det[k] = {"read_bytes": rd, "write_bytes": wr, "bytes": rd + wr, "fetch_size_kb": v.get("FETCH_SIZE")}
cur[wl] = tot
cur["detail_" + wl] = det
cur["read_requests_" + wl] = {k: v.get("TCC_EA0_RDREQ_sum", 0) for k, v in pmc.items()}  # memory-side read requests per launch
cur["method"] = ("HBM-side bytes per launch from rocprofv3 PMC (separate passes): reads = TCC_EA0_RDREQ_128B*128 + _64B*64 "
                 "+ _32B*32 (FETCH_SIZE tallies every request at 64 B, i.e. half the bytes of the 128-B requests, as "
                 "MI355X_MICROARCH.md warns); writes = WRITE_SIZE KB. Sources: the source_<workload> entries")
cur["source_" + wl] = label
# the kernels' source at the time of the profile: bench.py compares it with what it runs and says so when they differ
import hashlib, glob
hh = hashlib.sha256()
# (the kernels whose traffic bench.py reports: the chain kernels and phase B's)
for f in ("lean_kernels.hip", "lean_core.h", "anchor_core.h", "pileup_kernels.hip"):
    hh.update(open(os.path.join(root, "phylonium_amd", "csrc", f), "rb").read())
cur["kernels_sha256_" + wl] = hh.hexdigest()
json.dump(cur, open(out, "w"), indent=1)
print(json.dumps(tot, indent=1))

#!/usr/bin/env python3
"""What a rank's step cannot shed: the kernels behind the chains — bridges, fold, sort + filter — are latency chains per
wavefront / per query, so their time barely depends on how many queries they serve.  Phase A of rank R of 8's block of c4
(bench.py's genomes), then of its first 64, 16, 4 and 1 queries: every kernel's HIP-event time per call.  If one query's
bridges + fold + filter take about what 128 queries' do, starting a query's bridges and fold as soon as ITS chunks are
done cannot take the step below  chains + that one query's tail.
    gpurun -- 'python tools/tools_floor.py > gpurun_out/floor.json'"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from phylonium_amd import api, dist
    wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
    rank, world = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3/8").split("/"))
    n, length, d_range, indel, inv, desc = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    buf, offs, lens = bench.make_genomes_gpu(torch, n, length, 20260101, dev, d_range, indel, inv)
    torch.cuda.synchronize()
    out = {"workload": wl, "rank": f"{rank}/{world}", "rows": []}
    with api.Context(0) as c:
        c.set_genomes_device(buf.data_ptr(), offs, lens)
        c.set_reference(0)
        c.set_option("profile", 1)
        bounds = [dist.query_shard(n, r, world, lens)[0] for r in range(world)] + [n]
        qb, qe = bounds[rank], bounds[rank + 1]
        keys = ("anchor_spec", "anchor_overruns", "anchor_bridge", "anchor_fold", "anchor_filter")
        for m in (qe - qb, 64, 16, 4, 1):
            m = min(m, qe - qb)
            for rep in range(3):
                c.anchor(qb, qb + m)
            c.reset_stats()
            reps = 10
            for rep in range(reps):
                c.anchor(qb, qb + m)
            st = c.stats()
            row = {"queries": m, "chunk": st.get("anchor:chunk"), "chunks": (st.get("count:chunks") or 0) / reps}
            for k in keys:
                if ("n:" + k) in st and st["n:" + k]:
                    row[k + "_ms"] = round(st["ms:" + k] / reps, 4)
            row["tail_ms"] = round(sum(row.get(k + "_ms", 0.0) for k in ("anchor_bridge", "anchor_fold", "anchor_filter")), 4)
            out["rows"].append(row)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

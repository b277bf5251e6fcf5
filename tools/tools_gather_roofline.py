#!/usr/bin/env python3
"""Practical ceiling for random 128-byte line gathers out of a table far larger
than the Infinity Cache (what the anchor kernel's slot lookups are): torch
index_select of 128-byte rows from a 2 GiB table with uniformly random indices.
Prints GB/s and G lines/s. Run on the GPU box."""
import json, time, torch
dev = torch.device("cuda", 0)
out = {}
for rows_bytes in (128, 64):
    cols = rows_bytes // 4
    n = (2 << 30) // rows_bytes
    table = torch.empty((n, cols), dtype=torch.float32, device=dev).normal_()
    for m in (1 << 24, 1 << 26):
        idx = torch.randint(0, n, (m,), device=dev)
        for _ in range(2):
            r = table.index_select(0, idx)
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 5
        for _ in range(reps):
            r = table.index_select(0, idx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        out[f"row{rows_bytes}B_m{m}"] = {"ms": round(dt * 1e3, 3), "Glines_per_s": round(m / dt / 1e9, 2),
                                          "read_GBps": round(m * rows_bytes / dt / 1e9, 1)}
        del r, idx
    del table
print(json.dumps(out))

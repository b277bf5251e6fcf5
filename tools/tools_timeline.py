#!/usr/bin/env python3
"""usage: tools_timeline.py <kernel_trace.csv | rocprofv3 output directory> — dev: the last process() of a traced bench run
as a timeline: every kernel's (and, when the directory holds a memory-copy trace, every copy's) start (us from the step's
first kernel), duration and the idle gap before it."""
import csv, glob, os, sys


def load(path):
    if os.path.isdir(path):
        ks = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        ms = glob.glob(os.path.join(path, "**", "*memory_copy_trace.csv"), recursive=True)
    else:
        ks, ms = [path], []
    rows = []
    for f in ks:
        for r in csv.DictReader(open(f)):
            if "Start_Timestamp" in r:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], "k"))
    for f in ms:
        for r in csv.DictReader(open(f)):
            if "Start_Timestamp" in r:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Kind", "")), "c"))
    rows.sort()
    return rows


def short(n):
    n = n.split("(")[0]
    return n.replace("phy::", "").replace("void ", "")[:52]


rows = load(sys.argv[1])
# a step starts at the chain kernel <0>
starts = [i for i, r in enumerate(rows) if "lean_chain_kernel<0" in r[2]]
if len(starts) < 2:
    sys.exit("no steps found")
a, b = starts[-2], starts[-1]
t0 = rows[a][0]
prev_end = None
tot_k = busy = 0
for i in range(a, b):
    s, e, name, kind = rows[i]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:6.1f}  {short(name)}")
    busy += max(0, e - max(s, prev_end or 0))
    prev_end = max(prev_end or 0, e)
    if kind == "k":
        tot_k += e - s
span = (rows[b][0] - t0) / 1e3
print(f"step {span:.1f} us from chain kernel to chain kernel; kernels {tot_k / 1e3:.1f} us; device busy (kernels and copies, overlaps "
      f"counted once) {busy / 1e3:.1f} us; idle {span - busy / 1e3:.1f} us")

#!/usr/bin/env python3
"""usage: tools_timeline.py <kernel_trace.csv> — dev: the last process() of a traced bench run as a timeline:
every kernel's start (us from the step's first kernel), duration and the idle gap before it."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "Start_Timestamp" in r]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0]
    return n.replace("phy::", "").replace("void ", "")[:44]
# a step starts at the chain kernel <0>
starts = [i for i, r in enumerate(rows) if "lean_chain_kernel<0" in r["Kernel_Name"]]
if len(starts) < 2:
    sys.exit("no steps found")
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
# include what precedes the chain kernel of this step (set-up kernels, memsets) back to the end of the previous step's last kernel
prev_end = None
tot_k = 0
for i in range(a, b):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:6.1f}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end or 0, e)
    tot_k += e - s
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
print(f"step {span:.1f} us from chain kernel to chain kernel; kernels {tot_k / 1e3:.1f} us; idle {span - tot_k / 1e3:.1f} us")

#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + PMC passes) for the library's
own kernels (names containing 'phy::'), write small JSON/CSV files, drop the raw
CSVs.  usage: tools_prof_summary.py <prof_dir>"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

d = sys.argv[1]
out = {}

NAMES = [  # substring of the kernel's name -> the name bench.py and the library's stats use; first match wins
    ("chain_kernel<0>", "anchor_spec"), ("chain_kernel<1>", "anchor_bridge"), ("fold_kernel", "anchor_fold"),
    ("sort_filter_seg_kernel", "anchor_filter"), ("sort_filter_long_kernel", "anchor_filter_long"),
    ("long_sort_low_kernel", "anchor_filter_long_sort_tiles"), ("long_sort_high_kernel", "anchor_filter_long_sort_rows"), ("long_prepare_kernel", "anchor_filter_long_keys"), ("sort_filter_kernel", "anchor_filter_general"),
    ("lean_overrun_direct_kernel", "anchor_overruns_direct"), ("lean_overrun_chain_kernel", "anchor_overruns_chain"),
    ("gather_lists_kernel", "export_gather"), ("block_export_kernel", "exchange_block_export"), ("block_attach_kernel", "exchange_block_attach"),
    ("check_lists_kernel", "exchange_check_lists"), ("compact_raw_kernel", "anchor_compact"),
    ("project_kernel<true>", "pileup_project5"), ("project_kernel", "pileup_project"), ("tile_index_kernel", "pileup_tile_index"),
    ("pairs_mfma_kernel", "pileup_pairs_mfma"), ("bang_correct_kernel", "pileup_bang_correct"),
    ("pairs_kernel<true>", "pileup_pairs_bang"), ("pairs_kernel<false>", "pileup_pairs"), ("pack_triangle_kernel", "result_pack_triangle"),
    ("sym32_from", "result_sym32"), ("symmetrise_kernel", "result_symmetrise"), ("seqcmp_pass_kernel", "seqcmp_pass"),
    ("seqcmp_rounds_kernel", "seqcmp_rounds"), ("seqcmp_one_kernel", "seqcmp_one"), ("triangle_rows_kernel", "result_rows"), ("matrices_to_home_kernel", "result_to_home"),
    ("triangle_to_host_kernel", "result_triangle_to_host"), ("lean_bridge_prepare_kernel", "anchor_bridge_prepare"),
    ("phase_a_report_kernel", "anchor_report"), ("lean_work_kernel", "anchor_work_items"),
]


def short(name):
    for key, val in NAMES:
        if key in name:
            return val
    return None


# kernel trace
agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0, None])
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):  # (one file per traced process)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r.get("Kernel_Name", ""))
            if not k:
                continue
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            a = agg[k]
            a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
            a[4] = {x: r.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                          "Workgroup_Size", "Grid_Size") if x in r}
if agg:
    out["kernel_trace"] = {k: {"calls": a[0], "total_ms": round(a[1], 4), "avg_ms": round(a[1] / a[0], 4),
                               "min_ms": round(a[2], 4), "max_ms": round(a[3], 4), "resources": a[4]} for k, a in agg.items()}
    tot = sum(a[1] for a in agg.values())
    for k in out["kernel_trace"]:
        out["kernel_trace"][k]["pct_of_phy_kernels"] = round(100 * out["kernel_trace"][k]["total_ms"] / tot, 2)

# PMC passes
pmc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r.get("Kernel_Name", ""))
            if not k:
                continue
            c = r.get("Counter_Name"); v = float(r.get("Counter_Value", 0))
            pmc[k][c][0] += 1; pmc[k][c][1] += v
out["pmc_per_dispatch_avg"] = {k: {c: round(v[1] / v[0], 3) for c, v in cs.items()} for k, cs in pmc.items()}
out["pmc_dispatches"] = {k: {c: v[0] for c, v in cs.items()} for k, cs in pmc.items()}
for f in glob.glob(os.path.join(d, "*_bench.json")):
    try:
        out.setdefault("bench_lines", {})[os.path.basename(f)] = json.load(open(f))
    except Exception:
        pass
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
ks = os.path.join(d, "kernel_stats.csv")
if os.path.exists(ks):
    out["rocprof_kernel_stats_csv"] = open(ks).read()
    json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
for sub in glob.glob(os.path.join(d, "*")):
    if os.path.isdir(sub):
        shutil.rmtree(sub)
print(json.dumps({k: v for k, v in out.items() if k not in ("bench_lines", "rocprof_kernel_stats_csv")}, indent=1))

#!/usr/bin/env python3
"""One line per file of a directory of bench / wall-clock JSON files (tools_round_bench.sh writes gpurun_out/final)."""
import glob, json, os, sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "final")
for f in sorted(glob.glob(os.path.join(d, "*.json"))):
    txt = open(f).read().strip()
    name = os.path.basename(f)
    if "wallclock" in name:
        w = json.loads(txt)
        print(name)
        for r in w["runs"]:
            print("   ", r["label"], r["wall_s_including_exec"], r["timing"][:160], "identical" if r["matrix_identical"] else "DIFFERENT")
        continue
    try:
        b = json.loads(txt.splitlines()[-1])
    except Exception as e:
        print(name, "unreadable:", e)
        continue
    print(name, b["value"], "Gbp/s", b["ms_per_step"], "ms", "no-profile", b.get("ms_per_step_noprofile"), "n_gpus", b["n_gpus"],
          "frac", (b.get("roofline") or {}).get("frac"), "b0", (b.get("roofline_b0") or {}).get("frac"), "valu", (b.get("roofline_valu") or {}).get("frac"),
          "cpu", (b.get("cpu_baseline") or {}).get("value"), {k: round(v["avg_ms"], 3) for k, v in b["kernels"].items()})

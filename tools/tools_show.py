#!/usr/bin/env python3
"""print the interesting fields of a bench.py JSON line read from stdin"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read())
print(tag, d["value"], "Gbp/s", d["ms_per_step"], "ms", d["phases_ms_per_step"], d.get("extra_ms"))
print("   ", {k: v["avg_ms"] for k, v in d["kernels"].items()})

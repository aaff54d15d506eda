#!/usr/bin/env python3
"""print the essentials of bench.py JSON lines"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    r = d["roofline"]
    print(f, round(d["value"], 2), "Gk/s", round(d["ms_per_step"], 1), "ms/step |", r["kernel"], round(r["frac"], 3), "of peak,", round(r["avg_launch_ms"] * 1e3, 1),
          "us/launch | last bin ms", {k: round(v, 3) for k, v in d["phases_ms_last_bin_slot0"].items()}, "|", d["self_check"], "| local sort", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.get("local_sort", {}).items() if k != "kernel"})

#!/usr/bin/env python3
"""print the essentials of a bench.py JSON line"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print(f, round(d["value"], 2), "Gk/s", round(d["ms_per_step"], 1), "ms", {k: round(v, 2) for k, v in d["phases_ms_last_step"].items()}, "scatter GB/s", round(d["roofline"]["achieved"]))

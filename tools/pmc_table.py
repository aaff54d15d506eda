#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 counter-collection CSV (one line per kernel)."""
import csv, re, sys
from collections import defaultdict
tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for row in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
    tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[name][row["Counter_Name"]] += 1
for k in sorted(tot):
    print(k, {c: round(tot[k][c] / cnt[k][c]) for c in sorted(tot[k])}, "launches", max(cnt[k].values()))

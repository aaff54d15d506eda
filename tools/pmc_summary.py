#!/usr/bin/env python3
"""Turns two rocprofv3 counter-collection CSVs (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass, same command) into the
per-kernel HBM traffic summary bench.py reads (profiles/<round>/pmc_hbm_traffic.json).

Units and corrections as MI355X_MICROARCH.md §HBM prescribes and as calibrated on tools/ubench_scatter.hip: the counters
are KiB; FETCH_SIZE reports half of the bytes read on gfx950 (x2), WRITE_SIZE is exact.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <records_per_onesweep_launch> <out.json>
"""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
        tot[name] += float(row["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, recs, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    f_tot, f_cnt = per_kernel(fetch_csv, "FETCH_SIZE")
    w_tot, w_cnt = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in f_tot:
        n = f_cnt[k]
        rd = 2.0 * f_tot[k] * 1024.0 / n
        wr = w_tot.get(k, 0.0) * 1024.0 / max(w_cnt.get(k, 0), 1)
        kernels[k] = {"launches": n, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 1 --warmup 0 "
                       "--no-cpu-baseline --no-secondary --no-host-boundary --no-digest` (configs[2], one stream). Counters are KiB; FETCH_SIZE x2 (gfx950 reports half, calibrated on tools/ubench_scatter.hip), "
                       "WRITE_SIZE exact.",
               "kernels": kernels, "records_per_launch_avg": recs}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:40s} x{v['launches']:<4d} read {v['hbm_read_bytes_per_launch'] / 1e9:8.3f} GB  write {v['hbm_write_bytes_per_launch'] / 1e9:8.3f} GB")


if __name__ == "__main__":
    main()

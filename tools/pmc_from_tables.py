#!/usr/bin/env python3
"""Two tools/pmc_table.py outputs (one `rocprofv3 --pmc FETCH_SIZE` pass, one `--pmc WRITE_SIZE` pass over the same bench command: tools/gpu_session.sh pmck)
-> the per-kernel HBM traffic summary bench.py reads (profiles/<round>/pmc_hbm_traffic.json). Units and corrections as MI355X_MICROARCH.md prescribes and as
calibrated on tools/ubench_scatter.hip: the counters are KiB per launch (averaged here); FETCH_SIZE reports half of the bytes read on gfx950 (x2), WRITE_SIZE is exact.
usage: pmc_from_tables.py fetch.txt write.txt bench_line.json out.json "workload description" """
import ast
import json
import re
import sys


def table(path, counter):
    out = {}
    for ln in open(path):
        m = re.match(r"(.+?) (\{.*\}) launches (\d+)", ln.strip())
        if m:
            out[m.group(1)] = (ast.literal_eval(m.group(2)).get(counter, 0), int(m.group(3)))
    return out


def main():
    f, w = table(sys.argv[1], "FETCH_SIZE"), table(sys.argv[2], "WRITE_SIZE")
    line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    kernels = {}
    for k in f:
        rd, wr = 2.0 * f[k][0] * 1024.0, w.get(k, (0, 0))[0] * 1024.0
        kernels[k] = {"launches": f[k][1], "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    rpl = line["roofline"]["records_per_launch"]
    groups = kernels.get("k_parse_packs", kernels[next(iter(kernels))])["launches"]  # one parse launch per group, whatever the record width
    per_group = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in kernels.items()) / max(groups, 1)
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --k K --leg quarter --steps 1 --warmup 0 --no-digest` "
                       "(tools/gpu_session.sh pmck). Counters are KiB, averaged per launch; FETCH_SIZE x2 (gfx950 reports half, calibrated on tools/ubench_scatter.hip in "
                       "round 1), WRITE_SIZE exact.",
               "workload": sys.argv[5] if len(sys.argv) > 5 else "", "records_per_launch_avg": rpl, "hbm_bytes_per_group_all_kernels": per_group,
               "hbm_bytes_per_kmer_all_kernels": per_group / rpl, "kernels": kernels}, open(sys.argv[4], "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:36s} x{v['launches']:<4d} read {v['hbm_read_bytes_per_launch'] / 1e9:8.3f} GB  write {v['hbm_write_bytes_per_launch'] / 1e9:8.3f} GB")
    print("per group of %.1f M records: %.2f GB = %.1f B per k-mer" % (rpl / 1e6, per_group / 1e9, per_group / rpl))


if __name__ == "__main__":
    main()

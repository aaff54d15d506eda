"""Where does the finisher's work lie? The bucket-size spectrum of a synthetic workload as k_bucket_rank sees it: bins of the quarter workload are expanded on the
device (kmc_hip_debug_expand), sorted (torch.sort), and the runs of equal top key bits (the buckets the four HBM passes leave: 30 bits at k = 27, groups of 4
bins) and of equal k-mers are measured. Printed per size class: share of the records, and of the pairwise work (sum of squares). Usage:
  KMC_SYNTH_REPEATS=... python tools/bucket_hist.py [--k 27] [--bins-sampled 4]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmc_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=27)
ap.add_argument("--reads", type=int, default=50_000_000)
ap.add_argument("--genome", type=int, default=250_000_000)
ap.add_argument("--bins", type=int, default=128)
ap.add_argument("--bins-sampled", type=int, default=4)
ap.add_argument("--bucket-bits", type=int, default=30)
a = ap.parse_args()
sb = capi.synth_bins(seed=2026, genome_len=a.genome, n_reads=a.reads, k=a.k, n_bins=a.bins, copy=False)
ctx = capi.Context((0,))
p = capi.make_params(a.k)
edges = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 6144, 16384, 65536, 1 << 20, 1 << 40]
tot = {"records": 0}
acc = {name: np.zeros(len(edges) - 1) for name in ("bucket_records", "bucket_squares", "kmer_records")}
order = sorted(range(a.bins), key=lambda b: -sb.bins[b][1])
picks = order if a.bins_sampled >= a.bins else [order[(i * a.bins) // a.bins_sampled] for i in range(a.bins_sampled)]  # evenly over the size-sorted bins, the largest first
for b in picks:
    img, nrec, packs, _ = sb.bins[b]
    recs = ctx.debug_expand(p, np.ascontiguousarray(img), nrec, np.ascontiguousarray(packs))[:, 0]
    t = torch.from_numpy(recs.view(np.int64)).cuda()
    t, _ = torch.sort(t)
    shift = 2 * a.k - a.bucket_bits
    for name, keys in (("bucket", t >> shift), ("kmer", t)):
        _, counts = torch.unique_consecutive(keys, return_counts=True)
        c = counts.cpu().numpy().astype(np.float64)
        idx = np.searchsorted(np.array(edges[1:], dtype=np.float64), c, side="right")
        if name == "bucket":
            acc["bucket_records"] += np.bincount(idx, weights=c, minlength=len(edges) - 1)[: len(edges) - 1]
            acc["bucket_squares"] += np.bincount(idx, weights=c * c, minlength=len(edges) - 1)[: len(edges) - 1]
        else:
            acc["kmer_records"] += np.bincount(idx, weights=c, minlength=len(edges) - 1)[: len(edges) - 1]
    tot["records"] += nrec
print("workload: k=%d, %d reads of a %d bp genome, %d bins, %d sampled (evenly over the size-sorted order); KMC_SYNTH_REPEATS=%s; buckets = top %d key bits" % (
    a.k, a.reads, a.genome, a.bins, min(a.bins_sampled, a.bins), os.environ.get("KMC_SYNTH_REPEATS", ""), a.bucket_bits))
print("%-18s %14s %14s %14s" % ("size class", "records in", "pair work in", "records in"))
print("%-18s %14s %14s %14s" % ("(records)", "buckets of it", "buckets of it", "k-mers of it"))
for i in range(len(edges) - 1):
    if acc["bucket_records"][i] or acc["kmer_records"][i]:
        print("%-18s %13.2f%% %13.2f%% %13.2f%%" % ("[%d, %d)" % (edges[i], edges[i + 1]), 100 * acc["bucket_records"][i] / tot["records"],
                                                     100 * acc["bucket_squares"][i] / acc["bucket_squares"].sum(), 100 * acc["kmer_records"][i] / tot["records"]))
print(json.dumps({"records": tot["records"], "mean_pair_work_per_record": acc["bucket_squares"].sum() / tot["records"]}))

#!/usr/bin/env python3
"""Extra measurements for DESIGN.md: other k (config 5 record widths) and the many-bins regime (configs 2-3),
all device-resident, same timing method as bench.py. Prints one JSON line per case."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import capi

ctx = capi.Context((0,))


def upload(bins):
    dev = []
    for img, nrec, packs, _ in bins:
        ps = np.concatenate([[0], np.cumsum(packs)]).astype(np.uint64)
        d_in = ctx.malloc(img.size + 256); ctx.h2d(d_in, img)
        d_ps = ctx.malloc(ps.nbytes); ctx.h2d(d_ps, ps)
        dev.append((d_in, img.size, nrec, d_ps, packs.size))
    return dev


def run(name, k, reads, genome, n_bins, p_len, steps=3):
    bins = capi.synth_bins(seed=2026, genome_len=genome, n_reads=reads, k=k, n_bins=n_bins)
    p = capi.make_params(k, lut_prefix_len=p_len)
    rec = ctx.out_rec_bytes(p)
    dev = upload(bins)
    n_tot = sum(b[1] for b in bins)
    cap = max(((b[1] + 1) // 2) * rec for b in bins) + 256
    d_out = ctx.malloc(cap); d_lut = ctx.malloc(max(ctx.lut_entries(p), 1) * 8); d_small = ctx.malloc(64 * n_bins)
    def step():
        for i, (d_in, size, nrec, d_ps, npk) in enumerate(dev):
            ctx.process_bin_device(p, d_in, size, nrec, d_ps, npk, d_out, cap, d_small + 64 * i + 32, d_lut, d_small + 64 * i, sync=False)
        ctx.synchronize()
    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    st = np.zeros(8 * n_bins, dtype=np.uint64); ctx.d2h(st, d_small)
    st = st.reshape(n_bins, 8)
    W = 8 * ((k + 31) // 32); P = (2 * k + 7) // 8
    print(json.dumps({"case": name, "k": k, "bins": n_bins, "kmers": n_tot, "ms_per_step": dt * 1e3, "Gkmers_per_s": n_tot / dt / 1e9,
                      "unique": int(st[:, 0].sum()), "total_check": int(st[:, 3].sum()) == n_tot, "record_bytes": W, "passes": P,
                      "algorithmic_GBs": W * (2 * P + 3) * n_tot / dt / 1e9, "phases_last_bin_ms": ctx.last_timings()}), flush=True)
    for d in dev:
        ctx.free(d[0]); ctx.free(d[3])
    ctx.free(d_out); ctx.free(d_lut); ctx.free(d_small)


cases = sys.argv[1:] or ["k55", "k127", "bins64", "bins512", "bins512small"]
for c in cases:
    if c == "k55":
        run("k=55 single bin (2 Gbp)", 55, 13_300_000, 66_000_000, 1, 3)
    elif c == "k127":
        run("k=127 single bin (2 Gbp)", 127, 13_300_000, 66_000_000, 1, 3)
    elif c == "bins64":
        run("k=27, 2 Gbp in 64 bins", 27, 13_300_000, 66_000_000, 64, 3)
    elif c == "bins512":
        run("k=27, 2 Gbp in 512 bins", 27, 13_300_000, 66_000_000, 512, 3)
    elif c == "bins512small":
        run("k=27, 0.3 Gbp in 512 bins (the 248 M k-mer e2e case)", 27, 2_000_000, 10_000_000, 512, 3)

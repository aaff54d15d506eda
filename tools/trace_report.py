#!/usr/bin/env python3
"""Per-tile phase table from a trace written by tools/trace_run.py (a -DKMC_TRACE build, see tools/build_variants.py).

usage: trace_report.py gpurun_out/trace_<name>.npy [first_tile last_tile]

Stamp layout (kernels.hip.h, TRACE_STAMP(kind, tile, j); kind 0 = k_onesweep thread 0, kind 1 = k_compact thread 0,
kind 2 = k_onesweep thread 0 in its role as owner of digit 0):
  kind 0: 0 start (wall clock, 10 ns units, comparable across CUs)  1 loads landed  2 counted + first barrier
          3 digit totals / aggregate out / slot bases  4 ranked  5 exclusive prefix known (look-back done)
          6 first LDS scatter + barrier  7 stores issued           (1..7: shader cycle counter, ~0.47 ns, per XCD)
  kind 2: 1 look-back start  2 look-back end  3 round trips  4 spins on unpublished tiles  5 tiles walked
          7 wall clock when the aggregate was published
  kind 1: 0 start (wall)  2 loads + run flags  3 block max-scan  4 cross-tile run start  5 classify + block sum
          6 look-back  7 emitted
"""
import sys

import numpy as np

TICK_US = 0.47e-3  # shader cycle counter tick in microseconds (calibrated against the wall clock stamps)


def main():
    t = np.load(sys.argv[1])
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
    k0, k1, k2 = (t[i].astype(np.int64)[lo:hi] for i in (0, 1, 2))

    ok = (k0[:, 7] > 0) & (k0[:, 1] > 0)
    a, b = k0[ok], k2[ok]
    if len(a):
        d = np.diff(a[:, 1:8], axis=1) * TICK_US
        names = ["count + barrier", "totals, aggregate, slot bases", "ranking", "look-back (thread 0)", "LDS scatter + barrier", "stores"]
        print(f"k_onesweep, {len(a)} tiles, microseconds after the loads have landed (median / p90):")
        for n, c in zip(names, d.T):
            print(f"  {n:32s} {np.median(c):6.2f} {np.percentile(c, 90):6.2f}")
        print(f"  {'total':32s} {np.median(a[:, 7] - a[:, 1]) * TICK_US:6.2f}")
        s0 = a[:, 0] / 100.0
        print(f"  tiles started per microsecond: {len(a) / max(s0.max() - s0.min(), 1e-9):.1f}")
        lb = (b[:, 2] - b[:, 1]) * TICK_US
        has = b[:, 2] > 0
        if has.any():
            print(f"  look-back: {np.median(lb[has]):.2f} us, {b[has, 3].mean():.2f} round trips, {b[has, 4].mean():.2f} spins, "
                  f"{np.median(b[has, 5]):.0f} tiles walked (p90 {np.percentile(b[has, 5], 90):.0f})")
        agg = b[:, 7] / 100.0
        if (agg > 0).all():
            x = np.arange(len(agg))
            res = agg - np.polyval(np.polyfit(x, agg, 1), x)
            print(f"  start -> aggregate published: {np.median(agg - s0):.2f} us (jitter std {res.std():.2f} us)")
            for kk in (1, 8, 32):
                print(f"    P(tile-{kk} publishes after me) = {((agg[kk:] - agg[:-kk]) < 0).mean():.2f}")

    ok = k1[:, 7] > 0
    c = k1[ok]
    if len(c):
        d = np.diff(c[:, 1:8], axis=1) * TICK_US
        names = ["loads + run flags", "block max-scan", "cross-tile run start", "classify + block sum", "look-back", "emit"]
        print(f"k_compact, {len(c)} tiles, microseconds (median / p90):")
        for n, col in zip(names, d.T):
            print(f"  {n:32s} {np.median(col):6.2f} {np.percentile(col, 90):6.2f}")
        print(f"  {'total':32s} {np.median(c[:, 7] - c[:, 1]) * TICK_US:6.2f}")
        s0 = c[:, 0] / 100.0
        print(f"  tiles started per microsecond: {len(c) / max(s0.max() - s0.min(), 1e-9):.1f}")


if __name__ == "__main__":
    main()

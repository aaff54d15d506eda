#!/usr/bin/env python3
"""Sort-only sweep over n (device-resident random 54-bit keys): per-launch scatter time and GB/s vs working-set size."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import capi  # noqa: E402

ctx = capi.Context((0,))
rng = np.random.default_rng(1)
for logn in [20, 22, 23, 24, 25, 26, 27, 28, 29]:
    n = 1 << logn
    a = rng.integers(0, 2**54, size=n, dtype=np.uint64)
    d_a, d_b = ctx.malloc(n * 8 + 256), ctx.malloc(n * 8 + 256)
    best = None
    for rep in range(3):
        ctx.h2d(d_a, a)
        ctx.sort_records_device(d_a, d_b, n, 1, 7)
        nl, ms, keys = ctx.scatter_totals(reset=True); keys = keys // max(nl, 1)
        if best is None or ms < best:
            best = ms
    print(f"n=2^{logn} ({n*16>>20:6d} MB in+out): {nl} launches, {best:8.3f} ms total, {best/nl*1e3:9.1f} us/launch, {16*n*nl/best/1e6:8.1f} GB/s")
    ctx.free(d_a)
    ctx.free(d_b)

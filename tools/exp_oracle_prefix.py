#!/usr/bin/env python3
"""Experiment: scatter time with the look-back replaced by replayed prefixes (ceiling of any look-back optimisation)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import capi
ctx = capi.Context((0,))
n = 1 << 29
rng = np.random.default_rng(1)
a = rng.integers(0, 2**54, size=n, dtype=np.uint64)
d_a, d_b = ctx.malloc(n * 8 + 256), ctx.malloc(n * 8 + 256)
L = ctx.L
L.kmc_hip_debug_oracle_prefix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64]
tiles = 7 * ((n + 8191) // 8192) + 64
for mode, name in ((0, "normal look-back"), (1, "record"), (2, "replay (free look-back)"), (2, "replay again"), (0, "normal again")):
    assert L.kmc_hip_debug_oracle_prefix(ctx.h, 0, mode, tiles) == 0
    ctx.h2d(d_a, a)
    res = ctx.sort_records_device(d_a, d_b, n, 1, 7)
    nl, ms, keys = ctx.last_scatter_stats()
    out = np.empty(n, dtype=np.uint64); ctx.d2h(out, res)
    ok = bool(np.all(out[:-1] <= out[1:]))
    print(f"{name:28s}: {ms:8.3f} ms for {nl} launches, {16*n*nl/ms/1e6:8.1f} GB/s, sorted={ok}")

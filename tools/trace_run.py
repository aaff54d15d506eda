#!/usr/bin/env python3
"""Runs one bin through a -DKMC_TRACE build and dumps per-tile phase stamps to gpurun_out/trace_<name>.npy."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import capi  # noqa: E402

name = sys.argv[1]
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
ctx = capi.Context((0,))
(img, nrec, packs, _), = capi.synth_bins(seed=2026, genome_len=20_000_000, n_reads=reads, k=27, n_bins=1)
p = capi.make_params(27)
for _ in range(2):
    out, lut, st = ctx.process_bin(p, img, nrec, packs)
L = ctx.L
L.kmc_hip_debug_read_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int]
buf = np.zeros((1 << 17) * 8, dtype=np.uint64)
assert L.kmc_hip_debug_read_trace(ctx.h, 0, buf.ctypes.data_as(C.c_void_p), buf.size, 0) == 0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", f"trace_{name}.npy"), buf.reshape(4, -1, 8))
print(name, nrec, ctx.last_timings())

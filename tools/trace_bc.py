#!/usr/bin/env python3
"""Phase times of k_bucket_count from a -DKMC_TRACE build (kmc_amd/variants/libkmc_hip_trace.so): thread 0 of every tile adds the 100 MHz clock
ticks between its phase stamps to ten counters. Usage: KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_trace.so python tools/trace_bc.py [n_bins]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import capi  # noqa: E402

n_bins = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 27
PL = int(sys.argv[3]) if len(sys.argv) > 3 else 7
ctx = capi.Context((0,))
bins = capi.synth_bins(seed=2026, genome_len=1_000_000_000 * n_bins // 512, n_reads=200_000_000 * n_bins // 512, k=K, n_bins=n_bins)
p = capi.make_params(K, lut_prefix_len=PL)
rec, nl = ctx.out_rec_bytes(p), ctx.lut_entries(p)
descs = (capi.BinDesc * n_bins)()
for i, (img, nrec, packs, _) in enumerate(bins):
    ps = np.concatenate([[0], np.cumsum(packs)]).astype(np.uint64)
    cap = ((nrec + 1) // 2) * rec
    d_in, d_ps, d_out, d_lut, d_small = ctx.malloc(img.size + 256), ctx.malloc(ps.nbytes), ctx.malloc(cap + 256), ctx.malloc(nl * 8), ctx.malloc(64)
    ctx.h2d(d_in, np.concatenate([img, np.zeros(256, dtype=np.uint8)]))
    ctx.h2d(d_ps, ps)
    descs[i] = capi.BinDesc(d_in, img.size, nrec, d_ps, packs.size, d_out, cap, d_small + 32, d_lut, d_small)
L = ctx.L
if not hasattr(L, "kmc_hip_debug_read_trace"):
    for it in range(2):
        ctx.process_bins_device(p, descs, 1)
        ctx.synchronize()
    print("k", K, "bins", n_bins, "local sort", ctx.local_sort_totals())
    sys.exit(0)
L.kmc_hip_debug_read_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int]
buf = np.zeros((1 << 17) * 8, dtype=np.uint64)
for it in range(2):
    L.kmc_hip_debug_read_trace(ctx.h, 0, buf.ctypes.data_as(C.c_void_p), buf.size, 1)
    ctx.process_bins_device(p, descs, 1)
    ctx.synchronize()
assert L.kmc_hip_debug_read_trace(ctx.h, 0, buf.ctypes.data_as(C.c_void_p), buf.size, 0) == 0
ph = buf.reshape(4, -1, 8)[3].ravel()[:10].astype(np.float64)
names = ["clear+load+heads", "sub-buckets+count", "region scan", "probing", "classify", "counted scan", "ranks", "stage", "copy-out+lut", "tallies"]
tot = ph.sum()
ls = ctx.local_sort_totals()
print("local sort", ls)
for n, v in zip(names, ph):
    print(f"{n:20s} {v / tot * 100:5.1f} %")

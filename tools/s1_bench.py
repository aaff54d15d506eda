"""First timing of the stage-1 groundwork (docs/history/DESIGN_rounds_1_to_5.md §9) on one MI355X: reads already in HBM as codes -> bins in HBM.
Wall-clock of kmc_hip_split_reads_plan and _emit (both synchronise), best of --reps; run it under `rocprofv3 --kernel-trace --stats` for the
per-kernel durations. Not part of bench.py: stage 1 is not in the drop-in. No torch: numpy + the C-ABI only."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmc_amd import capi  # noqa: E402


def make_codes(n_symbols, read_len, seed):
    """reads of uniformly random symbols joined by one boundary byte: what reads of a random genome look like to stage 1 (no quality, no N)"""
    rng = np.random.default_rng(seed)
    n_reads = n_symbols // (read_len + 1)
    codes = rng.integers(0, 4, size=n_reads * (read_len + 1), dtype=np.int8)
    codes[read_len::read_len + 1] = -1
    return codes, n_reads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--symbols", type=int, default=300_000_000)
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--m", type=int, default=9)
    ap.add_argument("--bins", type=int, default=512)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    t = time.time()
    codes, n_reads = make_codes(a.symbols, 150, 1)
    smap = np.random.default_rng(2).integers(0, a.bins, size=(1 << (2 * a.m)) + 1).astype(np.int32)
    gen_s = time.time() - t
    capi.require_gpu_backend()
    ctx = capi.Context((0,))
    L, h = ctx.L, ctx.h
    L.kmc_hip_split_reads_free.restype = None
    d_codes = ctx.malloc(codes.size + 256)
    d_map = ctx.malloc(smap.nbytes)
    ctx.h2d(d_codes, codes)
    ctx.h2d(d_map, smap)
    nb = a.bins
    base, pbase = np.zeros(nb + 1, dtype=np.uint64), np.zeros(nb + 1, dtype=np.uint64)
    by, sk, km = (np.zeros(nb, dtype=np.uint64) for _ in range(3))
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    best = None
    for rep in range(a.reps + 1):  # the first round is a warm-up on a prefix (code objects, allocator)
        n = min(codes.size, 151 * 10_000) if rep == 0 else codes.size
        plan = C.c_void_p()
        t0 = time.perf_counter()
        ctx._chk(L.kmc_hip_split_reads_plan(h, 0, C.c_void_p(d_codes), C.c_uint64(n), C.c_uint32(a.k), C.c_uint32(a.m), C.c_void_p(d_map), C.c_uint32(nb),
                                            C.byref(plan), vp(base), vp(by), vp(sk), vp(km), vp(pbase)))
        t1 = time.perf_counter()
        d_bins = ctx.malloc(int(base[nb]))
        d_ps = ctx.malloc(int(pbase[nb]) * 8)
        t2 = time.perf_counter()
        ctx._chk(L.kmc_hip_split_reads_emit(h, plan, C.c_void_p(d_bins), C.c_void_p(d_ps)))
        t3 = time.perf_counter()
        L.kmc_hip_split_reads_free(h, plan)
        ctx.free(d_bins)
        ctx.free(d_ps)
        if rep and (best is None or (t1 - t0) + (t3 - t2) < best[0] + best[1]):
            best = (t1 - t0, t3 - t2)
    n = codes.size
    out = dict(what="stage-1 groundwork: codes in HBM -> bins in HBM (wall-clock of the two C-ABI calls, synchronous, incl. their hipMalloc/hipFree)",
               symbols=int(n), reads=int(n_reads), k=a.k, signature_len=a.m, bins=nb, superkmers=int(sk.sum()), kmers=int(km.sum()), bin_bytes=int(by.sum()),
               packs=int(pbase[nb]) - nb, plan_s=best[0], emit_s=best[1], gsymbols_per_s=n / (best[0] + best[1]) / 1e9,
               gkmers_per_s=float(km.sum()) / (best[0] + best[1]) / 1e9, generate_s=gen_s)
    print(json.dumps(out))
    ctx.free(d_codes)
    ctx.free(d_map)
    ctx.close()


if __name__ == "__main__":
    main()

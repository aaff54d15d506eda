#!/usr/bin/env python3
"""Host-boundary facts of the GPU box for DESIGN.md §6: how fast is hipHostRegister (is pinning the CMemoryBins arena
affordable?), and what do H2D / D2H give from pageable vs registered vs hipHostMalloc memory. Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import capi  # noqa: E402

ctx = capi.Context((0,))
L = ctx.L
res = {"cpus": os.cpu_count()}
n = 1 << 30
d = ctx.malloc(n)


def rate(f, nbytes, reps=3):
    best = None
    for _ in range(reps):
        t = time.perf_counter()
        f()
        dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best
    return nbytes / best / 1e9


a = np.ones(n, dtype=np.uint8)  # touched pageable memory
res["h2d_pageable_GBs"] = rate(lambda: ctx.h2d(d, a), n)
res["d2h_pageable_GBs"] = rate(lambda: ctx.d2h(a, d), n)
t = time.perf_counter()
rc = L.kmc_hip_host_register(ctx.h, C.c_void_p(a.ctypes.data), n)
res["host_register_GBs"] = n / (time.perf_counter() - t) / 1e9 if rc == 0 else None
if rc == 0:
    res["h2d_registered_GBs"] = rate(lambda: ctx.h2d(d, a), n)
    res["d2h_registered_GBs"] = rate(lambda: ctx.d2h(a, d), n)
    t = time.perf_counter()
    L.kmc_hip_host_unregister(ctx.h, C.c_void_p(a.ctypes.data))
    res["host_unregister_s_per_GiB"] = time.perf_counter() - t
t = time.perf_counter()
p = ctx.host_alloc(n)
res["host_alloc_GBs"] = n / (time.perf_counter() - t) / 1e9
t = time.perf_counter()
p[:] = 1
res["pinned_first_touch_GBs"] = n / (time.perf_counter() - t) / 1e9
res["h2d_pinned_GBs"] = rate(lambda: ctx.h2d(d, p), n)
res["d2h_pinned_GBs"] = rate(lambda: ctx.d2h(p, d), n)
t = time.perf_counter()
p[:] = a
res["memcpy_1thread_GBs"] = n / (time.perf_counter() - t) / 1e9
ctx.host_free(p)
ctx.free(d)
print(json.dumps(res))

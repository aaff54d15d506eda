"""What does unmapping KMC's stage-2 arena cost INSIDE a process that has copied from it with HIP? (tools/ubench_munmap.c: 0.15 s for 2.3 GB touched in a plain
process, 0.02 s when 16 threads return the pages first.) An anonymous 14 GB mapping, 512 pieces of 3.4 MB + 1.1 MB touched; variants: untouched by HIP / every piece
copied to the device from where it lies (pageable: the runtime pins user pages on the fly) / copied through a pinned staging buffer; then madvise(DONTNEED) by 16
threads or not; then munmap."""
import ctypes as C
import mmap
import sys
import threading
import time

sys.path.insert(0, ".")
import numpy as np
from kmc_amd import capi

libc = C.CDLL("libc.so.6", use_errno=True)
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
TOTAL, PIECE, STRIDE, N = 14 << 30, 4_500_000, 26_000_000, 512
ctx = capi.Context((0,))
d = ctx.malloc(PIECE)
pin = ctx.host_alloc(PIECE)
for variant in ("no hip copies", "hipMemcpy from the mapping", "memcpy to pinned, hipMemcpy from there", "d2h: hipMemcpy INTO the mapping (1.1 MB pieces)", "d2h small: hipMemcpy INTO the mapping (128 KB pieces)"):
    for zap in (0, 16):
        m = mmap.mmap(-1, TOTAL, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        a = np.frombuffer(m, dtype=np.uint8)
        base = a.ctypes.data
        libc.madvise(C.c_void_p(base), TOTAL, 14)  # MADV_HUGEPAGE
        t0 = time.perf_counter()
        for i in range(N):
            a[i * STRIDE:i * STRIDE + PIECE] = 1
        t1 = time.perf_counter()
        for i in range(N):
            piece = a[i * STRIDE:i * STRIDE + PIECE]
            if variant.startswith("hipMemcpy"):
                ctx.h2d(d, piece)
            elif variant.startswith("memcpy"):
                pin[:PIECE] = piece
                ctx.h2d(d, pin[:PIECE])
            elif variant.startswith("d2h:"):
                ctx.d2h(piece[:1_100_000], d)
            elif variant.startswith("d2h small"):
                for q in range(8):
                    ctx.d2h(piece[q * 131072:(q + 1) * 131072], d)
        t2 = time.perf_counter()
        tz = 0.0
        if zap:
            per = (TOTAL // zap) & ~((2 << 20) - 1)
            ths = [threading.Thread(target=lambda t=t: libc.madvise(C.c_void_p(base + 4096 + t * per), per - 4096, 4)) for t in range(zap)]  # MADV_DONTNEED
            z0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            tz = time.perf_counter() - z0
        del a, piece
        t3 = time.perf_counter()
        m.close()
        t4 = time.perf_counter()
        print("%-42s touch %.3f s, copies %.3f s, zap(%2d threads) %.3f s, munmap %.3f s" % (variant, t1 - t0, t2 - t1, zap, tz, t4 - t3), flush=True)

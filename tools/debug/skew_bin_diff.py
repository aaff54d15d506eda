"""One bin of the skew quarter workload through the shipped library and against the oracle: where do the outputs differ?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from kmc_amd import capi  # noqa: E402
from test_gpu_parity import _run_batch  # noqa: E402

os.environ["KMC_SYNTH_REPEATS"] = "10000:2000:10"
sb = capi.synth_bins(seed=2026, genome_len=250_000_000, n_reads=50_000_000, k=27, n_bins=128, copy=False)
ctx = capi.Context((0,))
pl = 7
p = capi.make_params(27, lut_prefix_len=pl)
op = O.make_params(27, lut_prefix_len=pl)
for b in (111, 82):
    img, nrec, packs, _ = sb.bins[b]
    bins = [(np.ascontiguousarray(img), nrec, np.ascontiguousarray(packs), 0)]
    got, e = _run_batch(ctx, p, bins, 1)
    assert e is None, e
    w = O.process_bin(op, bins[0][0], nrec)
    g = got[0]
    print("bin", b, nrec, "stats", [int(x) for x in g[2]], [int(x) for x in w[2]], "lut equal", bool(np.array_equal(g[1], w[1])), "out sizes", g[0].size, w[0].size)
    rb = 6  # (27 - 7) / 4 = 5 suffix bytes + 1 counter byte
    a, c = g[0].reshape(-1, rb), w[0].reshape(-1, rb)
    n = min(len(a), len(c))
    d = np.flatnonzero((a[:n] != c[:n]).any(axis=1))
    print("  differing records:", d.size, "of", n, "first", d[:10])
    for i in d[:6]:
        print("   rec", int(i), "got", a[i].tolist(), "want", c[i].tolist(), "| prev", c[i - 1].tolist(), "next", c[i + 1].tolist() if i + 1 < n else None)
    if d.size:
        # are the differing records a permutation of each other (order problem) or counts?
        lo, hi = int(d[0]), int(d[min(d.size - 1, 200)]) + 1
        sa = sorted(map(tuple, a[lo:hi].tolist()))
        sc = sorted(map(tuple, c[lo:hi].tolist()))
        print("   window [%d, %d): same multiset of records: %s" % (lo, hi, sa == sc))
print("paths", ctx.path_counters())

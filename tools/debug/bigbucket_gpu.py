"""GPU check of the big-bucket path of k_bucket_rank: bins with buckets of a few hundred to a few thousand records against the oracle, per bin; first mismatch located."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from kmc_amd import capi  # noqa: E402
from test_gpu_parity import _run_batch  # noqa: E402

ctx = capi.Context((0,))
bad = 0
CASES = (("5000:6:0", 200_000, 40_000, 4), ("5000:5:3", 200_000, 40_000, 4), ("20000:4:2,3000:8:0", 400_000, 80_000, 4), ("2000:10:0", 200_000, 40_000, 4), ("2000:10:0,3000:20:5", 400_000, 80_000, 4), ("1000:100:10", 400_000, 80_000, 2), ("", 200_000, 40_000, 4))
if os.environ.get("BB_CASES"):
    CASES = tuple((c, 200_000, 40_000, 4) for c in os.environ["BB_CASES"].split())
for rep, glen, nreads, nb in (CASES[:1] if os.environ.get("BB_QUICK") else CASES):
    os.environ["KMC_SYNTH_REPEATS"] = rep
    bins = capi.synth_bins(seed=3, genome_len=glen, n_reads=nreads, k=27, n_bins=nb, n_threads=4)
    p = capi.make_params(27, lut_prefix_len=3)
    op = O.make_params(27, lut_prefix_len=3)
    got, e = _run_batch(ctx, p, bins, 1)
    assert e is None, e
    for i, (img, nrec, packs, _) in enumerate(bins):
        w = O.process_bin(op, img, nrec)
        ok = [bool(np.array_equal(got[i][j], w[j])) for j in range(3)]
        msg = ""
        if not all(ok):
            bad += 1
            a, b = got[i][0], w[0]
            n = min(a.size, b.size)
            d = np.flatnonzero(a[:n] != b[:n])
            msg = " out sizes %d / %d, first differing byte %s (record %s)" % (a.size, b.size, d[:1], d[:1] // 7)
        print(repr(rep), i, nrec, ok, [int(x) for x in got[i][2]], [int(x) for x in w[2]], msg)
print("paths", ctx.path_counters(), "BAD" if bad else "ALL-OK")

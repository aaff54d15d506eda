"""Where exactly does the big-bucket path go wrong? One bin alone; expected (k-mer, count) sequence from numpy; first mismatching output record -> its bucket, the tile it lies in, the big buckets of that tile."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kmc_amd import capi  # noqa: E402
from test_gpu_parity import _run_batch  # noqa: E402

os.environ["KMC_SYNTH_REPEATS"] = os.environ.get("BB_REP", "5000:6:0")
bins = capi.synth_bins(seed=3, genome_len=200_000, n_reads=40_000, k=27, n_bins=4, n_threads=4)
ctx = capi.Context((0,))
k, pl = 27, 3
p = capi.make_params(k, lut_prefix_len=pl)
S, CAP = 5632, 6144
for b in (1, 2):
    img, nrec, packs, _ = bins[b]
    recs = np.sort(ctx.debug_expand(p, img, nrec, packs)[:, 0])
    got, e = _run_batch(ctx, p, [bins[b]], 1)
    assert e is None, e
    out, lut, st = got[0]
    uk, cnt = np.unique(recs, return_counts=True)
    keep = cnt >= 2
    ek, ec = uk[keep], np.minimum(cnt[keep], 255)
    rb = (k - pl) // 4 + 1
    o = out.reshape(-1, rb)
    # expected records: suffix bytes high -> low, then the counter
    sfx = (k - pl) // 4
    exp = np.zeros((ek.size, rb), dtype=np.uint8)
    for j in range(sfx):
        exp[:, j] = (ek >> np.uint64(8 * (sfx - 1 - j))).astype(np.uint8)
    exp[:, sfx] = ec.astype(np.uint8)
    n = min(len(o), len(exp))
    d = np.flatnonzero((o[:n] != exp[:n]).any(axis=1))
    print("bin", b, "records", nrec, "expected out records", len(exp), "got", len(o), "stats", [int(x) for x in st], "expected unique", uk.size, "mismatching records", d.size)
    if not d.size:
        continue
    i = int(d[0])
    K = int(ek[i])
    bucket = recs >> np.uint64(24)
    bstart = int(np.searchsorted(bucket, K >> 24, side="left"))
    bend = int(np.searchsorted(bucket, K >> 24, side="right"))
    # tiles: tile j = the buckets that start in window [jS, (j+1)S)
    starts = np.flatnonzero(np.concatenate([[True], bucket[1:] != bucket[:-1]]))
    sizes = np.diff(np.concatenate([starts, [recs.size]]))
    win = bstart // S
    in_win = (starts >= win * S) & (starts < (win + 1) * S)
    t0 = int(starts[in_win][0])
    nxt = starts[starts >= (win + 1) * S]
    t1 = int(nxt[0]) if nxt.size else recs.size
    sz = sizes[in_win]
    print("  first mismatch at output record", i, "k-mer %x count %d; got %s want %s" % (K, int(ec[i]), o[i].tolist(), exp[i].tolist()))
    print("  its bucket [%d, %d) size %d; window %d; tile [%d, %d) length %d (CAP %d); buckets in tile %d, big (>128) %d: sizes %s; tile-relative bucket start %d" % (
        bstart, bend, bend - bstart, win, t0, t1, t1 - t0, CAP, sz.size, int((sz > 128).sum()), sz[sz > 128].tolist(), bstart - t0))
    # how many distinct k-mers in that bucket, their counts
    inb = recs[bstart:bend]
    u2, c2 = np.unique(inb, return_counts=True)
    print("  bucket's k-mers: %d distinct, counts %s" % (u2.size, c2.tolist()[:20]))
    # the got output around i
    print("  got records i-2..i+4:", o[max(i - 2, 0):i + 5].tolist())
    print("  exp records i-2..i+4:", exp[max(i - 2, 0):i + 5].tolist())

"""Child process of tools/hostlib_sanitize.sh (needs KMC_HIP_LIB = a sanitized build of the emulated host library): the round-3 and round-4 paths of the host
library on small inputs — groups through k_bucket_rank fused with the counting (one-word records with 32- and 64-bit pairs, two-word A/B pairs, wider records,
chunked tiles), a tile for k_giant_tiles, an enormous bucket (redo through LSD passes + k_compact), several bins per host-boundary call (pageable buffers:
the library's pinned staging) — against the oracle, so that AddressSanitizer sees every "device" access (device memory is the heap here) and ThreadSanitizer every
pair of GPU threads (OS threads here) that touch the same LDS word without a barrier between them. Test infrastructure only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_py as O
from kmc_amd import capi
from test_gpu_parity import _run_batch
ctx = capi.Context((0,))
def check(k, bins, **kw):
    p = capi.make_params(k, **kw)
    op = O.make_params(p.kmer_len, p.both_strands, p.cutoff_min, p.cutoff_max, p.counter_max, p.lut_prefix_len, p.output_type, p.without_output)
    got, err = _run_batch(ctx, p, bins, 1)
    assert err is None, err
    for i, (img, nrec, packs, _) in enumerate(bins):
        w = O.process_bin(op, img, nrec)
        assert all(np.array_equal(a, b) for a, b in zip(got[i], w)), (k, i)
check(27, capi.synth_bins(seed=7, genome_len=3000, n_reads=300, k=27, n_bins=4, n_threads=1), lut_prefix_len=3)
check(32, capi.synth_bins(seed=7, genome_len=3000, n_reads=300, k=32, n_bins=2, n_threads=1), lut_prefix_len=4)
check(27, capi.synth_bins(seed=3, genome_len=400, n_reads=400, k=27, n_bins=4, err=0.0, n_threads=1), lut_prefix_len=3)   # chunks
check(27, capi.synth_bins(seed=5, genome_len=300, n_reads=1500, k=27, n_bins=2, err=0.01, n_threads=1), lut_prefix_len=3, cutoff_min=1)  # one k-mer more often than a tile holds records: k_giant_tiles
check(27, capi.synth_bins(seed=5, genome_len=160, n_reads=2600, k=27, n_bins=2, err=0.0, n_threads=1), lut_prefix_len=3)  # ... more often than k_giant_tiles takes: redo
check(55, capi.synth_bins(seed=7, genome_len=3000, n_reads=250, k=55, n_bins=4, n_threads=1), lut_prefix_len=3)   # two-word records: (A, B) pairs
check(127, capi.synth_bins(seed=7, genome_len=3000, n_reads=250, k=127, n_bins=3, n_threads=1), lut_prefix_len=3)  # four-word records: whole records compared
check(55, capi.synth_bins(seed=5, genome_len=300, n_reads=1200, k=55, n_bins=2, err=0.0, n_threads=1), lut_prefix_len=3)  # k_giant_tiles<2>
bins = capi.synth_bins(seed=9, genome_len=3000, n_reads=300, k=27, n_bins=4, n_threads=1)
got = ctx.process_bins_host(capi.make_params(27, lut_prefix_len=3), [(b[0], b[1], b[2]) for b in bins])
for i, b in enumerate(bins):
    w = O.process_bin(O.make_params(27, lut_prefix_len=3), b[0], b[1])
    assert all(np.array_equal(x, y) for x, y in zip(got[i], w)), i
print("SANITIZE-RUN-OK", ctx.local_sort_totals(), ctx.path_counters())

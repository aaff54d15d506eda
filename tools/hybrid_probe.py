#!/usr/bin/env python3
"""One synthetic bin through kmc_hip_process_bin with the hybrid sort: did k_bucket_count take it (hybrid_groups) or did the host have to run it again
with plain LSD passes (redo_groups)? Usage: tools/hybrid_probe.py K READS [GENOME]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kmc_amd import capi  # noqa: E402

k, reads = int(sys.argv[1]), int(sys.argv[2])
genome = int(sys.argv[3]) if len(sys.argv) > 3 else reads * 5
ctx = capi.Context((0,))
(img, nrec, packs, _), = capi.synth_bins(seed=3, genome_len=genome, n_reads=reads, k=k, n_bins=1)
p = capi.make_params(k)
out, lut, st = ctx.process_bin(p, img, nrec, packs)
print(k, nrec, st.tolist(), ctx.local_sort_totals(), ctx.last_timings())

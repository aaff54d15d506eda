"""CPU suite: the register / scratch budget of the hot kernels of the shipped build (hipcc cross-compiles gfx950 without a GPU). A spilled register in a
kernel that runs at the HBM roofline is HBM traffic nobody asked for (round 3: k_onesweep<1> 4 VGPRs / 12 B per lane, k_compact<1> 30 / 124 B): the kernels
of the default path stay at ZERO bytes of scratch, the others may not grow. profiles/r06/resource_usage.txt is the table of the committed tree."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# kernel -> (max scratch bytes per lane, min waves per SIMD)
BUDGET = {
    "k_onesweep<1>": (0, 8), "k_onesweep<2>": (0, 8), "k_onesweep<4>": (20, 8),          # the dominant kernel of every record width: 2 workgroups of 1024 per CU
    "k_bucket_rank<1, true>": (0, 6), "k_bucket_rank<2, true>": (0, 6), "k_bucket_rank<4, true>": (0, 6),  # two workgroups of 768 per CU
    "k_expand<1, true>": (0, 8), "k_expand<2, true>": (0, 8), "k_expand<4, true>": (0, 8),  # four workgroups of 512 per CU (LDS: 39 KB at k = 27)
    "k_parse_packs": (0, 7),  # one wave per pack, 5.7 KB of LDS each: 28 per CU either way
    "k_bucket_bounds<1>": (0, 8), "k_compact_fold": (0, 4), "k_compact_gather": (0, 8),
    # round 6: the arena of the repeat-rich buckets (arena_sort.hip.h). With nothing listed its launches return at once; the pass kernel is k_onesweep<1>'s tile body in a loop
    "k_bucket_detect": (0, 8), "k_arena_plan": (0, 8), "k_arena_gather": (0, 8), "k_arena_finish": (0, 4), "k_onesweep_dyn<1>": (24, 8),
    # off the default path since round 4 (redo / LSD runs; KMC_HIP_RANK=0): known spills, must not grow
    "k_compact<1>": (124, 8), "k_compact<2>": (60, 4),
}


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_hot_kernels_keep_their_register_and_scratch_budget():
    import resource_usage

    rows = {r["name"].split("(")[0]: r for r in resource_usage.collect()}
    missing = [k for k in BUDGET if k not in rows]
    assert not missing, (missing, sorted(rows)[:40])
    bad = []
    for name, (scratch, waves) in BUDGET.items():
        r = rows[name]
        if r["ScratchSize [bytes/lane]"] > scratch or r["Occupancy [waves/SIMD]"] < waves:
            bad.append((name, r["VGPRs"], r["VGPRs Spill"], r["ScratchSize [bytes/lane]"], r["Occupancy [waves/SIMD]"], "budget", scratch, waves))
    assert not bad, bad

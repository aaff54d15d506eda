"""CPU: the PRODUCT'S HOST LIBRARY — kmc_amd/csrc/kmc_hip.hip, every C-ABI entry point with its buffer management, launch sequences and error
handling — compiled with g++ over the emulated kernels and an emulated HIP runtime (tests/emu.py build_hostlib: the source is used as it is,
only the `<<<...>>>` launches are rewritten) and loaded through KMC_HIP_LIB in place of libkmc_hip.so. Two uses:
  * a quick subset of the -m gpu tests runs here, in a child process, on every CPU run of the suite (the whole -m gpu parity file passes this
    way too: `KMC_HIP_LIB=tests/hipemu/libkmc_hip_emu_small.so pytest tests -m gpu -k ...`, 142 tests in 25 min when this was written);
  * the product binary kmc_hip_s1 (all four plug-ins, both dlopen loaders) over that library: stage 1 through kmc_hip_split_part and stage 2
    through kmc_hip_process_bin_submit/_wait, i.e. everything but the silicon and the real runtime, against the reference's database.
Test infrastructure only: nothing loads the emulated library unless KMC_HIP_LIB names it."""
import hashlib
import os
import re
import subprocess
import sys

import pytest

import emu
from kmc_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def _exe(name):
    """oracle/_ref/<name> for the reference and the oracle builds (checkers), kmc_amd/bin/<name> for the product drop-in binaries"""
    return os.path.join(ROOT, "kmc_amd", "bin", name) if name.startswith("kmc_hip") else os.path.join(REF, name)

QUICK = ("(order_database and k27) or process_bin_edges or size_and_n_rec or sort_records_into or allreduce_stats_single or submit_wait or (test_stage1_kernels_match_the_oracle and 27-9) "
         "or (compact_stage_matches_oracle and 27-3) or (expand_stage_matches_oracle and 1-27) or (test_process_bin_matches_oracle and k27-cutoff)")


def test_gpu_tests_pass_on_the_emulated_host_library():
    lib = emu.build_hostlib("small")
    env = dict(os.environ, KMC_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_stage1.py"), "-m", "gpu",
                        "-q", "-x", "-p", "no:cacheprovider", "-k", QUICK], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail
    assert int(tail.split(" passed")[0].split()[-1]) >= 10, tail


def _run(exe, flags, inp, tmp_path, tag, env=None):
    t = tmp_path / ("tmp_" + tag)
    t.mkdir(exist_ok=True)
    db = str(tmp_path / ("db_" + tag))
    r = subprocess.run([_exe(exe), *flags, inp, db, str(t)], capture_output=True, text=True, env=dict(os.environ, **(env or {})), timeout=1500)
    assert r.returncode == 0, (exe, flags, (r.stdout + r.stderr)[-1500:])
    md5 = tuple(hashlib.md5(open(db + e, "rb").read()).hexdigest() for e in (".kmc_pre", ".kmc_suf"))
    stats = [ln.split(":")[1].strip() for ln in r.stdout.splitlines() if "No. of" in ln or "Total no." in ln]
    return md5, stats, r.stderr


@pytest.mark.parametrize("sorted_emit", [False, True], ids=["emit", "emit-through-a-sort"])
@pytest.mark.parametrize("flags", [["-k27", "-ci1"], ["-k55", "-b"]], ids=lambda f: "".join(f))
def test_product_binary_over_the_emulated_host_library_writes_the_reference_database(flags, sorted_emit, tmp_path):
    """sorted_emit: KMC_HIP_S1_SORTED_EMIT=1 — the alternative emit of stage 1, whose sort is the library's own radix path (k_hist, k_onesweep) on the
    super-k-mer keys"""
    if not os.path.exists(_exe("kmc_hip_s1")):
        pytest.skip("kmc_amd/bin/kmc_hip_s1 not built (needs /root/reference)")
    lib = emu.build_hostlib("small")
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=11, genome_len=30_000, n_reads=1_500, read_len=150)
    common = flags + ["-m2", "-sf1", "-n64"]
    want = _run("kmc", common + ["-sp1", "-sr1"], fq, tmp_path, "ref")
    env = {"KMC_HIP_LIB": lib, "KMC_HIP_VERBOSE": "1"}
    if sorted_emit:
        env["KMC_HIP_S1_SORTED_EMIT"] = "1"
    got = _run("kmc_hip_s1", common + ["-sp2", "-sr2"], fq, tmp_path, "emu", env=env)
    assert got[:2] == want[:2]
    assert "parts through the engine" in got[2] and "[kmc_hip stage 2] 64 bins, 2 workers" in got[2]


def test_worker_hands_waiting_bins_to_the_engine_together(tmp_path):
    """kb_sorter_plugin.h: a worker that finds more bins waiting takes up to KMC_HIP_WORKER_GROUP of them (CBinQueue::pop_if_any, never waiting) and
    hands them to kmc_hip_process_bins_submit/_wait in one call, where they share one sort; the database stays the reference's, with groups of 4
    (default), 16, and with grouping off."""
    if not os.path.exists(_exe("kmc_hip")):
        pytest.skip("kmc_amd/bin/kmc_hip not built (needs /root/reference)")
    lib = emu.build_hostlib("small")
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=11, genome_len=30_000, n_reads=1_500, read_len=150)
    common = ["-k27", "-ci1", "-m2", "-sf1", "-n64", "-sp1"]
    want = _run("kmc", common + ["-sr1"], fq, tmp_path, "ref")
    for grp, several in (("4", True), ("16", True), ("1", False)):
        got = _run("kmc_hip", common + ["-sr2"], fq, tmp_path, "g" + grp, env={"KMC_HIP_LIB": lib, "KMC_HIP_VERBOSE": "1", "KMC_HIP_WORKER_GROUP": grp})
        assert got[:2] == want[:2], grp
        calls = int(got[2].split(" engine calls with several bins")[0].split()[-1])
        assert (calls > 0) == several, (grp, calls)


def test_two_emulated_devices_context_wide_records_allreduce_and_dropin(tmp_path):
    """the multi-device code of the host library and of the worker loader on TWO emulated devices (HIPEMU_DEVICES=2) — no box this repo has seen
    had two GPUs, so this is where `KMC_HIP_DEVICES=0,1`, the per-device function attributes and kmc_hip_allreduce_stats over two devices run
    (the -m gpu original, test_two_devices_worker_allreduce_and_wide_records, passes on the emulated library as well: 230 s)"""
    lib = emu.build_hostlib("small")
    env = dict(os.environ, KMC_HIP_LIB=lib, HIPEMU_DEVICES="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostlib_two_devices.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "two devices ok" in r.stdout, (r.stdout + r.stderr)[-1500:]
    if not os.path.exists(_exe("kmc_hip")):
        return
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=5, genome_len=20_000, n_reads=800, read_len=150)
    want = _run("kmc", ["-k27", "-ci1", "-m2", "-sf1", "-n64", "-sp1", "-sr1"], fq, tmp_path, "ref")
    got = _run("kmc_hip", ["-k27", "-ci1", "-m2", "-sf1", "-n64", "-sp1", "-sr4"], fq, tmp_path, "emu", env={"KMC_HIP_LIB": lib, "HIPEMU_DEVICES": "2", "KMC_HIP_DEVICES": "0,1"})
    assert got[:2] == want[:2]


_HYBRID_CASE = r'''
import sys
import numpy as np
sys.path.insert(0, r"%(root)s"); sys.path.insert(0, r"%(root)s/tests")
import oracle_py as O
from kmc_amd import capi
from test_gpu_parity import _run_batch

ctx = capi.Context((0,))
for k, pl, kw in ((27, 3, {}), (55, 3, {"cutoff_min": 1}), (27, 0, {"output_type": 1})):
    bins = capi.synth_bins(seed=7, genome_len=6000, n_reads=1200, k=k, n_bins=5, n_threads=1)
    p = capi.make_params(k, lut_prefix_len=pl, **kw)
    op = O.make_params(p.kmer_len, p.both_strands, p.cutoff_min, p.cutoff_max, p.counter_max, p.lut_prefix_len, p.output_type, p.without_output)
    got, err = _run_batch(ctx, p, bins, 1)
    assert err is None, err
    for i, (img, nrec, packs, _) in enumerate(bins):
        w = O.process_bin(op, img, nrec)
        assert np.array_equal(got[i][0], w[0]) and np.array_equal(got[i][1], w[1]) and np.array_equal(got[i][2], w[2]), (k, i)
t = ctx.local_sort_totals()
assert t["hybrid_groups"] >= 3 and t["redo_groups"] == 0, t
# two-word records with one k-mer more often than k_giant_tiles takes (GT_MAX_RECORDS: 2048 in this build): the bins must come back through the LSD passes
bins = capi.synth_bins(seed=5, genome_len=190, n_reads=2600, k=55, n_bins=2, err=0.0, n_threads=1)
p = capi.make_params(55)
op = O.make_params(55)
got, err = _run_batch(ctx, p, bins, 1)
assert err is None, err
for i, (img, nrec, packs, _) in enumerate(bins):
    w = O.process_bin(op, img, nrec)
    assert np.array_equal(got[i][0], w[0]) and np.array_equal(got[i][1], w[1]) and np.array_equal(got[i][2], w[2]), i
assert ctx.local_sort_totals()["redo_groups"] >= 1
# the host boundary with several bins per call (kmc_hip_process_bins_submit/_wait): sorted together, results per bin; an empty bin among them
for k, pl, kw, nb in ((27, 3, {}, 7), (55, 3, {"cutoff_min": 1}, 5), (27, 0, {"output_type": 1}, 4), (32, 4, {}, 3), (25, 1, {}, 15)):
    bins = capi.synth_bins(seed=7, genome_len=6000, n_reads=1500, k=k, n_bins=nb, n_threads=1)
    p = capi.make_params(k, lut_prefix_len=pl, **kw)
    op = O.make_params(p.kmer_len, p.both_strands, p.cutoff_min, p.cutoff_max, p.counter_max, p.lut_prefix_len, p.output_type, p.without_output)
    hb = [(img, nrec, packs if i & 1 else None) for i, (img, nrec, packs, _) in enumerate(bins)]
    hb.insert(2, (np.zeros(0, np.uint8), 0, None))
    got = ctx.process_bins_host(p, hb, slot=3)
    for i, (img, nrec, _) in enumerate(hb):
        w = O.process_bin(op, img, nrec) if nrec else (np.zeros(0, np.uint8), np.zeros(ctx.lut_entries(p), np.uint64), np.zeros(4, np.uint64))
        assert np.array_equal(got[i][0], w[0]) and np.array_equal(got[i][1], w[1]) and np.array_equal(got[i][2], w[2]), (k, i)
bins = capi.synth_bins(seed=5, genome_len=190, n_reads=2600, k=55, n_bins=3, err=0.0, n_threads=1)  # ... and bins that come back for LSD passes
before = ctx.local_sort_totals()["redo_groups"]
got = ctx.process_bins_host(capi.make_params(55), [(b[0], b[1], b[2]) for b in bins])
for i, b in enumerate(bins):
    w = O.process_bin(O.make_params(55), b[0], b[1])
    assert np.array_equal(got[i][0], w[0]) and np.array_equal(got[i][1], w[1]) and np.array_equal(got[i][2], w[2]), i
assert ctx.local_sort_totals()["redo_groups"] > before
rng = np.random.default_rng(3)
a = rng.integers(0, 2**54, size=(5000, 1), dtype=np.uint64)
assert np.array_equal(ctx.sort_records(a, 7), O.sort(a))
print("HYBRID-OK")
'''


_RANK_CASE = r'''
import sys
import numpy as np
sys.path.insert(0, r"%(root)s"); sys.path.insert(0, r"%(root)s/tests")
import oracle_py as O
from kmc_amd import capi
from test_gpu_parity import _run_batch

ctx = capi.Context((0,))  # default mode: every group takes k_bucket_rank — tiles ranked and counted inside LDS
def check(k, bins, **kw):
    p = capi.make_params(k, **kw)
    op = O.make_params(p.kmer_len, p.both_strands, p.cutoff_min, p.cutoff_max, p.counter_max, p.lut_prefix_len, p.output_type, p.without_output)
    t0, c0 = ctx.local_sort_totals(), ctx.path_counters()
    got, err = _run_batch(ctx, p, bins, 1)
    assert err is None, err
    for i, (img, nrec, packs, _) in enumerate(bins):
        w = O.process_bin(op, img, nrec)
        assert np.array_equal(got[i][0], w[0]) and np.array_equal(got[i][1], w[1]) and np.array_equal(got[i][2], w[2]), (k, kw, i)
    t1, c1 = ctx.local_sort_totals(), ctx.path_counters()
    return t1["hybrid_groups"] - t0["hybrid_groups"], t1["redo_groups"] - t0["redo_groups"], {x: c1[x] - c0[x] for x in c1}
# one-word records. 24 key bits left below the passes: (rem, index) pairs of 32 bits; k = 32: 32 bits left, pairs of 64; KFF records; 16 bins per group (k = 25);
# forward strand only; without output; counters that clamp and a cutoff_max
for k, nb, kw in ((27, 5, dict(lut_prefix_len=3)), (32, 3, dict(lut_prefix_len=4)), (27, 4, dict(lut_prefix_len=0, output_type=1)), (25, 9, dict(lut_prefix_len=1, cutoff_min=1)),
                  (21, 4, dict(lut_prefix_len=1, both_strands=0)), (27, 4, dict(lut_prefix_len=3, without_output=1)), (27, 3, dict(lut_prefix_len=3, cutoff_max=20, counter_max=7))):
    h, r, c = check(k, capi.synth_bins(seed=7, genome_len=2500, n_reads=300, k=k, n_bins=nb, n_threads=1), **kw)
    assert h >= 1 and r == 0 and c["rank_count"] >= 1 and c["rank_compact"] == c["bucket_count"] == 0, (k, h, r, c)
import os
ARENA = os.environ.get("KMC_HIP_ARENA", "1") != "0"  # round 6 (one-word records): buckets beyond BR_MID records through arena_sort.hip.h; 0: round 5's finisher
# records that may outgrow a tile's span (k = 32 without a LUT prefix: 8 suffix bytes + a 4-byte counter): ranked in place, then k_compact
h, r, c = check(32, capi.synth_bins(seed=7, genome_len=2500, n_reads=300, k=32, n_bins=3, n_threads=1), lut_prefix_len=0, cutoff_max=100000, counter_max=70000)
assert c["rank_compact"] >= 1 and c["rank_count"] == 0, c
# wider records: two words (A/B pairs, rem <= 80 bits; k = 64: six HBM passes instead of sixteen), KFF, three and more words (whole records compared)
for k, nb, kw in (((55, 4, dict(lut_prefix_len=3)), (127, 4, dict(lut_prefix_len=3))) if not ARENA else
                  ((55, 4, dict(lut_prefix_len=3)), (40, 3, dict(lut_prefix_len=4, cutoff_min=1)), (64, 1, dict(lut_prefix_len=4)), (55, 4, dict(lut_prefix_len=0, output_type=1)),
                   (127, 4, dict(lut_prefix_len=3)), (70, 3, dict(lut_prefix_len=2, both_strands=0)), (200, 2, dict(lut_prefix_len=4)))):
    h, r, c = check(k, capi.synth_bins(seed=7, genome_len=2500, n_reads=250, k=k, n_bins=nb, n_threads=1, read_len=max(150, k + 40)), **kw)
    assert h >= 1 and r == 0 and c["rank_count"] >= 1 and c["bucket_count"] == 0, (k, h, r, c)
    # two words and more with FOUR HBM passes (k = 40, 200: five — bits above the key in the top byte; k = 64: six): the passes move (key top, record number) pairs of
    # 8 bytes, the records stay where k_expand wrote them and k_bucket_rank gathers them by number
    assert (c["indirect"] >= 1) == (k in (55, 127, 70)), (k, c)
# every k-mer a few hundred times: buckets longer than the room at the end of a window, tiles longer than the capacity -> taken in chunks (one-word records with the arena:
# buckets beyond BR_MID — 192 in this build — sorted there and written back), nothing comes back
for k, glen, err in ((27, 2000, 0.0), (27, 600, 0.002), (55, 1500, 0.0)):
    h, r, c = check(k, capi.synth_bins(seed=3, genome_len=glen, n_reads=700, k=k, n_bins=4, err=err, n_threads=1), lut_prefix_len=3)
    assert h >= 1 and r == 0, (k, glen, h, r)
# one k-mer more often than a tile holds records (also with read errors around it; two- and three-word records): k_giant_tiles / the arena's giant segments, nothing comes back
for k, kw, err in (((27, dict(lut_prefix_len=3, cutoff_min=1), 0.01),) if not ARENA else
                   ((27, dict(lut_prefix_len=3, cutoff_min=1), 0.01), (55, dict(lut_prefix_len=3), 0.0), (70, dict(lut_prefix_len=0, output_type=1), 0.0))):
    h, r, c = check(k, capi.synth_bins(seed=5, genome_len=300, n_reads=1500, k=k, n_bins=2, err=err, n_threads=1), **kw)
    assert h >= 1 and r == 0 and c["giant_tiles"] >= 1, (k, h, r, c)
# ... and more often than k_giant_tiles takes (GT_MAX_RECORDS: 2048 in this build): round 5's finisher sends the bins back for LSD passes; the arena takes a bucket of any length
h, r, c = check(27, capi.synth_bins(seed=5, genome_len=160, n_reads=2600, k=27, n_bins=2, err=0.0, n_threads=1), lut_prefix_len=3)
assert (r >= 1) == (not ARENA) and (not ARENA or c["giant_records"] > 100000), (h, r, c)
if ARENA:
    # repeat families, a satellite and a poly-A run in one group: buckets of every size class at once; KFF; clamped counters and a cutoff_max inside the giant runs; k = 32
    os.environ["KMC_SYNTH_REPEATS"] = "300:40:10,60:400:0,H500"
    b = capi.synth_bins(seed=9, genome_len=20000, n_reads=3000, k=27, n_bins=3, n_threads=1)
    del os.environ["KMC_SYNTH_REPEATS"]
    for kw in (dict(lut_prefix_len=3), dict(lut_prefix_len=0, output_type=1), dict(lut_prefix_len=3, cutoff_max=50, counter_max=7)):
        h, r, c = check(27, b, **kw)
        assert r == 0 and c["giant_tiles"] >= 1, (kw, h, r, c)
    h, r, c = check(32, capi.synth_bins(seed=5, genome_len=300, n_reads=1500, k=32, n_bins=2, err=0.01, n_threads=1), lut_prefix_len=4)
    assert r == 0 and c["giant_tiles"] >= 1, (h, r, c)
print("RANK-OK")
'''


@pytest.mark.parametrize("arena", ["1", "0"], ids=["arena", "round-5-finisher"])
def test_rank_path_on_the_emulated_host_library(arena):
    """k_bucket_rank (bucket_sort.hip.h) and, for one-word records, the arena of the repeat-rich buckets (arena_sort.hip.h: k_bucket_detect, k_arena_plan / _gather, k_onesweep_dyn,
    k_arena_finish; KMC_HIP_ARENA=0: round 5's finisher — big buckets walked by the whole workgroup, k_giant_tiles, bins with an enormous bucket sent back) on the CPU, as a default
    run takes them: groups of k-mers of every record width through the HBM passes, k_bucket_bounds, the pairwise ranking of every tile (32- and 64-bit pairs, A/B pairs of two-word
    records, whole records beyond) and the counting of the ranked tile inside LDS (fused), chunked tiles, the in-place variant + k_compact — per bin against the oracle."""
    lib = emu.build_hostlib("small")
    r = subprocess.run([sys.executable, "-c", _RANK_CASE % {"root": ROOT}], env=dict(os.environ, KMC_HIP_LIB=lib, KMC_HIP_ARENA=arena),
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and "RANK-OK" in r.stdout, (r.stdout + r.stderr)[-1500:]


def test_hybrid_sort_on_the_emulated_host_library():
    """bucket_sort.hip.h on the CPU, as a default run takes it: k_bucket_bounds + k_bucket_rank (groups of bins, k = 27 / 55, KFF records), a sort-only call (LSD passes)
    against the oracle, and the redo of bins with a bucket nothing on the device takes — through the product's host library over the emulated runtime; and the host
    boundary with several bins per call (kmc_hip_process_bins_submit/_wait), hybrid and redo included."""
    lib = emu.build_hostlib("small")
    r = subprocess.run([sys.executable, "-c", _HYBRID_CASE % {"root": ROOT}], env=dict(os.environ, KMC_HIP_LIB=lib), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and "HYBRID-OK" in r.stdout, (r.stdout + r.stderr)[-1500:]


def test_dropin_host_side_switches_of_round_6_keep_the_database(tmp_path):
    """The drop-in (kmc_amd/bin/kmc_hip, RAM-only mode) over the emulated host library under the host-side switches of round 6 — a pinned pool too small for the bins in flight
    (readers WAIT for a buffer: host_pool.h get_wait; with no wait they spill into the arena as in round 5; no pool at all), stream-slot slabs too small for a bin's buffers
    (kmc_hip_reserve_slot: carved while there is room, allocated on demand after), the allocator tunables by re-exec (both variants), 2 and 16 completer writers — every
    one must write the reference's `-sr1` bytes (the reorder buffer of kmc_order.h is on every path)."""
    if not (os.path.exists(_exe("kmc_hip")) and os.path.exists(_exe("kmc"))):
        pytest.skip("kmc_amd/bin/kmc_hip or oracle/_ref/kmc not built (needs /root/reference)")
    lib = emu.build_hostlib("small")
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=5, genome_len=40_000, n_reads=3000, read_len=150)
    flags = ["-k27", "-ci1", "-m2", "-sf1", "-n64", "-sp1", "-r"]
    want = _run("kmc", flags + ["-sr1"], fq, tmp_path, "ref")
    for tag, env in (("defaults", {}),
                     ("pool-wait", {"KMC_HIP_PINNED_POOL_MB": "1", "KMC_HIP_POOL_WAIT_MS": "30"}),
                     ("pool-spill", {"KMC_HIP_PINNED_POOL_MB": "1", "KMC_HIP_POOL_WAIT_MS": "0"}),
                     ("no-pool", {"KMC_HIP_PINNED_POOL_MB": "0"}),
                     ("slab-1MB", {"KMC_HIP_SLOT_SLAB_MB": "1"}),
                     ("tunables-1", {"KMC_HIP_TUNE_MALLOC": "1"}),
                     ("tunables-2", {"KMC_HIP_TUNE_MALLOC": "2", "KMC_HIP_WRITERS": "2"}),
                     ("writers-16", {"KMC_HIP_WRITERS": "16", "KMC_HIP_READERS": "3"})):
        got = _run("kmc_hip", flags + ["-sr4"], fq, tmp_path, tag, env=dict({"KMC_HIP_LIB": lib, "KMC_HIP_VERBOSE": "1"}, **env))
        assert got[:2] == want[:2], tag
        if tag == "slab-1MB":
            assert "stream slots have a slab of 1 MB" in got[2], got[2][-600:]

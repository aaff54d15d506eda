"""GPU suite (-m gpu): the HIP path, through the C-ABI (ctypes -> libkmc_hip.so), against the oracle and the
committed reference golden vectors. Bit-exact everywhere: this is integer/byte work."""
import ctypes as C
import glob
import hashlib
import os
import subprocess
import zlib

import numpy as np
import pytest

import binsynth
import golden_io
import oracle_py as O
from kmc_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.bins")))


def _first_diff(a, b):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    if a.size != b.size:
        return f"size {a.size} != {b.size}"
    d = np.flatnonzero(a != b)
    return "equal" if d.size == 0 else f"{d.size} diffs, first at {int(d[0])}: {a[d[0]]} != {b[d[0]]}"


def hp(k, **kw):
    return capi.make_params(k, **kw)


def op(p):
    return O.make_params(p.kmer_len, p.both_strands, p.cutoff_min, p.cutoff_max, p.counter_max, p.lut_prefix_len, p.output_type,
                         p.without_output)


# ------------------------------------------------------------------------------------------------ narrow boundary
@pytest.mark.parametrize("words,key_bytes", [(1, 7), (1, 8), (1, 1), (2, 14), (2, 16), (3, 20), (4, 32), (8, 64)])
@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 4095, 4096, 4097, 100_003])
def test_sort_records_matches_oracle(ctx, words, key_bytes, n):
    rng = np.random.default_rng(1000 * words + key_bytes + n)
    recs = rng.integers(0, 2**63, size=(n, words), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, words), dtype=np.uint64)
    # zero everything above key_bytes (contract of SortFunction: higher bytes are zero, kb_sorter.h:761-775)
    for w in range(words):
        lo = 8 * w
        if key_bytes <= lo:
            recs[:, w] = 0
        elif key_bytes - lo < 8:
            recs[:, w] &= np.uint64((1 << (8 * (key_bytes - lo))) - 1)
    got = ctx.sort_records(recs, key_bytes)
    want = O.sort(recs) if n else recs
    assert np.array_equal(got, want), _first_diff(got, want)


@pytest.mark.parametrize("kind", ["all_equal", "two_values", "low_entropy", "sorted", "reversed", "one_hot_byte"])
def test_sort_records_skewed_inputs(ctx, kind):
    """digit skew is what low-complexity bins (poly-A) look like: one bucket takes everything."""
    rng = np.random.default_rng(42)
    n = 300_001
    if kind == "all_equal":
        a = np.full(n, 0x00123456789ABCDE, dtype=np.uint64)
    elif kind == "two_values":
        a = np.where(rng.random(n) < 0.5, np.uint64(5), np.uint64(0x0011223344556677))
    elif kind == "low_entropy":
        a = rng.integers(0, 7, size=n, dtype=np.uint64) << np.uint64(40)
    elif kind == "sorted":
        a = np.sort(rng.integers(0, 2**54, size=n, dtype=np.uint64))
    elif kind == "reversed":
        a = np.sort(rng.integers(0, 2**54, size=n, dtype=np.uint64))[::-1].copy()
    else:
        a = rng.integers(0, 256, size=n, dtype=np.uint64) << np.uint64(8 * int(rng.integers(0, 7)))
    got = ctx.sort_records(a.reshape(-1, 1), 7)
    assert np.array_equal(got[:, 0], np.sort(a)), _first_diff(got[:, 0], np.sort(a))


def test_sort_records_large_is_a_sorted_permutation(ctx):
    """size-independent properties at a size the oracle would not finish quickly: sortedness + multiset checksum."""
    rng = np.random.default_rng(9)
    n = 40_000_000
    a = rng.integers(0, 2**54, size=n, dtype=np.uint64)
    got = ctx.sort_records(a.reshape(-1, 1), 7)[:, 0]
    assert np.all(got[:-1] <= got[1:])
    assert int(np.bitwise_xor.reduce(a)) == int(np.bitwise_xor.reduce(got))
    assert int(a.sum(dtype=np.uint64)) == int(got.sum(dtype=np.uint64))
    sub = rng.integers(0, n, size=1000)
    srt = np.sort(a)
    assert np.array_equal(got[sub], srt[sub])


# ------------------------------------------------------------------------------------------------ stage isolation
@pytest.mark.parametrize("k", [14, 27, 28, 32, 33, 55, 64, 96, 127, 200, 256])
@pytest.mark.parametrize("both", [1, 0])
def test_expand_stage_matches_oracle(ctx, k, both):
    rng = np.random.default_rng(k * 2 + both)
    img, nk, packs = binsynth.random_bin(rng, k, 3000, max_extra=120, pack_size=257)
    p = hp(k, both_strands=both, lut_prefix_len=0, output_type=1)
    got = ctx.debug_expand(p, img, nk, packs)
    want = O.expand(op(p), img)
    assert np.array_equal(got, want), _first_diff(got, want)


@pytest.mark.parametrize("k,max_extra,pack_size,n_super", [
    (27, 255, 1, 600),        # one super-k-mer per pack
    (27, 40, 10**9, 30000),   # a single pack far beyond 4096 super-k-mers / many parse chunks
    (27, 255, 37, 5000),      # long records straddling parse-chunk and expand-slice boundaries
    (14, 255, 4096, 9000),    # smallest bin-path k: shortest records (5 bytes), most starts per slice
    (256, 255, 500, 3000),    # longest records (up to 129 bytes)
    (31, 0, 4096, 20000),     # e = 0 everywhere: one k-mer per super-k-mer
])
def test_parse_and_expand_pack_shapes(ctx, k, max_extra, pack_size, n_super):
    """the parse kernel resolves the record chain speculatively per pack and per 4 KiB chunk; the expand kernel works on
    8 KiB slices: exercise every boundary case against the oracle's sequential walk"""
    rng = np.random.default_rng(k + max_extra + n_super)
    img, nk, packs = binsynth.random_bin(rng, k, n_super, max_extra=max_extra, pack_size=pack_size)
    p = hp(k, lut_prefix_len=0, output_type=1)
    want = O.expand(op(p), img)
    got = ctx.debug_expand(p, img, nk, packs)
    assert np.array_equal(got, want), _first_diff(got, want)
    # and with the library finding pack boundaries itself
    out, lut, st = ctx.process_bin(hp(k, cutoff_min=1, lut_prefix_len=0, output_type=1), img, nk, None)
    w = O.process_bin(op(hp(k, cutoff_min=1, lut_prefix_len=0, output_type=1)), img, nk)
    assert np.array_equal(out, w[0]) and np.array_equal(st, w[2])


@pytest.mark.parametrize("k,pl", [(27, 3), (27, 7), (55, 3), (127, 3), (32, 4), (64, 0)])
def test_compact_stage_matches_oracle(ctx, k, pl):
    rng = np.random.default_rng(k + pl)
    g = rng.integers(0, 4, size=20_000, dtype=np.uint8)
    img, nk, _ = binsynth.random_bin(rng, k, 4000, max_extra=60, genome=g)
    for kw in (dict(cutoff_min=2), dict(cutoff_min=1, counter_max=3), dict(cutoff_min=1, cutoff_max=4), dict(cutoff_min=3, counter_max=70000)):
        p = hp(k, lut_prefix_len=pl, output_type=0 if pl else 1, **kw)
        srt = O.sort(O.expand(op(p), img))
        out, lut, st = ctx.debug_compact(p, srt)
        w_out, w_lut, w_st = O.process_bin(op(p), img, nk)
        assert np.array_equal(st, w_st), (kw, st, w_st)
        assert np.array_equal(out, w_out), (kw, _first_diff(out, w_out))
        assert np.array_equal(lut, w_lut), (kw, _first_diff(lut, w_lut))


def test_compact_long_runs_cross_many_tiles(ctx):
    """runs far longer than a compaction tile (poly-A style): exercises the wave-cooperative run-start search."""
    reps = [1, 5000, 2, 70_000, 1, 1, 300_000, 3, 2049, 2048, 2047, 1]
    vals = np.sort(np.random.default_rng(5).integers(0, 2**54, size=len(reps), dtype=np.uint64))
    srt = np.repeat(vals, reps).reshape(-1, 1)
    for kw in (dict(cutoff_min=2), dict(cutoff_min=1, counter_max=255), dict(cutoff_min=1, cutoff_max=2048, counter_max=10**6)):
        p = hp(27, lut_prefix_len=3, **kw)
        out, lut, st = ctx.debug_compact(p, srt)
        obytes, olut, ostats = (np.zeros(srt.shape[0] * 8 + 8, dtype=np.uint8), np.zeros(64, dtype=np.uint64), np.zeros(4, dtype=np.uint64))
        ob = C.c_uint64()
        rc = O.lib().oracle_compact(C.byref(op(p)), srt.ctypes.data_as(C.POINTER(C.c_uint64)), srt.shape[0],
                                    obytes.ctypes.data_as(C.POINTER(C.c_uint8)), obytes.size, C.byref(ob),
                                    olut.ctypes.data_as(C.POINTER(C.c_uint64)), ostats.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0
        assert np.array_equal(st, ostats), (kw, st, ostats)
        assert np.array_equal(out, obytes[: ob.value]), (kw, _first_diff(out, obytes[: ob.value]))
        assert np.array_equal(lut, olut)


# ------------------------------------------------------------------------------------------------ full boundary
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_process_bin_reproduces_reference_golden_bins(ctx, path):
    """Real stage-1 bins of the reference with the reference's own stage-2 outputs (tests/golden/make_golden.py)."""
    for b in golden_io.read_bins(path):
        k, both, cmin, wo, cmax, cs, pl, ot = b["params"]
        p = hp(k, both_strands=both, cutoff_min=cmin, cutoff_max=cmax, counter_max=cs, lut_prefix_len=pl, output_type=ot, without_output=wo)
        for packs in (b["pack_bytes"], None):  # with the caller's expander packs, and letting the library find them
            out, lut, st = ctx.process_bin(p, b["image"], b["n_rec"], packs)
            assert np.array_equal(st, b["stats"]), (st, b["stats"])
            assert np.array_equal(out, b["out"]), _first_diff(out, b["out"])
            assert np.array_equal(lut, b["lut"]), _first_diff(lut, b["lut"])


CASES = [
    dict(k=27), dict(k=27, both_strands=0), dict(k=27, cutoff_min=1, counter_max=1), dict(k=27, cutoff_min=1, cutoff_max=3),
    dict(k=27, lut_prefix_len=7), dict(k=28, lut_prefix_len=4), dict(k=31, lut_prefix_len=3), dict(k=32, lut_prefix_len=4),
    dict(k=33, lut_prefix_len=5), dict(k=55), dict(k=55, counter_max=100000, cutoff_min=1), dict(k=64, lut_prefix_len=4),
    dict(k=96, lut_prefix_len=4), dict(k=127), dict(k=127, both_strands=0), dict(k=160, lut_prefix_len=4), dict(k=200, lut_prefix_len=4),
    dict(k=256, lut_prefix_len=4), dict(k=27, lut_prefix_len=0, output_type=1), dict(k=55, lut_prefix_len=0, output_type=1, counter_max=1000),
    dict(k=27, without_output=1), dict(k=14, lut_prefix_len=2, cutoff_min=1),
]


@pytest.mark.parametrize("kw", CASES, ids=lambda d: "-".join(f"{a}{b}" for a, b in d.items()))
def test_process_bin_matches_oracle(ctx, kw):
    rng = np.random.default_rng(zlib.crc32(repr(sorted(kw.items())).encode()))
    k = kw["k"]
    g = rng.integers(0, 4, size=30_000, dtype=np.uint8)
    img, nk, packs = binsynth.random_bin(rng, k, 6000, max_extra=100, genome=g, pack_size=1000)
    p = hp(**kw)
    out, lut, st = ctx.process_bin(p, img, nk, packs)
    w_out, w_lut, w_st = O.process_bin(op(p), img, nk)
    assert np.array_equal(st, w_st), (st, w_st)
    assert np.array_equal(out, w_out), _first_diff(out, w_out)
    if p.lut_prefix_len and not p.output_type and not p.without_output:
        assert np.array_equal(lut, w_lut), _first_diff(lut, w_lut)


def test_process_bin_edges(ctx):
    p = hp(27)
    out, lut, st = ctx.process_bin(p, np.zeros(0, dtype=np.uint8), 0)  # empty bin: still a valid call (kb_reader.h:198-205)
    assert out.size == 0 and int(lut.sum()) == 0 and st.tolist() == [0, 0, 0, 0]
    rng = np.random.default_rng(3)
    img, nk, packs = binsynth.random_bin(rng, 27, 1, max_extra=0)  # a single k-mer
    out, lut, st = ctx.process_bin(hp(27, cutoff_min=1), img, nk, packs)
    w = O.process_bin(op(hp(27, cutoff_min=1)), img, nk)
    assert np.array_equal(out, w[0]) and np.array_equal(st, w[2])
    img, nk, packs = binsynth.random_bin(rng, 27, 50, max_extra=255)  # maximum-length super-k-mers (e = 255, splitter.cpp:656)
    out, lut, st = ctx.process_bin(hp(27, cutoff_min=1), img, nk, packs)
    w = O.process_bin(op(hp(27, cutoff_min=1)), img, nk)
    assert np.array_equal(out, w[0]) and np.array_equal(lut, w[1]) and np.array_equal(st, w[2])
    # error behaviour: ragged stream / wrong n_rec / too-small output are reported, never silently wrong
    with pytest.raises(capi.KmcHipError) as e:
        ctx.process_bin(hp(27), img[:-3], nk, None)
    assert e.value.code == -4
    with pytest.raises(capi.KmcHipError) as e:
        ctx.process_bin(hp(27), img, nk + 1, packs)
    assert e.value.code == -4
    with pytest.raises(capi.KmcHipError) as e:
        ctx.process_bin(hp(27, cutoff_min=1), img, nk, packs, out_capacity=7 * 3)
    assert e.value.code == -5
    with pytest.raises(capi.KmcHipError) as e:
        ctx.process_bin(hp(27, lut_prefix_len=2), img, nk, packs)  # (k-p) % 4 != 0 is not a KMC configuration
    assert e.value.code == -1


def test_process_bin_medium_synthetic_bin_matches_oracle(ctx):
    """a realistic bin: ~12 M k-mers of 30x-coverage reads cut by minimizers (kmc_amd/csrc/synth_bins.cpp)."""
    (img, nrec, packs, _), = capi.synth_bins(seed=11, genome_len=400_000, n_reads=100_000, k=27, n_bins=1)
    p = hp(27)
    out, lut, st = ctx.process_bin(p, img, nrec, packs)
    w_out, w_lut, w_st = O.process_bin(op(p), img, nrec)
    assert np.array_equal(st, w_st), (st, w_st)
    assert np.array_equal(out, w_out), _first_diff(out, w_out)
    assert np.array_equal(lut, w_lut)


def test_process_bin_large_properties(ctx):
    """~250 M k-mers (k=27): properties that need no oracle — tally identities, LUT total, output ordering."""
    (img, nrec, packs, _), = capi.synth_bins(seed=12, genome_len=8_000_000, n_reads=2_000_000, k=27, n_bins=1)
    p = hp(27)
    out, lut, st = ctx.process_bin(p, img, nrec, packs)
    rec = ctx.out_rec_bytes(p)
    counted = out.size // rec
    assert out.size % rec == 0 and int(st[3]) == nrec
    assert int(st[0] - st[1] - st[2]) == counted == int(lut.sum())
    r = out.reshape(-1, rec)
    assert r[:, -1].min() >= 2  # cutoff_min respected, counter byte
    # inside each LUT prefix group suffixes ascend strictly (the DB's binary-search invariant, kmc_api/kmc_file.cpp)
    suf = np.zeros(counted, dtype=np.uint64)
    for j in range(rec - 1):
        suf = (suf << np.uint64(8)) | r[:, j].astype(np.uint64)
    bounds = np.concatenate([[0], np.cumsum(lut).astype(np.int64)])
    inc = suf[1:] > suf[:-1]
    is_boundary = np.zeros(counted - 1, dtype=bool)
    b = bounds[1:-1]
    is_boundary[b[(b > 0) & (b < counted)] - 1] = True
    assert np.all(inc | is_boundary)
    # idempotence of the device path
    out2, lut2, st2 = ctx.process_bin(p, img, nrec, None)
    assert np.array_equal(out, out2) and np.array_equal(lut, lut2) and np.array_equal(st, st2)


def test_submit_wait_double_buffering(ctx):
    rng = np.random.default_rng(8)
    L = ctx.L
    bins = [binsynth.random_bin(rng, 27, 2000, genome=rng.integers(0, 4, size=9000, dtype=np.uint8)) for _ in range(4)]
    p = hp(27)
    outs = []
    bufs = []
    for i, (img, nk, packs) in enumerate(bins):
        slot = i & 1
        if i >= 2:
            outs.append(_wait(ctx, slot, bufs[i - 2]))
        cap = (nk + 1) // 2 * 7
        out = np.zeros(cap, dtype=np.uint8)
        lut = np.zeros(64, dtype=np.uint64)
        bufs.append((out, lut, img, packs))
        rc = L.kmc_hip_process_bin_submit(ctx.h, 0, slot, C.byref(p), img.ctypes.data_as(C.c_void_p), img.size, nk,
                                          packs.ctypes.data_as(C.c_void_p), packs.size, out.ctypes.data_as(C.c_void_p), cap,
                                          lut.ctypes.data_as(C.c_void_p))
        assert rc == 0, L.kmc_hip_last_error(ctx.h)
    outs.append(_wait(ctx, 0, bufs[2]))
    outs.append(_wait(ctx, 1, bufs[3]))
    for (img, nk, packs), (o, l, s) in zip(bins, outs):
        w = O.process_bin(op(p), img, nk)
        assert np.array_equal(o, w[0]) and np.array_equal(l, w[1]) and np.array_equal(s, w[2])


def _wait(ctx, slot, buf):
    ob = C.c_uint64()
    st = np.zeros(4, dtype=np.uint64)
    rc = ctx.L.kmc_hip_process_bin_wait(ctx.h, 0, slot, C.byref(ob), st.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 0, ctx.L.kmc_hip_last_error(ctx.h)
    return buf[0][: ob.value].copy(), buf[1].copy(), st


def test_allreduce_stats_single_device(ctx):
    a = np.array([[1, 2, 3, 2**40]], dtype=np.uint64)
    assert np.array_equal(ctx.allreduce_stats(a), a)


# ------------------------------------------------------------------------------------------------ the drop-in, end to end
def _md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


@pytest.mark.parametrize("flags", [["-k27"], ["-k55"], ["-k127"], ["-k27", "-b", "-ci1", "-cs3"]], ids=lambda f: "".join(f))
def test_kmc_with_hip_sorter_writes_the_reference_database(flags, ref_bins, tmp_path):
    """kmc_amd/bin/kmc_hip = the reference's own pipeline (stage 1, bin reader, completer, CLI) with
    CWKmerBinSorter swapped for the HIP worker (kb_sorter_plugin.h + libkmc_hip.so). Its .kmc_pre/.kmc_suf must be
    byte-identical to the unmodified reference run with -sr1."""
    if ref_bins is None:
        pytest.skip("oracle/_ref binaries were not shipped")
    from kmc_amd import synth

    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, **synth.CONFIGS["C1"])
    # hip4: four workers sharing two stream slots of device 0 twice over (KMC_HIP_DEVICES=0,0 exercises the
    # multi-device mapping on a 1-GPU box); ordered emission keeps the bytes equal to the reference's -sr1 run
    runs = (("kmc", "ref", ["-sr1"], {}), ("kmc_hip", "hip", ["-sr1"], {}), ("kmc_hip", "hip4", ["-t8", "-sr4"], {"KMC_HIP_DEVICES": "0,0"}))
    if flags == ["-k27"]:  # the 4-GPU worker layout on one GPU: eight workers over four logical devices (what `kmc_hip -sr8` does on a 4-GPU node, queues.h:2045-2146)
        runs += (("kmc_hip", "hip8", ["-t16", "-sr8"], {"KMC_HIP_DEVICES": "0,0,0,0", "KMC_HIP_VERBOSE": "1"}),)
    for exe, out, mode, extra in runs:
        env = dict(os.environ, KMC_HIP_LIB=capi.lib_path(), **extra)
        tmp = tmp_path / ("tmp_" + out)
        tmp.mkdir()
        r = subprocess.run([ref_bins[exe], *flags, *mode, fq, str(tmp_path / out), str(tmp)], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        if out == "hip8":
            assert "8 workers" in r.stderr, r.stderr[-1500:]
    for ext in (".kmc_pre", ".kmc_suf"):
        assert _md5(str(tmp_path / ("ref" + ext))) == _md5(str(tmp_path / ("hip" + ext))), ext
        assert _md5(str(tmp_path / ("ref" + ext))) == _md5(str(tmp_path / ("hip4" + ext))), ext + " (4 workers, 2 devices)"
        if flags == ["-k27"]:
            assert _md5(str(tmp_path / ("ref" + ext))) == _md5(str(tmp_path / ("hip8" + ext))), ext + " (8 workers, 4 devices)"


def test_sort_and_bin_cross_portion_boundaries(monkeypatch):
    """A bin of more than 2^29 k-mers is scattered in portions (one launch each, digit bases carried from portion to portion,
    kmc_hip.hip sort_device_t). With the portion shrunk to 2^15 records ($KMC_HIP_DEBUG_PORTION_LOG2, read at init) a
    small sort takes the same path and can be checked against the oracle bit for bit."""
    monkeypatch.setenv("KMC_HIP_DEBUG_PORTION_LOG2", "15")
    c2 = capi.Context((0,))
    try:
        rng = np.random.default_rng(2029)
        for words, key_bytes, n in ((1, 7, 100_003), (1, 8, 32_768 * 3), (2, 14, 70_001), (4, 32, 40_000)):
            recs = rng.integers(0, 2**62, size=(n, words), dtype=np.uint64)
            for w in range(words):
                lo = 8 * w
                if key_bytes - lo < 8:
                    recs[:, w] &= np.uint64((1 << (8 * max(key_bytes - lo, 0))) - 1)
            got = c2.sort_records(recs, key_bytes)
            assert np.array_equal(got, O.sort(recs)), (words, key_bytes, n)
        for k in (27, 55):
            img, nk, packs = binsynth.random_bin(rng, k, 3000, max_extra=60, genome=rng.integers(0, 4, size=20_000, dtype=np.uint8))
            assert nk > 3 * 32_768 // 2
            p = hp(k, cutoff_min=1)
            out, lut, st = c2.process_bin(p, img, nk, packs)
            want_out, want_lut, want_st = O.process_bin(op(p), img, nk)
            assert np.array_equal(st, want_st) and np.array_equal(out, want_out) and np.array_equal(lut, want_lut), k
    finally:
        c2.close()
        monkeypatch.delenv("KMC_HIP_DEBUG_PORTION_LOG2")
        capi.Context((0,)).close()  # kmc_hip_init re-reads the variable: the session's context is back on full portions


# ------------------------------------------------------------------------------------------------ many bins in one call
def _run_batch(ctx, p, bins, n_streams=0, shrink_cap_of=None, dev=0):
    """bins: list of (image, n_rec, pack_bytes, ...). Uploads everything, ONE kmc_hip_process_bins_device call, returns per bin
    (out bytes array, lut array, stats[4])."""
    rec = ctx.out_rec_bytes(p)
    nl = ctx.lut_entries(p)
    n = len(bins)
    descs = (capi.BinDesc * n)()
    allocs, caps = [], []
    for i, b in enumerate(bins):
        img, nrec, packs = b[0], b[1], b[2]
        ps = np.concatenate([[0], np.cumsum(packs)]).astype(np.uint64)
        cap = ((nrec + 1) // max(p.cutoff_min, 1)) * rec
        if shrink_cap_of is not None and i == shrink_cap_of:
            cap = rec  # room for one record only
        d_in = ctx.malloc(img.size + 256, dev)
        d_ps = ctx.malloc(ps.nbytes, dev)
        d_out = ctx.malloc(cap + 256, dev)
        d_lut = ctx.malloc(max(nl, 1) * 8, dev)
        d_small = ctx.malloc(64, dev)
        ctx.h2d(d_in, np.concatenate([img, np.zeros(256, dtype=np.uint8)]), dev)
        ctx.h2d(d_ps, ps, dev)
        allocs.append((d_in, d_ps, d_out, d_lut, d_small))
        caps.append(cap)
        descs[i] = capi.BinDesc(d_in, img.size, nrec, d_ps, packs.size, d_out, cap, d_small + 32, d_lut, d_small)
    err = None
    try:
        ctx.process_bins_device(p, descs, n_streams, dev)
        ctx.synchronize(dev)
    except capi.KmcHipError as e:
        err = e
    out = []
    for i in range(n):
        small = np.zeros(8, dtype=np.uint64)
        ctx.d2h(small, allocs[i][4], dev)
        ob = int(small[4])
        o = np.zeros(min(ob, caps[i]), dtype=np.uint8)
        if o.size:
            ctx.d2h(o, allocs[i][2], dev)
        lut = np.zeros(max(nl, 1), dtype=np.uint64)
        if nl:
            ctx.d2h(lut, allocs[i][3], dev)
        out.append((o, lut[:nl], small[:4].copy()))
    for a in allocs:
        for d in a:
            ctx.free(d, dev)
    return out, err


@pytest.mark.parametrize("n_bins,reads,genome,pl,streams,k", [(512, 400_000, 2_000_000, 7, 0, 27), (16, 400_000, 2_000_000, 7, 3, 27), (64, 60_000, 300_000, 3, 16, 27),
                                                              (23, 300_000, 1_500_000, 7, 1, 27), (40, 100_000, 500_000, 5, 2, 25), (21, 100_000, 500_000, 3, 2, 55),
                                                              (10, 100_000, 500_000, 4, 1, 32), (9, 40_000, 200_000, 3, 1, 127),
                                                              (8, 100_000, 500_000, 0, 1, 27), (8, 60_000, 300_000, 0, 1, 55)])  # pl 0: KFF records, bins grouped
def test_many_bins_in_one_call_match_the_oracle_per_bin(ctx, n_bins, reads, genome, pl, streams, k):
    """configs[2]'s shape in small: one read set cut into signature bins (30x coverage, lut_prefix_len 7 as KMC picks for 30 Gbp),
    ALL bins through ONE kmc_hip_process_bins_device call (bin i on stream i mod n_streams, several host threads enqueueing; consecutive
    bins of a stream share one sort, tagged in the spare bits of the top radix digit: groups of 4 at k = 27, 55, 127, of 16 at k = 25, none
    at k = 32), every bin compared with the oracle bit for bit — suffix records, LUT, tallies."""
    bins = capi.synth_bins(seed=2026, genome_len=genome, n_reads=reads, k=k, n_bins=n_bins)
    p = hp(k, lut_prefix_len=pl, output_type=0 if pl else 1)
    got, err = _run_batch(ctx, p, bins, streams)
    assert err is None, err
    tot = np.zeros(4, dtype=np.uint64)
    for i, (img, nrec, packs, _) in enumerate(bins):
        w_out, w_lut, w_st = O.process_bin(op(p), img, nrec)
        assert np.array_equal(got[i][2], w_st), (i, got[i][2], w_st)
        assert np.array_equal(got[i][0], w_out), (i, _first_diff(got[i][0], w_out))
        assert np.array_equal(got[i][1], w_lut), i
        tot += w_st
    assert int(tot[3]) == sum(b[1] for b in bins)


def test_deferred_device_error_of_an_early_async_bin_is_not_lost(ctx):
    """an error raised by an asynchronous bin used to be wiped by the next bin on the same stream slot. 40 bins on
    2 streams, bin 1's out_capacity holds one record: kmc_hip_synchronize must still report KMC_HIP_ECAPACITY, the other bins
    must be correct, and the error must be gone afterwards."""
    bins = capi.synth_bins(seed=3, genome_len=100_000, n_reads=20_000, k=27, n_bins=40)
    p = hp(27, cutoff_min=1)
    got, err = _run_batch(ctx, p, bins, n_streams=2, shrink_cap_of=1)
    assert err is not None and err.code == -5, err
    for i, (img, nrec, packs, _) in enumerate(bins):
        if i == 1:
            continue
        w_out, w_lut, w_st = O.process_bin(op(p), img, nrec)
        assert np.array_equal(got[i][2], w_st) and np.array_equal(got[i][0], w_out) and np.array_equal(got[i][1], w_lut), i
    ctx.synchronize()  # the sticky word was cleared by the failing synchronize
    got, err = _run_batch(ctx, p, bins[:4], n_streams=2)
    assert err is None


def test_size_and_n_rec_must_both_be_zero_or_neither(ctx):
    rng = np.random.default_rng(5)
    img, nk, packs = binsynth.random_bin(rng, 27, 50)
    for image, n in ((img, 0), (np.zeros(0, dtype=np.uint8), 7)):
        with pytest.raises(capi.KmcHipError) as e:
            ctx.process_bin(hp(27), image, n, packs if image.size else None)
        assert e.value.code == -4


def test_sort_records_into_a_second_buffer(ctx):
    """kmc_hip_sort_records_into: what the SortFunction adapter binds (result in `tmp` when rec_len is odd)."""
    rng = np.random.default_rng(77)
    for n in (0, 1, 70_001):
        a = rng.integers(0, 2**54, size=(n, 1), dtype=np.uint64)
        src, dst = a.copy(), np.zeros_like(a)
        rc = ctx.L.kmc_hip_sort_records_into(ctx.h, 0, src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), n, 1, 7)
        assert rc == 0
        assert np.array_equal(src, a) or n < 2  # the source is not required to survive, but is not written for n >= 2 either
        assert np.array_equal(dst[:, 0], np.sort(a[:, 0]))


# ------------------------------------------------------------------------------------------------ drop-ins at bench scale
def _kmc(exe, flags, fq, out, tmp, env=None, timeout=900):
    os.makedirs(tmp, exist_ok=True)
    r = subprocess.run([exe, *flags, fq, out, tmp], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r


def _hip_env(**extra):
    return dict(os.environ, KMC_HIP_LIB=capi.lib_path(), **extra)


@pytest.mark.parametrize("k,reads,genome", [(27, 13_300_000, 66_000_000), (55, 4_000_000, 20_000_000)], ids=["k27-2Gbp", "k55-0.6Gbp"])
def test_dropin_database_at_bench_scale(k, reads, genome, ref_bins, tmp_path):
    """bit-exactness was only shown up to ~12 M k-mers. Here the drop-in (kmc_hip: worker + reader plug-ins,
    16 workers, 8 reader threads) and the unmodified reference (-sr1) count the configs[1]-sized input (1.65 G k-mers at k=27):
    .kmc_pre/.kmc_suf must be byte-identical."""
    if ref_bins is None:
        pytest.skip("oracle/_ref binaries were not shipped")
    import shutil

    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > reads * 316 * 3 else str(tmp_path)
    import tempfile

    with tempfile.TemporaryDirectory(dir=base) as td:
        fq = os.path.join(td, "in.fq")
        capi.synth_fastq(fq, seed=2026, genome_len=genome, n_reads=reads)
        _kmc(ref_bins["kmc"], [f"-k{k}", "-t64", "-m64", "-sr1"], fq, os.path.join(td, "ref"), os.path.join(td, "t_ref"))
        r = _kmc(ref_bins["kmc_hip"], [f"-k{k}", "-t64", "-m64", "-sr16"], fq, os.path.join(td, "hip"), os.path.join(td, "t_hip"),
                 env=_hip_env(KMC_HIP_VERBOSE="1"))
        for ext in (".kmc_pre", ".kmc_suf"):
            assert _md5(os.path.join(td, "ref" + ext)) == _md5(os.path.join(td, "hip" + ext)), ext
        log = os.path.join(ROOT, "gpurun_out", f"dropin_bench_scale_k{k}.log")
        if os.path.isdir(os.path.dirname(log)):
            open(log, "w").write(r.stdout + r.stderr)


@pytest.mark.parametrize("flags", [["-k27"], ["-k21"], ["-k55"], ["-k127"]], ids=lambda f: "".join(f))
def test_narrow_boundary_sortfunction_adapter(flags, ref_bins, tmp_path):
    """kmc_amd/bin/kmc_hipsort = the reference with ONLY its SortFunction replaced by the GPU sort (hip_sort_function.h; the
    reference's own CKmerBinSorter expands and compacts on the CPU): rec_len even (k=27: 8, k=127: 32 -> result in place) and
    odd (k=21: 7, k=55: 15 -> result in tmp). Database byte-identical to the unmodified reference, both with one sorter."""
    exe = os.path.join(ROOT, "kmc_amd", "bin", "kmc_hipsort")
    if ref_bins is None or not os.path.exists(exe):
        pytest.skip("kmc_amd/bin/kmc_hipsort was not shipped")
    from kmc_amd import synth

    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, **synth.CONFIGS["C1"])
    _kmc(ref_bins["kmc"], [*flags, "-t4", "-sr1"], fq, str(tmp_path / "ref"), str(tmp_path / "t_ref"))
    _kmc(exe, [*flags, "-t4", "-sr1"], fq, str(tmp_path / "hs"), str(tmp_path / "t_hs"), env=_hip_env())
    for ext in (".kmc_pre", ".kmc_suf"):
        assert _md5(str(tmp_path / ("ref" + ext))) == _md5(str(tmp_path / ("hs" + ext))), ext


def test_kff_output_end_to_end(ref_bins, tmp_path):
    """-okff. The compaction kernel writes KFF record order (big-endian counter, no LUT, kb_sorter.h:1043-1049);
    the reference's own KFF writer packs it. The .kff file must be byte-identical to the unmodified reference's."""
    if ref_bins is None:
        pytest.skip("oracle/_ref binaries were not shipped")
    from kmc_amd import synth

    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, **synth.CONFIGS["C1"])
    for k in (27, 55):
        _kmc(ref_bins["kmc"], [f"-k{k}", "-okff", "-t4", "-sr1"], fq, str(tmp_path / f"ref{k}"), str(tmp_path / f"t_ref{k}"))
        _kmc(ref_bins["kmc_hip"], [f"-k{k}", "-okff", "-t8", "-sr4"], fq, str(tmp_path / f"hip{k}"), str(tmp_path / f"t_hip{k}"), env=_hip_env())
        assert _md5(str(tmp_path / f"ref{k}.kff")) == _md5(str(tmp_path / f"hip{k}.kff")), k


@pytest.mark.parametrize("readers,flags", [("1", ["-sr1"]), ("8", ["-t16", "-sr12"]), ("8", ["-t8", "-sr4", "-r"])], ids=["r1-sr1", "r8-sr12", "r8-sr4-ram"])
def test_reader_plugin_with_the_hip_worker(readers, flags, ref_bins, tmp_path):
    """kmc_hip (worker + reader plug-ins) for several reader/worker counts, disk and RAM-only (-r) temporaries, against the
    reference's -sr1 database; kmc_hip_sr (reference reader kept) must agree too."""
    if ref_bins is None:
        pytest.skip("oracle/_ref binaries were not shipped")
    from kmc_amd import synth

    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, **synth.CONFIGS["C1"])
    ram = [f for f in flags if f == "-r"]
    _kmc(ref_bins["kmc"], ["-k27", "-sr1", *ram], fq, str(tmp_path / "ref"), str(tmp_path / "t_ref"))
    _kmc(ref_bins["kmc_hip"], ["-k27", *flags], fq, str(tmp_path / "hip"), str(tmp_path / "t_hip"), env=_hip_env(KMC_HIP_READERS=readers))
    sr = os.path.join(ROOT, "kmc_amd", "bin", "kmc_hip_sr")
    outs = ["hip"]
    if os.path.exists(sr):
        _kmc(sr, ["-k27", *flags], fq, str(tmp_path / "hipsr"), str(tmp_path / "t_hipsr"), env=_hip_env())
        outs.append("hipsr")
    for o in outs:
        for ext in (".kmc_pre", ".kmc_suf"):
            assert _md5(str(tmp_path / ("ref" + ext))) == _md5(str(tmp_path / (o + ext))), (o, ext)


# ------------------------------------------------------------------------------------------------ more than one device
def _n_devices():
    try:
        return capi.load().kmc_hip_device_count()
    except Exception:
        return 0


def test_two_devices_worker_allreduce_and_wide_records(ref_bins, tmp_path):
    """the multi-device code had never met a second device. On a box with >= 2 GPUs: a context on devices
    (0, 1) — k=200 (SIZE 7: > 64 KiB of dynamic LDS, needs the per-device function attribute) on the SECOND device against the
    oracle; kmc_hip_allreduce_stats over 2 devices; the drop-in with KMC_HIP_DEVICES=0,1 against the reference database."""
    if _n_devices() < 2:
        pytest.skip("needs two GPUs")
    c2 = capi.Context((0, 1))
    try:
        rng = np.random.default_rng(200)
        img, nk, packs = binsynth.random_bin(rng, 200, 1500, max_extra=60)
        p = hp(200, lut_prefix_len=4)
        for dev in (1, 0):
            out, lut, st = c2.process_bin(p, img, nk, packs, dev=dev)
            w = O.process_bin(op(p), img, nk)
            assert np.array_equal(out, w[0]) and np.array_equal(lut, w[1]) and np.array_equal(st, w[2]), dev
        a = np.array([[1, 2, 3, 2**40], [10, 20, 30, 5]], dtype=np.uint64)
        r = c2.allreduce_stats(a)
        assert np.array_equal(r[0], a.sum(axis=0)) and np.array_equal(r[1], a.sum(axis=0))
    finally:
        c2.close()
    if ref_bins is None:
        return
    from kmc_amd import synth

    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, **synth.CONFIGS["C1"])
    _kmc(ref_bins["kmc"], ["-k27", "-sr1"], fq, str(tmp_path / "ref"), str(tmp_path / "t_ref"))
    _kmc(ref_bins["kmc_hip"], ["-k27", "-t8", "-sr6"], fq, str(tmp_path / "hip"), str(tmp_path / "t_hip"), env=_hip_env(KMC_HIP_DEVICES="0,1"))
    for ext in (".kmc_pre", ".kmc_suf"):
        assert _md5(str(tmp_path / ("ref" + ext))) == _md5(str(tmp_path / ("hip" + ext))), ext


def test_eight_logical_devices_run_their_lpt_shards_concurrently_and_reduce_the_tallies():
    """configs[3]'s control flow without the hardware: a context that names device 0 EIGHT times — every logical device has its own streams, slots and
    buffers, exactly as eight physical devices would —, 64 signature bins sharded over them by LPT on their k-mer counts (kmc_amd/sharding.py, the order KMC hands bins to
    sorters: queues.h:499-558, :2045-2146), one host thread per logical device running its shard through kmc_hip_process_bins_device at the same time as the seven others,
    then kmc_hip_allreduce_stats over the eight members. Every bin byte-equal to the SAME bin from a one-device context, per-device tallies summing to the one-device
    totals. RCCL may refuse a communicator that names one physical device more than once: then the reduce leg says so (the tallies are summed on the host for the check) —
    the collective itself runs on two real devices in test_two_devices_worker_allreduce_and_wide_records."""
    import threading

    from kmc_amd import sharding

    bins = capi.synth_bins(seed=11, genome_len=3_000_000, n_reads=600_000, k=27, n_bins=64, n_threads=4)
    p = hp(27, lut_prefix_len=7)
    c1 = capi.Context((0,))
    try:
        want, err = _run_batch(c1, p, bins, 0)
        assert err is None, err
    finally:
        c1.close()
    shards = sharding.lpt_assign([b[1] for b in bins], 8)
    assert sorted(i for sh in shards for i in sh) == list(range(64)) and all(shards)
    loads = [sum(bins[i][1] for i in sh) for sh in shards]
    assert max(loads) <= 1.15 * (sum(loads) / 8), loads  # LPT on 64 near-equal bins: a few per cent of imbalance
    c8 = capi.Context((0,) * 8)
    try:
        got, errs = [None] * 64, [None] * 8
        gate = threading.Barrier(8)

        def rank(r):
            try:
                gate.wait(timeout=60)  # all eight enter the library together
                outs, e = _run_batch(c8, p, [bins[i] for i in shards[r]], 0, dev=r)
                if e is not None:
                    raise e
                for i, o in zip(shards[r], outs):
                    got[i] = o
            except Exception as e:  # noqa: BLE001
                errs[r] = repr(e)

        ths = [threading.Thread(target=rank, args=(r,)) for r in range(8)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert errs == [None] * 8, errs
        for i in range(64):
            assert all(np.array_equal(a, b) for a, b in zip(got[i], want[i])), i
        per_dev = np.array([np.sum([got[i][2] for i in shards[r]], axis=0, dtype=np.uint64) for r in range(8)], dtype=np.uint64)
        total = np.sum([w[2] for w in want], axis=0, dtype=np.uint64)
        assert np.array_equal(per_dev.sum(axis=0, dtype=np.uint64), total)
        try:
            red = c8.allreduce_stats(per_dev)
            assert all(np.array_equal(red[r], total) for r in range(8)), (red, total)
            print("kmc_hip_allreduce_stats over 8 logical devices of one GPU: ok")
        except capi.KmcHipError as e:
            # an 8-member communicator over ONE physical device: RCCL's own rule, not this library's (ncclCommInitAll rejects duplicate devices)
            assert "nccl" in str(e).lower(), e
            print("kmc_hip_allreduce_stats over 8 logical devices of one GPU refused by RCCL:", e)
    finally:
        c8.close()


# ------------------------------------------------------------------------------------------------ k_bucket_rank (default path of one-word k-mers)
def test_rank_path_takes_repeats_in_chunks_giant_buckets_on_their_own_and_hands_an_enormous_one_back(ctx):
    """default mode: the top bytes through HBM, tiles ranked and counted inside LDS by k_bucket_rank. (The many-bins tests above run this path on ordinary
    data.) Here: every k-mer ~1600 times — buckets longer than the room at the end of a window, tiles longer than the capacity, taken in chunks of whole
    buckets (one-word records since round 6: buckets beyond BR_MID records go through the arena); then one k-mer more often than a tile holds records (8000 times;
    k = 27: the arena; k = 55, 127: k_giant_tiles) — nothing comes back; then one k-mer more than a million times — one-word records: the arena takes a bucket of any
    length, nothing comes back; two-word records: the bin is run again with LSD passes."""
    small = capi.backend_kind() != 0  # the emulated host library: the paths, not the scale (GT_MAX_RECORDS is 2048 there)
    p = hp(27)
    bins = capi.synth_bins(seed=3, genome_len=6000 if not small else 2000, n_reads=80_000 if not small else 1000, k=27, n_bins=4, err=0.0)
    t0 = ctx.local_sort_totals()
    got, err = _run_batch(ctx, p, bins, 1)
    assert err is None, err
    for i, (img, nrec, packs, _) in enumerate(bins):
        w = O.process_bin(op(p), img, nrec)
        assert all(np.array_equal(a, b) for a, b in zip(got[i], w)), i
    t1 = ctx.local_sort_totals()
    assert t1["hybrid_groups"] > t0["hybrid_groups"] and t1["redo_groups"] == t0["redo_groups"], (t0, t1)
    g0 = ctx.path_counters()
    for k, kw, err_rate in ((27, {}, 0.0), (27, dict(cutoff_min=1, lut_prefix_len=0, output_type=1), 0.01), (55, dict(lut_prefix_len=3), 0.0), (127, dict(lut_prefix_len=3), 0.002)):
        pk = hp(k, **kw)
        bins = capi.synth_bins(seed=5, genome_len=300, n_reads=20_000 if not small else 1500, k=k, n_bins=2, err=err_rate, read_len=max(150, k + 40))
        got, err = _run_batch(ctx, pk, bins, 1)
        assert err is None, err
        for i, (img, nrec, packs, _) in enumerate(bins):
            w = O.process_bin(op(pk), img, nrec)
            assert all(np.array_equal(a, b) for a, b in zip(got[i], w)), (k, i)
    t2, g1 = ctx.local_sort_totals(), ctx.path_counters()
    assert t2["hybrid_groups"] > t1["hybrid_groups"] and t2["redo_groups"] == t1["redo_groups"], (t1, t2)
    assert g1["giant_tiles"] >= g0["giant_tiles"] + 4 and g1["giant_records"] > g0["giant_records"], (g0, g1)
    arena = os.environ.get("KMC_HIP_ARENA", "1") != "0"
    for k, comes_back in ((27, not arena), (55, True)):
        pk = hp(k)
        before, g2 = ctx.local_sort_totals()["redo_groups"], ctx.path_counters()
        bins = capi.synth_bins(seed=5, genome_len=160 if k == 27 else 190, n_reads=1_300_000 if not small else 2600, k=k, n_bins=2, err=0.0)
        got, err = _run_batch(ctx, pk, bins, 1)
        assert err is None, err
        for i, (img, nrec, packs, _) in enumerate(bins):
            w = O.process_bin(op(pk), img, nrec)
            assert all(np.array_equal(a, b) for a, b in zip(got[i], w)), (k, i)
        assert (ctx.local_sort_totals()["redo_groups"] > before) == comes_back, (k, comes_back)
        if not comes_back:
            g3 = ctx.path_counters()
            assert g3["giant_records"] - g2["giant_records"] > (2_000_000 if not small else 2000), (g2, g3)


# ------------------------------------------------------------------------------------------------ several bins per host-boundary call
HOST_GROUP_CASES = [
    (27, dict(lut_prefix_len=3), 7),
    (27, dict(lut_prefix_len=7, cutoff_min=1, cutoff_max=50, counter_max=20), 4),
    (55, dict(lut_prefix_len=3), 5),
    (127, dict(lut_prefix_len=3, cutoff_min=1), 6),
    (32, dict(lut_prefix_len=4), 3),       # no spare bits: every bin sorted on its own, still one call
    (25, dict(lut_prefix_len=1), 15),      # 6 spare bits: groups of 16
    (27, dict(output_type=1), 4),          # KFF records
    (21, dict(lut_prefix_len=1, both_strands=0, without_output=1), 4),
]


@pytest.mark.parametrize("k,kw,n_bins", HOST_GROUP_CASES, ids=lambda v: str(v) if not isinstance(v, dict) else "-".join(f"{a}{b}" for a, b in v.items()))
@pytest.mark.parametrize("hybrid", [0, 1], ids=["lsd", "default"])
def test_host_boundary_with_several_bins_per_call_matches_the_oracle_per_bin(ctx, k, kw, n_bins, hybrid):
    """kmc_hip_process_bins_submit/_wait: the bins of one call are uploaded together and sorted together (tags in the spare bits of the top digit), every
    bin gets its own records, LUT and tallies — those of kmc_hip_process_bin. With caller-supplied packs and without, an empty bin among them."""
    before = ctx.set_hybrid(hybrid)
    try:
        bins = capi.synth_bins(seed=41, genome_len=40_000, n_reads=8_000, k=k, n_bins=n_bins)
        p = hp(k, **kw)
        hb = [(img, nrec, packs if i % 2 else None) for i, (img, nrec, packs, _) in enumerate(bins)]
        hb.insert(2, (np.zeros(0, np.uint8), 0, None))
        got = ctx.process_bins_host(p, hb, slot=5)
        for i, (img, nrec, _) in enumerate(hb):
            w = O.process_bin(op(p), img, nrec) if nrec else (np.zeros(0, np.uint8), np.zeros(ctx.lut_entries(p), np.uint64), np.zeros(4, np.uint64))
            assert np.array_equal(got[i][0], w[0]) and np.array_equal(got[i][1], w[1]) and np.array_equal(got[i][2], w[2]), (i, got[i][2], w[2])
    finally:
        ctx.set_hybrid(before)


def _host_boundary_storm(c, p, bins, n_threads, group, passes):
    """bench.py's host-boundary pattern: thread t owns stream slots 2t, 2t+1 and keeps two calls in flight; every bin's out_bytes + four tallies are kept per
    pass. Returns [pass][bin] -> (out_bytes, unique, below, above, total); raises with the library's message (the watchdog's diagnostics) on any error."""
    import ctypes as C
    import threading
    from kmc_amd.capi import HostBin
    L = c.L
    rec, nl = c.out_rec_bytes(p), c.lut_entries(p)
    caps = [((b[1] + 1) // max(p.cutoff_min, 1)) * rec for b in bins]
    outs = [c.host_alloc(max(caps) + 256) for _ in range(2 * n_threads * group)]
    luts = [c.host_alloc(max(nl, 1) * 8) for _ in range(2 * n_threads * group)]
    pins = []
    for img, nrec, packs, _ in bins:
        a = c.host_alloc(img.size + 256)
        a[:img.size] = img
        pins.append(a)
    errors, results = [], []

    def worker(tid, res):
        calls = [list(range(len(bins)))[q:q + group] for q in range(tid * group, len(bins), n_threads * group)]
        inflight = []
        ob, st = (C.c_uint64 * group)(), (C.c_uint64 * (4 * group))()

        def wait(sl, mine):
            rc = (L.kmc_hip_process_bin_wait if group == 1 else L.kmc_hip_process_bins_wait)(c.h, 0, 2 * tid + sl, ob, st)
            if rc:
                raise RuntimeError(L.kmc_hip_last_error(c.h).decode())
            for j, i in enumerate(mine):
                res[i] = (ob[j],) + tuple(st[4 * j + q] for q in range(4))

        try:
            for j, mine in enumerate(calls):
                sl = j & 1
                if len(inflight) == 2:
                    wait(*inflight.pop(0))
                arr = (HostBin * group)()
                for q, i in enumerate(mine):
                    pk = bins[i][2]
                    buf = (2 * tid + sl) * group + q
                    arr[q] = HostBin(pins[i].ctypes.data, bins[i][0].size, bins[i][1], pk.ctypes.data, pk.size, outs[buf].ctypes.data, caps[i], luts[buf].ctypes.data)
                if group == 1:
                    h = arr[0]
                    rc = L.kmc_hip_process_bin_submit(c.h, 0, 2 * tid + sl, C.byref(p), C.c_void_p(h.superkmers), h.size, h.n_rec, C.c_void_p(h.pack_bytes), h.n_packs,
                                                      C.c_void_p(h.out_suffix), h.out_capacity, C.c_void_p(h.lut))
                else:
                    rc = L.kmc_hip_process_bins_submit(c.h, 0, 2 * tid + sl, C.byref(p), arr, len(mine))
                if rc:
                    raise RuntimeError(L.kmc_hip_last_error(c.h).decode())
                inflight.append((sl, mine))
            while inflight:
                wait(*inflight.pop(0))
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d, call %d of %d (bins %s): %r" % (tid, j, len(calls), mine, e))

    try:
        for ps in range(passes):
            res = [None] * len(bins)
            ths = [threading.Thread(target=worker, args=(t, res)) for t in range(n_threads)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            if errors:
                raise RuntimeError("pass %d: %s" % (ps, "; ".join(errors)))
            results.append(res)
    finally:
        for a in outs + luts + pins:
            c.host_free(a)
    return results


@pytest.mark.parametrize("group", [1, 4], ids=["one-bin-per-call", "four-bins-per-call"])
def test_host_boundary_storm_sixteen_slots_in_flight(group):
    """The pattern that once (round 3, 1 leg of 11) ended in KMC_HIP_EINTERNAL "look-back watchdog": 8 host threads x 2 stream slots, default mode (one-word
    k-mers: HBM passes + k_bucket_rank), 64 bins of ~10 M k-mers from pinned memory, 30 passes back to back on a FRESH context (buffers grow in the first pass, while
    the other slots are busy). Every pass must give every bin the tallies of the first pass, three bins of it are checked against the oracle, and an error carries
    the watchdog's diagnostics (which look-back, which tile, what it saw, for how long)."""
    n_bins, passes = 64, 30
    if capi.backend_kind() != 0:
        n_bins, passes = 8, 2  # the emulated library: the pattern, not the scale
    scale = 1 if capi.backend_kind() == 0 else 0
    bins = capi.synth_bins(seed=2026, genome_len=25_000_000 if scale else 20_000, n_reads=5_300_000 if scale else 1_000, k=27, n_bins=n_bins)
    p = hp(27, lut_prefix_len=7 if scale else 3)
    c = capi.Context((0,))
    try:
        before = c.path_counters()
        res = _host_boundary_storm(c, p, bins, n_threads=8 if scale else 2, group=group, passes=passes)
        assert c.path_counters()["rank_count"] > before["rank_count"]  # the default path ran
    finally:
        c.close()
    for ps in range(1, len(res)):
        assert res[ps] == res[0], "pass %d differs from pass 0 at bins %s" % (ps, [i for i in range(n_bins) if res[ps][i] != res[0][i]][:8])
    assert sum(r[4] for r in res[0]) == sum(b[1] for b in bins)
    order = sorted(range(n_bins), key=lambda i: bins[i][1])
    for i in (order[0], order[n_bins // 2], order[-1]):
        w = O.process_bin(op(p), bins[i][0], bins[i][1])
        assert res[0][i] == (w[0].size,) + tuple(int(x) for x in w[2]), i


def test_host_boundary_group_redo_errors_and_slot_reuse(ctx):
    small = capi.backend_kind() != 0  # the emulated host library: GT_MAX_RECORDS is 2048 there
    # two-word records with one k-mer more often than k_giant_tiles takes (2^20 copies): the bins come back for LSD passes over every byte, inside the same call
    bins = capi.synth_bins(seed=5, genome_len=190, n_reads=1_300_000 if not small else 2600, k=55, n_bins=2, err=0.0)
    p55 = hp(55)
    r0 = ctx.local_sort_totals()["redo_groups"]
    got = ctx.process_bins_host(p55, [(b[0], b[1], b[2]) for b in bins])
    for i, b in enumerate(bins):
        w = O.process_bin(op(p55), b[0], b[1])
        assert all(np.array_equal(x, y) for x, y in zip(got[i], w)), i
    assert ctx.local_sort_totals()["redo_groups"] > r0
    bins = capi.synth_bins(seed=5, genome_len=300, n_reads=20_000 if not small else 1500, k=27, n_bins=3, err=0.0)
    p = hp(27)
    img, nk, packs = binsynth.random_bin(np.random.default_rng(1), 27, 500, max_extra=20)
    with pytest.raises(capi.KmcHipError) as e:
        ctx.process_bins_host(p, [(bins[0][0], bins[0][1], bins[0][2]), (img, nk + 1, packs)])
    assert e.value.code == -4  # KMC_HIP_ECORRUPT
    with pytest.raises(capi.KmcHipError) as e:
        ctx.process_bins_host(hp(27, cutoff_min=1), [(img, nk, packs)] * 2, out_capacity=8)
    assert e.value.code == -5  # KMC_HIP_ECAPACITY
    with pytest.raises(capi.KmcHipError) as e:
        ctx.process_bins_host(p, [(img, nk, packs)] * 17)
    assert e.value.code == -1
    w = O.process_bin(op(p), img, nk)  # the slot is usable afterwards, by either kind of call
    got = ctx.process_bins_host(p, [(img, nk, packs)])
    assert all(np.array_equal(x, y) for x, y in zip(got[0], w))
    got = ctx.process_bin(p, img, nk, packs)
    assert all(np.array_equal(x, y) for x, y in zip(got, w))


# ------------------------------------------------------------------------------------------------ one bin over several devices
MULTI_CASES = [
    (27, dict(lut_prefix_len=3), 2),
    (27, dict(lut_prefix_len=7, cutoff_min=1), 3),
    (55, dict(lut_prefix_len=3), 2),
    (127, dict(lut_prefix_len=3, cutoff_min=1), 4),
    (27, dict(output_type=1), 2),
    (21, dict(lut_prefix_len=1, both_strands=0, without_output=1), 2),
]


@pytest.mark.parametrize("k,kw,n_dev", MULTI_CASES, ids=lambda v: str(v).replace(" ", "") if not isinstance(v, dict) else "-".join(f"{a}{b}" for a, b in v.items()))
def test_process_bin_multi_over_logical_devices_matches_the_oracle(k, kw, n_dev):
    """kmc_hip_process_bin_multi (the oversized-bin path) on a context that names device 0 n_dev times: every logical device has its
    own streams and buffers, so each step of the path runs as it would over n_dev GPUs — shares of packs, top-byte histograms, range cuts, one
    k_onesweep pass, the exchange (copies here; ncclSend / ncclRecv between distinct GPUs: the test below and, emulated, tests/hostlib_two_devices.py),
    per-range sort + compaction, ordered emission — and the result must be the bin's."""
    c = capi.Context((0,) * n_dev)
    try:
        bins = capi.synth_bins(seed=31, genome_len=60_000, n_reads=9_000, k=k, n_bins=2, read_len=150)
        p = hp(k, **kw)
        for i, (img, nrec, packs, _) in enumerate(bins):
            w = O.process_bin(op(p), img, nrec)
            for pk in (packs, None):
                got = c.process_bin(p, img, nrec, pk, multi=True)
                assert np.array_equal(got[0], w[0]) and np.array_equal(got[1], w[1]) and np.array_equal(got[2], w[2]), (i, pk is None, got[2], w[2])
    finally:
        c.close()


def test_process_bin_multi_equals_process_bin_on_a_large_bin_and_on_edge_cases(ctx):
    """a bin of ~10 M k-mers: multi over (0, 0, 0) == kmc_hip_process_bin on one device, byte for byte (size-independent property; the oracle has
    checked the latter at this size elsewhere); then tiny bins (fewer k-mers than devices), the empty bin, a bin whose records share their top byte,
    and the error returns"""
    c = capi.Context((0, 0, 0))
    try:
        (img, nrec, packs, _), = capi.synth_bins(seed=77, genome_len=3_000_000, n_reads=100_000, k=27, n_bins=1)
        p = hp(27, lut_prefix_len=7)
        want = ctx.process_bin(p, img, nrec, packs)
        got = c.process_bin(p, img, nrec, packs, multi=True)
        assert nrec > 5_000_000 and all(np.array_equal(a, b) for a, b in zip(got, want))
        rng = np.random.default_rng(3)
        p1 = hp(27, lut_prefix_len=3, cutoff_min=1)
        for n_sk, mx in ((1, 0), (1, 5), (2, 0), (3, 1), (40, 10)):
            img, nk, packs = binsynth.random_bin(rng, 27, n_sk, max_extra=mx)
            w = O.process_bin(op(p1), img, nk)
            got = c.process_bin(p1, img, nk, packs, multi=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, w)), (n_sk, mx)
        out, lut, st = c.process_bin(p1, np.zeros(0, np.uint8), 0, None, multi=True)
        assert out.size == 0 and not lut.any() and not st.any()
        (img, nk, packs, _), = capi.synth_bins(seed=5, genome_len=300, n_reads=1500, k=27, n_bins=1, err=0.0)
        p2 = hp(27)
        w = O.process_bin(op(p2), img, nk)
        got = c.process_bin(p2, img, nk, packs, multi=True)
        assert all(np.array_equal(a, b) for a, b in zip(got, w))
        with pytest.raises(capi.KmcHipError) as e:
            c.process_bin(p2, img, nk + 1, packs, multi=True)
        assert e.value.code == -4  # KMC_HIP_ECORRUPT
        with pytest.raises(capi.KmcHipError) as e:
            c.process_bin(hp(27, cutoff_min=1), img, nk, packs, out_capacity=8, multi=True)
        assert e.value.code == -5  # KMC_HIP_ECAPACITY
        # the context is still usable
        got = c.process_bin(p2, img, nk, packs, multi=True)
        assert all(np.array_equal(a, b) for a, b in zip(got, w))
    finally:
        c.close()


def test_process_bin_multi_over_two_gpus_exchanges_through_rccl():
    if _n_devices() < 2:
        pytest.skip("needs two GPUs")
    c = capi.Context(tuple(range(min(_n_devices(), 8))))
    try:
        for k, kw in ((27, dict(lut_prefix_len=3)), (55, dict(lut_prefix_len=3, cutoff_min=1))):
            (img, nrec, packs, _), = capi.synth_bins(seed=31, genome_len=200_000, n_reads=30_000, k=k, n_bins=1)
            p = hp(k, **kw)
            w = O.process_bin(op(p), img, nrec)
            got = c.process_bin(p, img, nrec, packs, multi=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, w)), k
    finally:
        c.close()


# ------------------------------------------------------------------------------------------------ globally ordered database
def read_kmc2_database(db):
    """(k, p, counter bytes, per-bin list of (records bytes, LUT counts)) of a KMC database as kmc writes it (kb_completer.cpp:287-320): 'KMCP',
    per emitted bin uint64[4^p] running record offsets, uint64 total, uint32 signature map, 72-byte header, 'KMCP'."""
    pre = np.fromfile(db + ".kmc_pre", dtype=np.uint8)
    suf = np.fromfile(db + ".kmc_suf", dtype=np.uint8)[4:-4]
    hdr = pre[-76:-4]
    k, mode, cs, p, sig_len = (int(x) for x in hdr[:20].view(np.uint32))
    lut_area = pre[4: pre.size - 4 - 72 - ((1 << (2 * sig_len)) + 1) * 4].view(np.uint64)
    n_entries = 1 << (2 * p)
    n_bins = (lut_area.size - 1) // n_entries
    offs = np.concatenate([lut_area[: n_bins * n_entries], lut_area[-1:]])
    rb = (k - p) // 4 + cs
    bins = []
    for b in range(n_bins):
        o = offs[b * n_entries: (b + 1) * n_entries + 1].astype(np.int64)
        bins.append((suf[o[0] * rb: o[-1] * rb].copy(), np.diff(o).astype(np.uint64)))
    return k, p, cs, bins


def read_kmc1_database(db):
    """(lut_prefix_len, LUT uint64[4^p], record bytes) of a database as kmc_tools writes it (kmc1_db_writer.h:309-370)"""
    pre = np.fromfile(db + ".kmc_pre", dtype=np.uint8)
    suf = np.fromfile(db + ".kmc_suf", dtype=np.uint8)[4:-4]
    hdr = pre[-72:-8]
    p = int(hdr[12:16].view(np.uint32)[0])
    lut = pre[4: 4 + (8 << (2 * p))].view(np.uint64)
    return p, lut.copy(), suf.copy()


def order_database_on_device(ctx, hparams, bins, p_out):
    """bins: [(record bytes, LUT counts)] -> (records bytes, LUT) of kmc_hip_order_database_device"""
    rb = ctx.out_rec_bytes(hparams)
    n = len(bins)
    descs = (capi.BinDesc * n)()
    allocs = []
    total = 0
    for i, (recs, lut) in enumerate(bins):
        d_out, d_lut, d_small = ctx.malloc(recs.size + 256), ctx.malloc(lut.nbytes), ctx.malloc(64)
        if recs.size:
            ctx.h2d(d_out, recs)
        ctx.h2d(d_lut, lut)
        ctx.h2d(d_small, np.array([0, 0, 0, 0, recs.size, 0, 0, 0], dtype=np.uint64))
        allocs += [d_out, d_lut, d_small]
        descs[i] = capi.BinDesc(0, 0, 0, 0, 0, d_out, recs.size, d_small + 32, d_lut, d_small)
        total += recs.size // rb
    rb_out = (hparams.kmer_len - p_out) // 4 + (rb - (hparams.kmer_len - hparams.lut_prefix_len) // 4)
    d_res, d_lut_out = ctx.malloc(total * rb_out + 256), ctx.malloc(8 << (2 * p_out))
    got_n = ctx.order_database_device(hparams, descs, p_out, d_res, total * rb_out, d_lut_out)
    out = np.zeros(got_n * rb_out, dtype=np.uint8)
    lut = np.zeros(1 << (2 * p_out), dtype=np.uint64)
    if out.size:
        ctx.d2h(out, d_res)
    ctx.d2h(lut, d_lut_out)
    for a in allocs + [d_res, d_lut_out]:
        ctx.free(a)
    return out, lut, got_n


@pytest.mark.parametrize("flags", [["-k27"], ["-k55", "-ci1", "-cs1000"], ["-k21", "-b"]], ids=lambda f: "".join(f))
def test_order_database_matches_kmc_tools_transform_sort(ctx, flags, ref_bins, tmp_path):
    """kmc's database (ordered inside every signature bin) -> one ascending sequence on the device == what the reference's `kmc_tools transform db sort out`
    writes, records and LUT byte for byte."""
    if ref_bins is None:
        pytest.skip("oracle/_ref not shipped")
    fq = str(tmp_path / "in.fq")
    from kmc_amd import synth

    small = capi.load().kmc_hip_backend_kind() != 0  # the emulated host library of the CPU suite: one OS thread per GPU thread
    synth.make_fastq(fq, seed=13, genome_len=60_000 if small else 300_000, n_reads=8_000 if small else 40_000, read_len=150)
    (tmp_path / "t").mkdir()
    db, sdb = str(tmp_path / "db"), str(tmp_path / "sorted")
    subprocess.run([ref_bins["kmc"], *flags, "-sr1", fq, db, str(tmp_path / "t")], check=True, capture_output=True)
    subprocess.run([ref_bins["kmc_tools"], "transform", db, "sort", sdb], check=True, capture_output=True)
    k, p, cs, bins = read_kmc2_database(db)
    p_out, want_lut, want_recs = read_kmc1_database(sdb)
    hparams = hp(k, lut_prefix_len=p, both_strands=0 if "-b" in flags else 1, cutoff_min=1 if "-ci1" in flags else 2, counter_max=1000 if "-cs1000" in flags else 255)
    assert ctx.out_rec_bytes(hparams) == (k - p) // 4 + cs
    out, lut, n = order_database_on_device(ctx, hparams, bins, p_out)
    assert n == sum(b[0].size for b in bins) // ((k - p) // 4 + cs) and n > 500
    assert np.array_equal(lut, want_lut), _first_diff(lut, want_lut)
    assert np.array_equal(out, want_recs), _first_diff(out, want_recs)


ARENA_CASES = [
    # (KMC_SYNTH_REPEATS, k, params, giant buckets expected): buckets of copies x 30 records
    ("3000:20:5", 27, dict(lut_prefix_len=3), False),                              # ~600: dense buckets that fit a tile (kind 1: sorted in the arena, written back, ranked in place)
    ("2000:60:0,3000:13:0", 27, dict(lut_prefix_len=3), False),                    # ~1800 and ~390 (just beyond BR_MID)
    ("1000:150:10", 27, dict(lut_prefix_len=7, cutoff_min=1), False),              # ~4500: most of a tile
    ("2000:300:10", 27, dict(lut_prefix_len=3), False),                            # up to ~9000 (the copies diverge: most buckets still fit a tile)
    ("300:1500:5,5000:6:0", 27, dict(lut_prefix_len=3, cutoff_max=3000, counter_max=255), True),  # ~45 000: giant buckets (kind 0: counted from the arena, segment by segment), clamped counters, a cutoff_max inside the giant runs
    ("171:40000:20,H20000", 27, dict(lut_prefix_len=3), True),                     # a satellite (> 2^20 copies of its k-mers: no bin comes back) and a poly-A run
    ("300:1500:5,3000:20:5", 27, dict(lut_prefix_len=0, output_type=1), True),    # KFF records (no LUT: the suffix reaches up to the k-mer's top)
    ("300:1500:5,3000:20:5", 27, dict(lut_prefix_len=3, without_output=1), True), # tallies only
    ("300:1500:5,3000:20:5", 27, dict(lut_prefix_len=3, both_strands=0), True),   # forward strand only
    ("300:1500:5,3000:20:5", 32, dict(lut_prefix_len=4), True),                   # 32 key bits below the passes: a 5-byte arena key, 64-bit pairs in the tiles
    ("300:1500:5,3000:20:5", 25, dict(lut_prefix_len=1), True),                   # 6 spare bits: 16 bins per group, the tag rides in the bucket number
    ("300:1500:5,3000:20:5", 21, dict(lut_prefix_len=1), True),                   # 42-bit keys: 10 bits below the four passes
]


@pytest.mark.parametrize("repeats,k,kw,giant", ARENA_CASES, ids=lambda v: str(v).replace(" ", "") if not isinstance(v, dict) else "-".join(f"{a}{b}" for a, b in v.items()))
def test_arena_sorts_the_repeat_rich_buckets_of_one_word_records(repeats, k, kw, giant, monkeypatch):
    """Round 6 (arena_sort.hip.h): every bucket beyond BR_MID records of a group of one-word records — found by k_bucket_detect's samples — is sorted by HBM passes of its own over
    (entry ordinal, key bits below the bucket bits); buckets that fit a tile come back in place before k_bucket_rank runs, giant ones are counted from the arena. Per bin
    against the oracle; no group and no bin may come back for LSD passes, whatever the length of a bucket."""
    monkeypatch.setenv("KMC_SYNTH_REPEATS", repeats)
    bins = capi.synth_bins(seed=3, genome_len=200_000, n_reads=40_000, k=k, n_bins=16 if k == 25 else 4, n_threads=4)
    monkeypatch.delenv("KMC_SYNTH_REPEATS")
    ctx = capi.Context((0,))
    try:
        p = capi.make_params(k, **kw)
        op_ = O.make_params(p.kmer_len, p.both_strands, p.cutoff_min, p.cutoff_max, p.counter_max, p.lut_prefix_len, p.output_type, p.without_output)
        t0, c0 = ctx.local_sort_totals(), ctx.path_counters()
        got, err = _run_batch(ctx, p, bins, 1)
        assert err is None, err
        for i, (img, nrec, packs, _) in enumerate(bins):
            w = O.process_bin(op_, img, nrec)
            assert np.array_equal(got[i][2], w[2]), (repeats, i, got[i][2], w[2])
            assert np.array_equal(got[i][0], w[0]) and np.array_equal(got[i][1], w[1]), (repeats, i)
        t1, c1 = ctx.local_sort_totals(), ctx.path_counters()
        assert t1["redo_groups"] == t0["redo_groups"] and c1["rank_count"] > c0["rank_count"] and c1["lsd"] == c0["lsd"], (t0, t1, c0, c1)
        if giant:
            assert c1["giant_tiles"] > c0["giant_tiles"], (c0, c1)
    finally:
        ctx.close()


def test_arena_that_cannot_take_its_buckets_sends_the_group_back_and_the_old_finisher_stays_selectable():
    """KMC_HIP_ARENA_CAP=3 (read once per process): the list of buckets overflows, k_arena_plan drops the plan, k_bucket_rank reports nothing for the group and the host runs it
    again with LSD passes over every byte — results equal to the oracle's. KMC_HIP_ARENA=0: round 5's finisher (big buckets walked by the whole workgroup, k_giant_tiles)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, numpy as np; sys.path.insert(0, 'tests');"
            "import oracle_py as O; from kmc_amd import capi; from test_gpu_parity import _run_batch;"
            "ctx = capi.Context((0,));"
            "os.environ['KMC_SYNTH_REPEATS'] = '300:400:100,2000:60:10,2000:300:10,H3000';"
            "bins = capi.synth_bins(seed=9, genome_len=400_000, n_reads=120_000, k=27, n_bins=8);"
            "p = capi.make_params(27, lut_prefix_len=3); got, err = _run_batch(ctx, p, bins, 1); assert err is None, err;"
            "ok = all(all(np.array_equal(a, b) for a, b in zip(got[i], O.process_bin(O.make_params(27, lut_prefix_len=3), img, nrec))) for i, (img, nrec, pk, _) in enumerate(bins));"
            "print('repeats', ok, 'redo', ctx.local_sort_totals()['redo_groups'], ctx.path_counters())")
    for env, redo in (({"KMC_HIP_ARENA_CAP": "3"}, True), ({"KMC_HIP_ARENA": "0"}, False)):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "repeats True" in r.stdout, (env, (r.stdout + r.stderr)[-1500:])
        assert (int(r.stdout.split("redo ")[1].split()[0]) > 0) == redo, (env, r.stdout[-400:])


@pytest.mark.parametrize("repeats", ["5000:6:0", "5000:9:0", "1000:100:10", "2000:40:5,300:230:0,H2000"])
def test_rank_path_buckets_of_a_few_hundred_to_a_few_thousand_records(repeats, monkeypatch):
    """Repeat families make buckets of copies x coverage records: 180 (one-word pairs of 32 bits), 270 (64-bit pairs), 3000 (several records per thread), and a mixture with a
    giant bucket and a poly-A run — the buckets beyond BR_BIG whose records k_bucket_rank deals out again over the whole workgroup. Dense: a dozen of them per tile. The first
    version of that path passed every other test of this file and the emulations and was wrong exactly here (profiles/r05/experiments/README.md 5)."""
    monkeypatch.setenv("KMC_SYNTH_REPEATS", repeats)
    bins = capi.synth_bins(seed=3, genome_len=200_000, n_reads=40_000, k=27, n_bins=4, n_threads=4)
    monkeypatch.delenv("KMC_SYNTH_REPEATS")
    ctx = capi.Context((0,))
    try:
        for k_, kw in ((27, dict(lut_prefix_len=3)),):
            p = capi.make_params(k_, **kw)
            op = O.make_params(p.kmer_len, p.both_strands, p.cutoff_min, p.cutoff_max, p.counter_max, p.lut_prefix_len, p.output_type, p.without_output)
            got, err = _run_batch(ctx, p, bins, 1)
            assert err is None, err
            for i, (img, nrec, packs, _) in enumerate(bins):
                w = O.process_bin(op, img, nrec)
                assert np.array_equal(got[i][2], w[2]), (repeats, i, got[i][2], w[2])
                assert np.array_equal(got[i][0], w[0]) and np.array_equal(got[i][1], w[1]), (repeats, i)
    finally:
        ctx.close()


def test_rank_path_big_buckets_of_two_word_records(monkeypatch):
    """the same at k = 55 (A/B pairs, the indirect sort's gathered records): buckets of ~200 and ~3000 two-word records"""
    monkeypatch.setenv("KMC_SYNTH_REPEATS", "4000:7:0,1000:100:5")
    bins = capi.synth_bins(seed=4, genome_len=200_000, n_reads=40_000, k=55, n_bins=4, n_threads=4)
    monkeypatch.delenv("KMC_SYNTH_REPEATS")
    ctx = capi.Context((0,))
    try:
        p = capi.make_params(55, lut_prefix_len=3)
        op = O.make_params(55, lut_prefix_len=3)
        got, err = _run_batch(ctx, p, bins, 1)
        assert err is None, err
        for i, (img, nrec, packs, _) in enumerate(bins):
            w = O.process_bin(op, img, nrec)
            assert all(np.array_equal(a, b) for a, b in zip(got[i], w)), i
    finally:
        ctx.close()

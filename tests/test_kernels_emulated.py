"""The kernel SOURCE of kmc_amd/csrc/kernels.hip.h executed on the CPU (tests/hipemu: one OS thread per GPU thread, wave64 cross-lane
operations through per-wave exchanges) and compared bit for bit with the oracle. This is how kernel logic is checked in a container
without a GPU before a GPU minute is spent; the `-m gpu` suite remains the parity test of the product (libkmc_hip.so on gfx950).
k_onesweep's ranking relies on a wave executing an LDS instruction for all its lanes at once; the one place is marked KMC_WAVE_LOCKSTEP()
(nothing on the device — the ISA is byte-identical with and without it — a wave rendezvous here).
All cases but the last use a build of the same source with 128-thread workgroups (tests/emu.py GEOMETRY_FLAGS), which emulates ~10x faster."""
import ctypes as C

import numpy as np
import pytest

import binsynth
import emu
import oracle_py as O


def _oracle_compact(p, srt):
    n, words = srt.shape
    out = np.zeros(n * (8 * words + 8) + 8, dtype=np.uint8)
    lut = np.zeros(max(O.lut_entries(p), 1), dtype=np.uint64)
    st = np.zeros(4, dtype=np.uint64)
    ob = C.c_uint64()
    srt = np.ascontiguousarray(srt)
    rc = O.lib().oracle_compact(C.byref(p), srt.ctypes.data_as(C.POINTER(C.c_uint64)), n, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size,
                                C.byref(ob), lut.ctypes.data_as(C.POINTER(C.c_uint64)), st.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 0
    return out[: ob.value], lut[: O.lut_entries(p)], st


def _check_compact(p, srt):
    r = emu.run(p, 4, recs=srt)
    w_out, w_lut, w_st = _oracle_compact(p, srt)
    assert r["err"] == 0
    assert np.array_equal(r["stats"], w_st), (r["stats"], w_st)
    assert np.array_equal(r["out"], w_out)
    if p.output_type == 0 and p.lut_prefix_len:
        assert np.array_equal(r["lut"], w_lut)


@pytest.mark.parametrize("k,pl", [(27, 3), (27, 7), (55, 3), (127, 3), (32, 4), (64, 0), (200, 3)])
def test_emulated_compaction_matches_oracle(k, pl):
    rng = np.random.default_rng(k + pl)
    g = rng.integers(0, 4, size=8_000, dtype=np.uint8)
    img, nk, _ = binsynth.random_bin(rng, k, 1200, max_extra=60, genome=g)
    for kw in (dict(cutoff_min=2), dict(cutoff_min=1, counter_max=3), dict(cutoff_min=1, cutoff_max=4), dict(cutoff_min=3, counter_max=70000)):
        p = O.make_params(k, lut_prefix_len=pl, output_type=0 if pl else 1, **kw)
        _check_compact(p, O.sort(O.expand(p, img)))


def test_emulated_compaction_runs_longer_than_tiles_and_waves():
    """runs that cross row, wave and tile boundaries, and runs longer than the 64 records inspected below a tile (64-ary search)"""
    reps = [1, 700, 2, 9000, 1, 1, 20_000, 3, 4097, 4096, 4095, 1, 63, 64, 65, 1023, 1024, 1025, 1]
    vals = np.sort(np.random.default_rng(5).choice(2**40, size=len(reps), replace=False).astype(np.uint64)) << np.uint64(10)
    srt = np.repeat(vals, reps).reshape(-1, 1)
    for kw in (dict(cutoff_min=2), dict(cutoff_min=1, counter_max=255), dict(cutoff_min=1, cutoff_max=4096, counter_max=10**6), dict(cutoff_min=65, cutoff_max=9000)):
        _check_compact(O.make_params(27, lut_prefix_len=3, **kw), srt)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 8193])
def test_emulated_compaction_edge_sizes(n):
    rng = np.random.default_rng(n)
    srt = np.sort(rng.integers(0, max(2, n // 3), size=n).astype(np.uint64) << np.uint64(17)).reshape(-1, 1)
    for kw in (dict(cutoff_min=1), dict(cutoff_min=2, counter_max=2)):
        _check_compact(O.make_params(27, lut_prefix_len=7, **kw), srt)
    srt2 = np.repeat(srt, 2, axis=1)  # two-word records (k = 55): both words take part in the comparison
    srt2[:, 0] = np.arange(n, dtype=np.uint64) // 2  # low word: pairs
    srt2[:, 1] = srt2[:, 1] >> np.uint64(30)
    order = np.lexsort((srt2[:, 0], srt2[:, 1]))
    _check_compact(O.make_params(55, lut_prefix_len=3, cutoff_min=1), np.ascontiguousarray(srt2[order]))


def test_emulated_compaction_without_output_and_capacity_error():
    rng = np.random.default_rng(9)
    srt = np.sort(rng.integers(0, 3000, size=10_000).astype(np.uint64) << np.uint64(20)).reshape(-1, 1)
    p = O.make_params(27, lut_prefix_len=3, cutoff_min=1, without_output=1)
    r = emu.run(p, 4, recs=srt)
    _, _, w_st = _oracle_compact(O.make_params(27, lut_prefix_len=3, cutoff_min=1), srt)
    assert r["err"] == 0 and r["out"].size == 0 and np.array_equal(r["stats"], w_st)
    r = emu.run(O.make_params(27, lut_prefix_len=3, cutoff_min=1), 4, recs=srt, out_capacity=1000)
    assert r["err"] & 4  # KERR_CAPACITY


@pytest.mark.parametrize("k,both", [(27, 1), (27, 0), (55, 1), (14, 1), (127, 1)])
def test_emulated_parse_and_expand_match_oracle(k, both):
    rng = np.random.default_rng(k + both)
    img, nk, packs = binsynth.random_bin(rng, k, 2500, max_extra=120, pack_size=257)
    p = O.make_params(k, both_strands=both, lut_prefix_len=0, output_type=1)
    r = emu.run(p, 1, img, nk, packs)
    assert r["err"] == 0
    assert np.array_equal(r["recs"], O.expand(p, img))


def test_emulated_front_end_and_compaction_around_the_oracle_sort():
    """parse + expand (emulated) -> sort (oracle) -> compaction (emulated) == oracle_process_bin"""
    rng = np.random.default_rng(77)
    g = rng.integers(0, 4, size=6_000, dtype=np.uint8)
    img, nk, packs = binsynth.random_bin(rng, 27, 1500, max_extra=40, genome=g, pack_size=300)
    p = O.make_params(27, lut_prefix_len=3)
    recs = emu.run(p, 1, img, nk, packs)["recs"]
    r = emu.run(p, 4, recs=O.sort(recs))
    w_out, w_lut, w_st = O.process_bin(p, img, nk)
    assert np.array_equal(r["out"], w_out) and np.array_equal(r["lut"], w_lut) and np.array_equal(r["stats"], w_st)


@pytest.mark.parametrize("k,max_extra,pack_size,n_super", [(27, 255, 37, 1500), (27, 40, 10**9, 6000), (14, 255, 4096, 3000), (256, 255, 500, 800), (31, 0, 4096, 5000)])
def test_emulated_parse_deep_entries_and_pack_shapes(k, max_extra, pack_size, n_super):
    """records longer than the speculated entry window (PARSE_CAND positions per 128-byte sub-block) take the exact slow path of the
    chain walk; single huge packs cross many parse chunks; e = 0 everywhere gives the shortest records"""
    rng = np.random.default_rng(k + max_extra + n_super)
    img, nk, packs = binsynth.random_bin(rng, k, n_super, max_extra=max_extra, pack_size=pack_size)
    p = O.make_params(k, lut_prefix_len=0, output_type=1)
    r = emu.run(p, 1, img, nk, packs)
    assert r["err"] == 0
    assert np.array_equal(r["recs"], O.expand(p, img))


@pytest.mark.parametrize("tile,back", [(0, 0), (1, 1), (5, 5), (64, 1), (64, 64), (65, 65), (700, 3), (700, 63), (700, 64), (700, 65), (700, 350),
                                       (700, 511), (700, 512), (700, 513), (700, 700), (5000, 1500)])
def test_emulated_lookback_walks_several_windows_per_round_trip(tile, back):
    """the 64-bit decoupled look-back fetches several 64-tile windows per round trip: the nearest inclusive prefix `back` tiles behind,
    aggregates in between, unpublished words (zero) beyond the prefix must not matter; before tile 0 lies a virtual empty prefix"""
    AGG, PREFIX = 1 << 62, 2 << 62
    rng = np.random.default_rng(tile * 1000 + back)
    status = np.zeros(tile + 1, dtype=np.uint64)
    agg = rng.integers(0, 5000, size=tile + 1)
    want = 0
    if tile:
        p = tile - back  # the tile that holds a prefix (p < 0: none, the walk reaches the virtual prefix before tile 0)
        if p >= 0:
            incl = int(rng.integers(0, 1 << 40))
            status[p] = PREFIX | incl
            want = incl
        for t in range(max(p + 1, 0), tile):
            status[t] = AGG | int(agg[t])
            want += int(agg[t])
    got, err = emu.lookback(status, tile, 1234)
    assert err == 0 and got == want
    assert int(status[tile]) == PREFIX | (want + 1234)


def test_emulated_kernels_with_the_product_tile_geometry():
    """the cases above run a build with 128-thread workgroups (same source, ~10x faster to emulate); this one runs the product's geometry
    (512-thread workgroups, 16 KB expand slices) once through the front end and the compaction"""
    rng = np.random.default_rng(2026)
    g = rng.integers(0, 4, size=3_000, dtype=np.uint8)
    img, nk, packs = binsynth.random_bin(rng, 27, 500, max_extra=30, genome=g, pack_size=120)
    p = O.make_params(27, lut_prefix_len=3)
    recs = emu.run(p, 1, img, nk, packs, geometry="product")
    assert recs["err"] == 0 and np.array_equal(recs["recs"], O.expand(p, img))
    r = emu.run(p, 4, recs=O.sort(recs["recs"]), geometry="product")
    w_out, w_lut, w_st = O.process_bin(p, img, nk)
    assert np.array_equal(r["out"], w_out) and np.array_equal(r["lut"], w_lut) and np.array_equal(r["stats"], w_st)


@pytest.mark.parametrize("k,pl,g", [(27, 3, 4), (27, 7, 3), (25, 5, 9), (55, 3, 4), (27, 3, 16)])
def test_emulated_group_of_bins_shares_one_record_array(k, pl, g):
    """the grouped path of kmc_hip.hip: g bins parsed and expanded by ONE launch each into one record array, the bin's number inside the
    group above the k-mer (in the spare bits of the top radix digit, or in a digit of its own: g = 9 and 16 at k = 27/25); after a sort
    on the whole key the array is bin-major, and ONE compaction launch + one fold launch give every bin's records, LUT and tallies."""
    rng = np.random.default_rng(k * 100 + g)
    genome = rng.integers(0, 4, size=5_000, dtype=np.uint8)
    bins = [binsynth.random_bin(rng, k, int(rng.integers(1, 400)), max_extra=40, genome=genome, pack_size=97) for _ in range(g)]
    p = O.make_params(k, lut_prefix_len=pl)
    tag_bits = max(1, (g - 1).bit_length())
    n_pass = (2 * k + tag_bits + 7) // 8
    err, recs = emu.group_front(p, bins, n_pass)
    assert err == 0
    words = recs.shape[1]
    off = 0
    want_sorted = []
    for i, (img, nk, _) in enumerate(bins):
        w = O.expand(p, img).copy()
        w[:, (2 * k) // 64] |= np.uint64(i) << np.uint64((2 * k) % 64)  # the tag the kernel puts above the k-mer
        assert np.array_equal(recs[off:off + nk], w), i
        want_sorted.append(O.sort(w))
        off += nk
    srt = np.concatenate(want_sorted)  # bin-major = ascending on the tagged key
    if words == 1:
        assert np.array_equal(srt[:, 0], np.sort(recs[:, 0]))
    err, got = emu.group_compact(p, srt, [b[1] for b in bins])
    assert err == 0
    for i, (img, nk, _) in enumerate(bins):
        w_out, w_lut, w_st = O.process_bin(p, img, nk)
        assert np.array_equal(got[i][2], w_st), (i, got[i][2], w_st)
        assert np.array_equal(got[i][0], w_out), i
        assert np.array_equal(got[i][1], w_lut), i


@pytest.mark.parametrize("words,key_bytes,n", [(1, 7, 1), (1, 7, 2), (1, 7, 2559), (1, 7, 2560), (1, 7, 2561), (1, 8, 9000), (2, 14, 3000), (4, 32, 1500)])
def test_emulated_onesweep_sort_matches_oracle(words, key_bytes, n):
    """the 8-bit LSD passes (histograms, digit bases, one onesweep launch per pass with its decoupled look-back) on tiles of 256 threads"""
    rng = np.random.default_rng(words * 1000 + n)
    recs = rng.integers(0, 1 << 62, size=(n, words), dtype=np.uint64)
    if key_bytes < 8 * words:  # bytes above the key are zero by construction (kb_sorter.h:757-780)
        full, rem = divmod(key_bytes, 8)
        recs[:, full + (1 if rem else 0):] = 0
        if rem:
            recs[:, full] &= np.uint64((1 << (8 * rem)) - 1)
    k = {7: 27, 8: 32, 14: 55, 32: 127}[key_bytes]
    p = O.make_params(k, lut_prefix_len=0, output_type=1)
    r = emu.run(p, 2, recs=recs)
    assert r["err"] == 0
    assert np.array_equal(r["sorted"], O.sort(recs))


def test_emulated_whole_bin_matches_oracle():
    """parse -> expand (+ fused histograms, digit bases) -> 7 onesweep passes -> compaction -> fold, all emulated, against oracle_process_bin"""
    rng = np.random.default_rng(4242)
    g = rng.integers(0, 4, size=4_000, dtype=np.uint8)
    img, nk, packs = binsynth.random_bin(rng, 27, 700, max_extra=30, genome=g, pack_size=150)
    p = O.make_params(27, lut_prefix_len=3)
    r = emu.run(p, 7, img, nk, packs)
    w_out, w_lut, w_st = O.process_bin(p, img, nk)
    assert r["err"] == 0
    assert np.array_equal(r["out"], w_out) and np.array_equal(r["lut"], w_lut) and np.array_equal(r["stats"], w_st)


def test_emulated_group_with_tiny_bins():
    """bins of one super-k-mer (one k-mer, or a handful) next to a normal one in the same group: one-slice bins, one-tile bins, runs of one"""
    rng = np.random.default_rng(99)
    k = 27
    bins = [binsynth.random_bin(rng, k, 1, max_extra=0), binsynth.random_bin(rng, k, 1, max_extra=5), binsynth.random_bin(rng, k, 250, max_extra=40),
            binsynth.random_bin(rng, k, 2, max_extra=255)]
    p = O.make_params(k, lut_prefix_len=3, cutoff_min=1)
    err, recs = emu.group_front(p, bins, (2 * k + 2 + 7) // 8)
    assert err == 0
    parts, off = [], 0
    for i, (img, nk, _) in enumerate(bins):
        w = O.expand(p, img).copy()
        w[:, 0] |= np.uint64(i) << np.uint64(2 * k)
        assert np.array_equal(recs[off:off + nk], w), i
        parts.append(O.sort(w))
        off += nk
    err, got = emu.group_compact(p, np.concatenate(parts), [b[1] for b in bins])
    assert err == 0
    for i, (img, nk, _) in enumerate(bins):
        w_out, w_lut, w_st = O.process_bin(p, img, nk)
        assert np.array_equal(got[i][2], w_st) and np.array_equal(got[i][0], w_out) and np.array_equal(got[i][1], w_lut), i


def _valid_prefix_lens(k):
    return [p for p in range(0, min(k, 8)) if p == 0 or ((k - p) % 4 == 0 and p < k)]


@pytest.mark.parametrize("seed", range(24))
def test_emulated_whole_bin_random_parameters(seed):
    """a seeded sweep over the parameter space of CKMCParams that reaches the kernels: k (1-4 record words), strands, cutoffs, counter width,
    LUT prefix length / KFF, output on or off — every stage emulated, the result against oracle_process_bin. Both output modes of the
    compaction occur (the two-phase one whenever a tile's span is certain to hold its records)."""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([9, 14, 21, 25, 27, 28, 31, 32, 33, 40, 55, 64, 70, 100, 127]))
    both = int(rng.integers(0, 2))
    pls = _valid_prefix_lens(k)
    pl = int(rng.choice(pls))
    kff = 1 if pl == 0 else 0
    cutoff_min = int(rng.choice([1, 1, 2, 3, 5]))
    cutoff_max = int(rng.choice([10**9, 10**9, 6, 40]))
    counter_max = int(rng.choice([255, 255, 1, 3, 70000, 2**32 - 1]))
    without_output = int(rng.random() < 0.1)
    genome = rng.integers(0, 4, size=int(rng.integers(600, 4000)), dtype=np.uint8)
    img, nk, packs = binsynth.random_bin(rng, k, int(rng.integers(1, 500)), max_extra=int(rng.choice([0, 10, 60, 255])), genome=genome if genome.size > k + 260 else None,
                                         pack_size=int(rng.choice([1, 50, 4096])))
    p = O.make_params(k, both_strands=both, cutoff_min=cutoff_min, cutoff_max=max(cutoff_max, cutoff_min), counter_max=counter_max, lut_prefix_len=pl,
                      output_type=kff, without_output=without_output)
    r = emu.run(p, 7, img, nk, packs)
    assert r["err"] == 0
    w_out, w_lut, w_st = O.process_bin(p, img, nk)
    assert np.array_equal(r["stats"], w_st), (k, both, pl, cutoff_min, cutoff_max, counter_max, r["stats"], w_st)
    if not without_output:
        assert np.array_equal(r["out"], w_out), (k, both, pl, cutoff_min, cutoff_max, counter_max)
        if not kff:
            assert np.array_equal(r["lut"], w_lut)


def test_emulated_two_phase_output_over_several_fold_chunks():
    """eight-word records make 256-record tiles in the small geometry: 330 tiles, i.e. two rounds of the fold's scan of tile counts (256 per
    round there), then the gather of every tile's records to its place"""
    rng = np.random.default_rng(31)
    n = 330 * 256 - 17
    vals = np.sort(rng.integers(0, 1 << 40, size=n // 3).astype(np.uint64))
    keys = np.sort(rng.choice(vals, size=n))
    srt = np.zeros((n, 8), dtype=np.uint64)
    srt[:, 0] = keys << np.uint64(8)
    srt[:, 7] = keys >> np.uint64(20)  # both the lowest and the highest word take part in the comparison; order stays ascending
    order = np.lexsort(tuple(srt[:, w] for w in range(8)))
    srt = np.ascontiguousarray(srt[order])
    p = O.make_params(240, lut_prefix_len=4, cutoff_min=2)
    _check_compact(p, srt)

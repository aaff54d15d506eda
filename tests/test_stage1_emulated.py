"""First stage-1 kernels (kmc_amd/csrc/stage1_kernels.hip.h: minimizer signature per k-mer, super-k-mer cutting) executed on the CPU under
tests/hipemu and compared with the stage-1 oracle, which tests/test_stage1_oracle.py pins to the reference. Groundwork for SURVEY.md §8f
rank 2; the kernels are not part of the drop-in yet."""
import numpy as np
import pytest

import emu
import oracle_s1 as S1


def _stream(reads):
    codes, off = S1.encode(reads)
    parts = []
    for i in range(len(reads)):
        parts.append(codes[int(off[i]):int(off[i + 1])])
        parts.append(np.array([-1], dtype=np.int8))  # read boundary = an invalid symbol
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int8)


def _reads(rng, k, n_reads, read_len):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rnd = lambda n: acgt[rng.integers(0, 4, size=n)].tobytes()
    per = rnd(11)
    reads = [rnd(int(rng.integers(max(1, read_len - 30), read_len + 30))) for _ in range(n_reads)]
    reads += [rnd(150) + b"N" + rnd(80) + b"NN" + rnd(k - 1) + b"N" + rnd(k) + b"n" + rnd(200), (per * 120)[:1100], rnd(k), rnd(k - 1), rnd(k + 1), b"N" * 40,
              b"A" * 700, b"AC" * 300, b"T" * (k + 300), rnd(3000)]
    return reads


@pytest.mark.parametrize("fused,geometry", [(True, "small"), (False, "small"), (True, "product")])
@pytest.mark.parametrize("k,m", [(27, 9), (21, 9), (55, 9), (14, 7), (200, 9), (28, 11), (256, 11), (9, 9), (12, 9), (13, 9), (256, 5)])
def test_emulated_stage1_signatures_and_cut_match_the_oracle(k, m, fused, geometry):
    """small geometry: two 1024-position tiles per cutting workgroup, product: four"""
    rng = np.random.default_rng(k * 10 + m)
    codes = _stream(_reads(rng, k, 40, 150))
    err, sig, pos, ln, sg = emu.s1_split(codes, k, m, fused=fused, geometry=geometry)
    assert err == 0
    w_pos, w_len, w_sig = S1.split_stream(codes, k, m)
    assert pos.size == w_pos.size, (pos.size, w_pos.size)
    assert np.array_equal(pos, w_pos.astype(np.uint64)) and np.array_equal(ln, w_len) and np.array_equal(sg, w_sig)
    # every valid k-mer position lies in exactly one super-k-mer, and carries that super-k-mer's signature
    covered = np.zeros(codes.size, dtype=np.int32)
    for p, l, s in zip(w_pos, w_len, w_sig):
        q = np.arange(int(p), int(p) + int(l) - k + 1)
        covered[q] += 1
        assert np.all(sig[q] == s)
    assert np.array_equal(covered == 1, sig != 0xFFFFFFFF) and covered.max() <= 1


def test_emulated_stage1_tile_boundaries():
    """runs of one signature far longer than a tile (1024 positions) and than the 256-k-mer cap, cut points next to tile boundaries"""
    k, m = 27, 9
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    per = acgt[rng.integers(0, 4, size=13)].tobytes()
    for pad in (0, 1, 1000, 1023, 1024, 1025, 2047, 2048, 4095, 4096, 4097, 8191):
        codes = _stream([acgt[rng.integers(0, 4, size=pad)].tobytes() if pad else b"", (per * 400)[:4000], b"A" * 3000])
        for fused, geometry in ((True, "small"), (False, "small"), (True, "product")):
            err, sig, pos, ln, sg = emu.s1_split(codes, k, m, fused=fused, geometry=geometry)
            w_pos, w_len, w_sig = S1.split_stream(codes, k, m)
            assert err == 0 and np.array_equal(pos, w_pos.astype(np.uint64)) and np.array_equal(ln, w_len) and np.array_equal(sg, w_sig), (pad, fused, geometry)


@pytest.mark.parametrize("m", [5, 6, 7, 8, 9, 10, 11])
def test_computed_mmer_normalisation_equals_the_reference_table(m):
    """the kernels compute norm(m-mer) with bit operations instead of reading CMmer's table: all 4^m values against the oracle's table"""
    assert np.array_equal(emu.s1_norm_all(m), S1.norm_table(m))


def _parse_bin(img, k):
    """bin byte stream -> sorted list of (symbol count, record bytes)"""
    out, p = [], 0
    while p < img.size:
        ln = int(img[p]) + k
        nb = 1 + (ln + 3) // 4
        out.append(bytes(img[p:p + nb]))
        p += nb
    assert p == img.size
    return sorted(out)


def _sig_map(m, n_bins, seed):
    """any signature -> bin map will do for the scatter (the reference builds its map from signature statistics, kmc.h:1138-1200)"""
    rng = np.random.default_rng(seed)
    return rng.integers(0, n_bins, size=(1 << (2 * m)) + 1).astype(np.int32)


@pytest.mark.parametrize("k,m,n_bins", [(27, 9, 64), (21, 7, 512), (55, 9, 2000), (14, 7, 3), (200, 9, 1)])
def test_emulated_stage1_bin_scatter_matches_the_oracle(k, m, n_bins):
    rng = np.random.default_rng(k + n_bins)
    reads = _reads(rng, k, 60, max(150, 2 * k))
    need_bytes = 4 * emu.s1_geometry()[1] * n_bins if n_bins <= 3 else 0  # few bins: several packs per bin
    while len(S1.split(reads, k, m)[0]) <= 1100 or S1.split(reads, k, m)[2].size < need_bytes:  # more than one tile of super-k-mers
        reads += _reads(rng, k, 60, max(150, 2 * k))
    codes = _stream(reads)
    err, _, pos, ln, sg = emu.s1_split(codes, k, m)
    assert err == 0 and pos.size > 1024
    smap = _sig_map(m, n_bins, 7)
    r = emu.s1_scatter(codes, pos, ln, sg, k, smap, n_bins)
    assert r["err"] == 0
    base, tot, out = r["base"], r["totals"], r["out"]
    _, pack_bytes, align = emu.s1_geometry()
    # the oracle's records (emission order), grouped by bin
    w_sig, w_off, w_recs = S1.split(reads, k, m)
    want = [[] for _ in range(n_bins)]
    for i, s in enumerate(w_sig):
        want[smap[s]].append(bytes(w_recs[int(w_off[i]):int(w_off[i + 1])]))
    assert int(tot[0].sum()) == w_recs.size
    multi_pack = 0
    for b in range(n_bins):
        lo, size = int(base[b]), int(tot[0, b])
        assert lo % align == 0 and lo + size <= int(base[b + 1])
        img = out[lo:lo + size]
        assert _parse_bin(img, k) == sorted(want[b]), b
        assert int(tot[1, b]) == len(want[b]) and int(tot[2, b]) == sum(int(x[0]) + 1 for x in want[b])
        # the bin's pack boundaries: increasing, first 0, last = size, every one a record start, no pack longer than two pack sizes
        ps = r["pack_start"][int(r["pack_base"][b]):int(r["pack_base"][b + 1])].astype(np.int64)
        assert ps.size == (size + pack_bytes - 1) // pack_bytes + 1 and ps[-1] == size and (size == 0 or ps[0] == 0)
        assert np.all(np.diff(ps) > 0) if size else ps.tolist() == [0]
        assert np.all(np.diff(ps) <= 2 * pack_bytes)
        starts, p = set(), 0
        while p < size:
            starts.add(p)
            p += 1 + (int(img[p]) + k + 3) // 4
        assert set(ps[:-1].tolist()) <= starts
        multi_pack += ps.size > 2
    if n_bins <= 3:
        assert multi_pack  # the pack-boundary rule was exercised across several packs


def test_emulated_stage1_bin_scatter_reports_an_unknown_signature():
    k, m = 27, 9
    rng = np.random.default_rng(1)
    codes = _stream(_reads(rng, k, 30, 150))
    err, _, pos, ln, sg = emu.s1_split(codes, k, m)
    smap = _sig_map(m, 16, 3)
    bad = smap.copy()
    bad[sg[5]] = -1
    assert emu.s1_scatter(codes, pos, ln, sg, k, bad, 16)["err"] & 1  # KERR_CORRUPT
    bad[sg[5]] = 16
    assert emu.s1_scatter(codes, pos, ln, sg, k, bad, 16)["err"] & 1


def test_emulated_reads_to_database_records_stage1_into_stage2():
    """the hand-over this groundwork is for: reads -> signatures -> super-k-mers -> bins + pack boundaries (stage-1 kernels) -> parse -> expand
    -> sort -> compaction (stage-2 kernels), every kernel emulated. Each bin's suffix records, LUT and tallies must equal what the stage-2
    oracle makes of the SAME bin as the stage-1 oracle (pinned to the reference's splitter) writes it, in read order."""
    import oracle_py as O

    k, m, n_bins = 27, 9, 4
    rng = np.random.default_rng(77)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=3000)].tobytes()
    reads = []
    for _ in range(500):  # reads of a small genome: repeated k-mers, both strands
        a = int(rng.integers(0, len(genome) - 160))
        r = genome[a:a + int(rng.integers(60, 160))]
        if rng.random() < 0.5:
            r = r[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
        reads.append(r)
    reads += [genome[:40] + b"N" + genome[40:90], b"A" * 400]
    codes = _stream(reads)
    err, _, pos, ln, sg = emu.s1_split(codes, k, m)
    assert err == 0
    smap = _sig_map(m, n_bins, 11)
    r = emu.s1_scatter(codes, pos, ln, sg, k, smap, n_bins)
    assert r["err"] == 0
    w_sig, w_off, w_recs = S1.split(reads, k, m)
    p = O.make_params(k, lut_prefix_len=3, cutoff_min=2)
    total = np.zeros(4, dtype=np.uint64)
    for b in range(n_bins):
        lo, size, nk = int(r["base"][b]), int(r["totals"][0, b]), int(r["totals"][2, b])
        ps = r["pack_start"][int(r["pack_base"][b]):int(r["pack_base"][b + 1])]
        got = emu.run(p, 7, r["out"][lo:lo + size], nk, np.diff(ps.astype(np.int64)).astype(np.uint64))
        assert got["err"] == 0
        ref_img = np.concatenate([w_recs[int(w_off[i]):int(w_off[i + 1])] for i in np.nonzero(smap[w_sig] == b)[0]])
        w_out, w_lut, w_st = O.process_bin(p, ref_img, nk)
        assert np.array_equal(got["out"], w_out) and np.array_equal(got["lut"], w_lut) and np.array_equal(got["stats"], w_st), b
        total += w_st
    assert total[0] > total[1] > 0  # repeated k-mers were counted, and some passed the cutoff


def test_emulated_stage1_cut_over_more_than_64_workgroup_tiles():
    """both look-backs of the cutting kernel walk more than one window of 64 status words: 70 workgroup tiles, among them a run of one
    signature that spans several tiles (poly-A) right where the second window starts"""
    k, m = 27, 9
    rng = np.random.default_rng(12)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = [acgt[rng.integers(0, 4, size=150)].tobytes() for _ in range(840)] + [b"A" * 9000] + [acgt[rng.integers(0, 4, size=150)].tobytes() for _ in range(60)]
    codes = _stream(reads)
    assert codes.size > 70 * 2048
    err, _, pos, ln, sg = emu.s1_split(codes, k, m, fused=True)
    w_pos, w_len, w_sig = S1.split_stream(codes, k, m)
    assert err == 0 and np.array_equal(pos, w_pos.astype(np.uint64)) and np.array_equal(ln, w_len) and np.array_equal(sg, w_sig)

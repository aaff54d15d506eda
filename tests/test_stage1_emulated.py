"""First stage-1 kernels (kmc_amd/csrc/stage1_kernels.hip.h: minimizer signature per k-mer, super-k-mer cutting) executed on the CPU under
tests/hipemu and compared with the stage-1 oracle, which tests/test_stage1_oracle.py pins to the reference. Groundwork for SURVEY.md §8f
rank 2; the kernels are not part of the drop-in yet."""
import ctypes as C

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


# ---- the kernels a HIP split engine still needs (kmc_amd/host/split_engine.h): text -> codes, k+x-mer sums. Emulation only so far.
def _records_text(fmt, eol, recs):
    out = []
    for i, r in enumerate(recs):
        if fmt == "fq":
            out.append(b"@r%d x" % i + eol + r + eol + b"+" + (b"r%d" % i if i % 3 == 0 else b"") + eol + bytes([33 + (j * 7 + i) % 60 for j in range(len(r))]) + eol)
        else:
            out.append(b">r%d x" % i + eol + r + eol)
    return b"".join(out)


def _want_stream(seqs):
    parts = []
    for s in seqs:
        parts += [s, np.array([-1], dtype=np.int8)]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int8)


@pytest.mark.parametrize("fmt,eol", [("fq", b"\n"), ("fq", b"\r\n"), ("fa", b"\n"), ("fa", b"\r\n")])
def test_emulated_text_to_codes_matches_the_reference_parser(fmt, eol):
    """the code stream of a part = the sequences CSplitter::GetSeq hands out (oracle_s1_parse_part, pinned by tests/test_stage1_plugin.py), each
    followed by one separator; quality bytes that look like titles ('@' at a line start) and '+' inside titles must not confuse the line count"""
    k = 27
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rnd = lambda n: acgt[rng.integers(0, 4, size=n)].tobytes()
    recs = [rnd(int(rng.integers(1, 400))) for _ in range(300)] + [b"", rnd(30) + b"N" + rnd(5) + b"nnacgtRYK", b"", b"A", rnd(5000), b"acgt" * 10]
    text = _records_text(fmt, eol, recs)
    assert len(text) > 3 * 4096  # several tiles
    err, codes, nl = emu.s1_text_to_codes(text, 4 if fmt == "fq" else 2)
    assert err == 0
    seqs, n_reads = S1.parse_part(text, 1 if fmt == "fq" else 0, k)
    assert n_reads == len(recs) and len(seqs) == len(recs)
    assert np.array_equal(codes, _want_stream(seqs))
    assert np.array_equal(nl, np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10).astype(np.uint64))
    if fmt == "fa":  # a FASTA part may end inside its last sequence line: same sequences, no separator behind the last one
        cut = text[: len(text) - len(eol)]
        err, codes2, _ = emu.s1_text_to_codes(cut, 2)
        assert err == 0 and np.array_equal(codes2, codes[:-1])
        seqs2, _ = S1.parse_part(cut, 0, k)
        assert all(np.array_equal(a, b) for a, b in zip(seqs2, seqs)) and len(seqs2) == len(seqs)


@pytest.mark.parametrize("what", ["blank line", "short quality", "no plus", "lone cr", "missing last eol", "no title marker", "control char"])
def test_emulated_text_check_flags_parts_outside_its_domain(what):
    """anything GetSeq treats by rules the line model does not reproduce must be recognised, so that the engine can give the part to the
    reference splitter"""
    good = b"@a\nACGTACGT\n+\nIIIIIIII\n@b\nACGT\n+\nIIII\n"
    bad = {"blank line": good.replace(b"@b", b"\n@b"), "short quality": good.replace(b"IIIIIIII", b"IIIIIII"), "no plus": good.replace(b"+\nIIII\n", b"-\nIIII\n"),
           "lone cr": good.replace(b"ACGT\n+", b"AC\rGT\n+"), "missing last eol": good[:-1], "no title marker": b"x" + good[1:],
           "control char": good.replace(b"@b\nACGT", b"@b\n\tACGT")}[what]
    assert emu.s1_text_to_codes(good, 4)[0] == 0
    assert emu.s1_text_to_codes(bad, 4)[0] & 0x1000  # S1_TEXT_BAD (its own bit: 0x10 is KERR_PEER)
    assert emu.s1_text_to_codes(b">a\nACGT\n>b\nAC", 2)[0] == 0 and emu.s1_text_to_codes(b">a\nACGT\n>b", 2)[0] & 0x1000


@pytest.mark.parametrize("k,max_x,both", [(27, 3, True), (27, 3, False), (21, 3, True), (55, 2, True), (40, 1, True), (27, 0, True), (14, 3, True)])
def test_emulated_kxmer_sums_match_the_oracle(k, max_x, both):
    """n_plus_x_recs per bin (k_s1_bin_plus_x) against oracle_s1_kxmer_recs summed over the bin's super-k-mers; the oracle function is pinned to
    the reference's collectors by tests/test_stage1_plugin.py::test_bin_descriptors_*"""
    m, n_bins = 7 if k < 20 else 9, 37
    rng = np.random.default_rng(k + max_x)
    reads = _reads(rng, k, 60, 150) + [b"ACGT" * 100, b"AT" * 150, b"A" * 400]  # palindromic stretches: k-mer equals its reverse complement
    while len(S1.split(reads, k, m)[0]) <= 1100:
        reads += _reads(rng, k, 60, 150)
    codes = _stream(reads)
    err, _, pos, ln, sg = emu.s1_split(codes, k, m)
    assert err == 0
    smap = _sig_map(m, n_bins, 4)
    got = emu.s1_plus_x(codes, pos, ln, sg, k, max_x, both, smap, n_bins)
    want = np.zeros(n_bins, dtype=np.uint64)
    for p, l, s in zip(pos, ln, sg):
        want[smap[s]] += S1.kxmer_recs(codes[int(p):int(p) + int(l)], k, max_x, both)
    assert np.array_equal(got, want) and (want.sum() > 0) == (max_x > 0)


@pytest.mark.parametrize("k,m,n_bins", [(27, 9, 64), (21, 7, 512), (55, 9, 2000), (14, 7, 3), (200, 9, 1)])
def test_emulated_emit_through_a_sort_writes_bins_in_read_order(k, m, n_bins):
    """k_s1_sort_keys + a stable sort by bin + k_s1_emit_sorted: every bin image is EXACTLY the oracle's — the super-k-mers of the bin's signatures in
    read order, which is what the reference's splitter writes with one thread — not just the same multiset; pack boundaries as for k_s1_emit"""
    rng = np.random.default_rng(k + n_bins)
    reads = _reads(rng, k, 60, max(150, 2 * k))
    need_bytes = 4 * emu.s1_geometry()[1] * n_bins if n_bins <= 3 else 0
    while len(S1.split(reads, k, m)[0]) <= 1100 or S1.split(reads, k, m)[2].size < need_bytes:
        reads += _reads(rng, k, 60, max(150, 2 * k))
    codes = _stream(reads)
    err, _, pos, ln, sg = emu.s1_split(codes, k, m)
    assert err == 0
    smap = _sig_map(m, n_bins, 7)
    r = emu.s1_scatter(codes, pos, ln, sg, k, smap, n_bins, through_sort=True)
    assert r["err"] == 0
    w_sig, w_off, w_recs = S1.split(reads, k, m)
    _, pack_bytes, align = emu.s1_geometry()
    bins_of = smap[w_sig]
    for b in range(n_bins):
        idx = np.flatnonzero(bins_of == b)
        want = np.concatenate([w_recs[int(w_off[i]):int(w_off[i + 1])] for i in idx]) if idx.size else np.zeros(0, dtype=np.uint8)
        lo, size = int(r["base"][b]), int(r["totals"][0, b])
        assert lo % align == 0 and size == want.size
        assert np.array_equal(r["out"][lo:lo + size], want), b
        ps = r["pack_start"][int(r["pack_base"][b]):int(r["pack_base"][b + 1])].astype(np.int64)
        assert ps.size == (size + pack_bytes - 1) // pack_bytes + 1 and ps[-1] == size and (size == 0 or ps[0] == 0)
        assert np.all(np.diff(ps) > 0) if size else ps.tolist() == [0]
        starts = np.concatenate([[0], np.cumsum([int(w_off[i + 1] - w_off[i]) for i in idx])]) if idx.size else np.zeros(1, dtype=np.int64)
        assert np.all(np.isin(ps, starts))


# ---- lines of mem_part_pmm_reads symbols or more, and the parts the reader labels ReadType::long_read (round 5: the stage-1 worker has no path into the
# reference splitter any more, so the engine takes them): the whole chain of kmc_hip_split_part (stage1_chain.h under the emulation, through the C-ABI of
# tests/hipemu/libkmc_hip_mock.so) against the oracle's restatement of CSplitter::GetSeq / GetSeqLongRead + ProcessReads, with a small line_cap
class _SplitParams(C.Structure):
    _fields_ = [("kmer_len", C.c_uint32), ("signature_len", C.c_uint32), ("n_bins", C.c_uint32), ("max_x", C.c_uint32), ("both_strands", C.c_uint32),
                ("file_type", C.c_uint32), ("line_cap", C.c_uint64), ("part_kind", C.c_uint32), ("reserved", C.c_uint32)]


def _mock_split_part(text, file_type, k, m, n_bins, smap, line_cap, long_read, max_x=3, both=True):
    L = C.CDLL(emu.build_mock())
    assert L.kmc_hip_abi_version() == 4
    h = C.c_void_p()
    assert L.kmc_hip_init(None, 1, C.byref(h)) == 0
    L.kmc_hip_last_error.restype = C.c_char_p
    try:
        assert L.kmc_hip_split_set_map(h, 0, smap.ctypes.data_as(C.c_void_p), m) == 0
        p = _SplitParams(k, m, n_bins, max_x, 1 if both else 0, file_type, line_cap, 1 if long_read else 0, 0)
        t = np.frombuffer(text, dtype=np.uint8)
        recs = np.zeros(2 * t.size + 256 * (n_bins + 1) + 4096, dtype=np.uint8)
        arr = [np.zeros(n_bins, dtype=np.uint64) for _ in range(5)]
        need, n_reads = C.c_uint64(0), C.c_uint64(0)
        rc = L.kmc_hip_split_part(h, 0, 0, C.byref(p), t.ctypes.data_as(C.c_void_p), C.c_uint64(t.size), recs.ctypes.data_as(C.c_void_p), C.c_uint64(recs.size), C.byref(need),
                                  *[a.ctypes.data_as(C.c_void_p) for a in arr], C.byref(n_reads))
        if rc:
            return rc, L.kmc_hip_last_error(h)
        off, nbytes, kmers, supers, plus_x = arr
        return 0, dict(bins=[recs[int(off[b]):int(off[b] + nbytes[b])].copy() for b in range(n_bins)], kmers=kmers, supers=supers, plus_x=plus_x, n_reads=n_reads.value)
    finally:
        L.kmc_hip_destroy(h)


def _oracle_split_part(text, file_type, k, m, n_bins, smap, line_cap, long_read, max_x=3, both=True):
    """what CSplitter::ProcessReads + the collectors make of the part (oracle/oracle_engine_s1.h, in Python): records per bin, the three sums, n_reads"""
    seqs, n_reads = S1.parse_part(text, file_type, k, line_cap, long_read=long_read)
    want = dict(bins=[[] for _ in range(n_bins)], kmers=np.zeros(n_bins, dtype=np.uint64), supers=np.zeros(n_bins, dtype=np.uint64),
                plus_x=np.zeros(n_bins, dtype=np.uint64), n_reads=n_reads, pieces=len(seqs))
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for q in seqs:
        if q.size < k:
            continue
        ascii_ = letters[np.where(q < 0, 4, q)].tobytes()
        sig, off, recs = S1.split([ascii_], k, m)
        pos, ln, sg = S1.split_stream(q, k, m)
        assert np.array_equal(sig, sg)
        for i in range(sig.size):
            b = int(smap[sig[i]])
            want["bins"][b].append(bytes(recs[int(off[i]):int(off[i + 1])]))
            want["kmers"][b] += int(ln[i]) - k + 1
            want["supers"][b] += 1
            want["plus_x"][b] += S1.kxmer_recs(q[int(pos[i]):int(pos[i] + ln[i])], k, max_x, both)
    return want


def _check_part(text, file_type, k, m, long_read, line_cap, max_x=3, both=True, n_bins=37):
    smap = _sig_map(m, n_bins, 5)
    rc, got = _mock_split_part(text, file_type, k, m, n_bins, smap, line_cap, long_read, max_x, both)
    assert rc == 0, got
    want = _oracle_split_part(text, file_type, k, m, n_bins, smap, line_cap, long_read, max_x, both)
    assert got["n_reads"] == want["n_reads"]
    for b in range(n_bins):
        assert _parse_bin(got["bins"][b], k) == sorted(want["bins"][b]), b
    for key in ("kmers", "supers", "plus_x"):
        assert np.array_equal(got[key], want[key]), key
    return want


@pytest.mark.parametrize("fmt,eol,k,both", [("fq", b"\n", 27, True), ("fq", b"\r\n", 27, False), ("fa", b"\n", 55, True), ("fa", b"\r\n", 21, True)])
def test_lines_beyond_the_line_cap_are_cut_where_the_reference_cuts_them(fmt, eol, k, both):
    """CSplitter::GetSeq hands a line of mem_part_pmm_reads symbols or more to ProcessReads in pieces that overlap by k - 1 symbols (splitter.cpp:141-145,
    :226-231), and every piece starts its super-k-mers afresh. The kernels mark the piece starts in the code stream (S1_PIECE_MARK) and k_s1_cut starts a run
    there: records, k-mer / super-k-mer / k+x-mer sums of every bin must equal the reference restatement's — with a line cap small enough (k + 4105: a stride
    just beyond one workgroup window of k_s1_cut) for text of test size to have lines of 1, 2 and 5 pieces, one of exactly the cap, one a symbol short of it."""
    rng = np.random.default_rng(k)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rnd = lambda n: acgt[rng.integers(0, 4, size=n)].tobytes()
    line_cap = k + 4105
    stride = line_cap - k + 1
    long5 = bytearray(rnd(4 * stride + 900))
    long5[stride] = ord("N")  # an invalid symbol exactly where a piece starts: no mark is needed there, none may be invented
    long5[2 * stride - 3] = ord("n")
    recs = [rnd(int(rng.integers(k, 300))) for _ in range(40)] + [rnd(line_cap), rnd(60), rnd(line_cap - 1), bytes(long5), rnd(100), rnd(line_cap + 1), rnd(2 * stride + 5),
                                                                    b"A" * (line_cap + 300), rnd(80)]
    text = _records_text(fmt, eol, recs)
    want = _check_part(text, 1 if fmt == "fq" else 0, k, 9, False, line_cap, both=both)
    assert want["pieces"] >= len(recs) + 8  # the oracle did hand out pieces (1 extra for the lines of the cap and one beyond, 4 for the five-piece line, 2 + 1)
    if fmt == "fa":  # a FASTA part may end inside a long last line
        text2 = _records_text(fmt, eol, recs[:5] + [rnd(3 * stride + 77)])
        _check_part(text2[: len(text2) - len(eol)], 0, k, 9, False, line_cap, both=both)


@pytest.mark.parametrize("fmt,k", [("fa", 27), ("fq", 27), ("fa", 55), ("fq", 14)])
def test_long_read_parts_go_through_the_kernels(fmt, k):
    """ReadType::long_read parts (queues.h:40; reader: fastq_reader.cpp:619-661, :704-721, :746-790, :843-897): the first part of a read carries its title, the
    others start inside the sequence and overlap the previous one by k - 1 symbols; a FASTQ read's last part ends with the sequence line's end of line. After the
    title every byte is a symbol (CSplitter::GetSeqLongRead, splitter.cpp:70-86), handed out in pieces of mem_part_pmm_reads."""
    rng = np.random.default_rng(100 + k)
    acgt = np.frombuffer(b"ACGTACGTACGTACGTN", dtype=np.uint8)
    rnd = lambda n: acgt[rng.integers(0, acgt.size, size=n)].tobytes()
    line_cap = k + 4105
    ft = 1 if fmt == "fq" else 0
    marker = b"@" if fmt == "fq" else b">"
    body = rnd(3 * (line_cap - k + 1) + 1234)
    first = marker + b"read 1 of a long-read file\n" + body
    w = _check_part(first, ft, k, 7 if k == 14 else 9, True, line_cap)
    assert w["n_reads"] == 1 and w["pieces"] == 4
    w = _check_part(marker + b"t\r\n" + body[:3000], ft, k, 7 if k == 14 else 9, True, line_cap)  # CRLF behind the title: two invalid symbols in front
    assert w["n_reads"] == 1
    cont = body[-(k - 1):] + rnd(2 * (line_cap - k + 1) + 17)  # a continuation: no title, no read counted — even if it starts with a letter only
    w = _check_part(cont, ft, k, 7 if k == 14 else 9, True, line_cap)
    assert w["n_reads"] == 0 and w["pieces"] == 3
    w = _check_part(cont[:500] + b"\n", ft, k, 7 if k == 14 else 9, True, line_cap)  # the last part of a FASTQ read: the end of line comes along as an invalid symbol
    assert w["n_reads"] == 0
    w = _check_part(b"ACGT", ft, k, 7 if k == 14 else 9, True, line_cap)  # shorter than a k-mer
    assert sum(int(x) for x in w["kmers"]) == 0

"""-m gpu: the first stage-1 kernels on the device (kmc_hip_debug_split_reads) against the stage-1 oracle, which tests/test_stage1_oracle.py
pins to the reference. Groundwork for SURVEY.md §8f rank 2 — these kernels are not part of the drop-in yet."""
import numpy as np
import pytest

import oracle_s1 as S1
from kmc_amd import capi
from test_stage1_emulated import _reads, _stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context((0,))
    yield c
    c.close()


@pytest.mark.parametrize("k,m,n_reads", [(27, 9, 400), (21, 9, 200), (55, 9, 200), (14, 7, 100), (200, 9, 60), (28, 11, 100)])
def test_stage1_kernels_match_the_oracle(ctx, k, m, n_reads):
    rng = np.random.default_rng(k * 10 + m)
    codes = _stream(_reads(rng, k, n_reads, 150))
    sig, pos, ln, sg = ctx.debug_split_reads(codes, k, m)
    w_pos, w_len, w_sig = S1.split_stream(codes, k, m)
    assert np.array_equal(pos, w_pos.astype(np.uint64)) and np.array_equal(ln, w_len) and np.array_equal(sg, w_sig)
    valid = sig != 0xFFFFFFFF
    assert int(valid.sum()) == int(np.sum(w_len.astype(np.int64) - k + 1))  # every valid k-mer lies in exactly one super-k-mer


def test_stage1_kernels_on_a_large_stream(ctx):
    """~6 Mbp of reads: thousands of tiles, so the two look-backs of the cutting kernel really walk"""
    k = 27
    rng = np.random.default_rng(99)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, size=400_000)]
    starts = rng.integers(0, genome.size - 150, size=40_000)
    reads = [genome[s:s + 150].tobytes() for s in starts]
    codes = _stream(reads)
    sig, pos, ln, sg = ctx.debug_split_reads(codes, k, 9)
    w_pos, w_len, w_sig = S1.split_stream(codes, k, 9)
    assert np.array_equal(pos, w_pos.astype(np.uint64)) and np.array_equal(ln, w_len) and np.array_equal(sg, w_sig)


def _sig_map(m, n_bins, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, n_bins, size=(1 << (2 * m)) + 1).astype(np.int32)


def _genome_reads(rng, n_reads, genome_len, read_len=150):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, size=genome_len)]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for s in rng.integers(0, genome.size - read_len, size=n_reads):
        r = genome[s:s + read_len].tobytes()
        reads.append(r[::-1].translate(comp) if rng.random() < 0.5 else r)
    reads += [b"N" * 30, b"A" * 500, genome[:70].tobytes() + b"N" + genome[70:200].tobytes()]
    return reads


def _split_on_device(ctx, codes, k, m, smap, n_bins):
    d_codes = ctx.malloc(codes.size + 256)
    d_map = ctx.malloc(smap.nbytes)
    ctx.h2d(d_codes, codes)
    ctx.h2d(d_map, smap)
    try:
        r = ctx.split_reads_device(d_codes, codes.size, k, m, d_map, n_bins)
    finally:
        ctx.free(d_codes)
        ctx.free(d_map)
    return r


def _oracle_bins(reads, k, m, smap, n_bins):
    w_sig, w_off, w_recs = S1.split(reads, k, m)
    bins = smap[w_sig]
    lens = np.diff(w_off.astype(np.int64))
    out = []
    for b in range(n_bins):
        idx = np.nonzero(bins == b)[0]
        out.append([bytes(w_recs[int(w_off[i]):int(w_off[i + 1])]) for i in idx])
    return out, int(lens.sum())


def _records(img, k):
    out, p = [], 0
    while p < img.size:
        nb = 1 + (int(img[p]) + k + 3) // 4
        out.append(bytes(img[p:p + nb]))
        p += nb
    assert p == img.size
    return out


@pytest.mark.parametrize("k,m,n_bins,n_reads", [(27, 9, 512, 20_000), (55, 9, 64, 8_000), (21, 7, 2000, 8_000), (27, 9, 1, 30_000)])
def test_bins_made_on_the_device_hold_the_oracles_records(ctx, k, m, n_bins, n_reads):
    rng = np.random.default_rng(k + n_bins)
    reads = _genome_reads(rng, n_reads, 200_000)
    codes = _stream(reads)
    smap = _sig_map(m, n_bins, 5)
    r = _split_on_device(ctx, codes, k, m, smap, n_bins)
    try:
        want, total = _oracle_bins(reads, k, m, smap, n_bins)
        assert int(r["bytes"].sum()) == total
        buf = np.zeros(int(r["base"][n_bins]), dtype=np.uint8)
        ctx.d2h(buf, r["d_bins"])
        ps_all = np.zeros(int(r["pack_base"][n_bins]), dtype=np.uint64)
        ctx.d2h(ps_all, r["d_pack_start"])
        for b in range(n_bins):
            lo, size = int(r["base"][b]), int(r["bytes"][b])
            assert lo % 256 == 0
            got = _records(buf[lo:lo + size], k)
            assert sorted(got) == sorted(want[b]), b
            assert int(r["superkmers"][b]) == len(got) and int(r["kmers"][b]) == sum(x[0] + 1 for x in got)
            ps = ps_all[int(r["pack_base"][b]):int(r["pack_base"][b + 1])].astype(np.int64)
            assert ps[-1] == size and (size == 0 or ps[0] == 0) and (np.all(np.diff(ps) > 0) if size else ps.size == 1)
            starts = np.concatenate([[0], np.cumsum([len(x) for x in got])])
            assert np.all(np.isin(ps, starts))
            if n_bins == 1:
                assert ps.size > 3  # several packs: the boundary rule ran across pack multiples
    finally:
        ctx.free(r["d_bins"])
        ctx.free(r["d_pack_start"])


@pytest.mark.parametrize("k,n_bins", [(27, 32), (55, 8)])
def test_reads_to_database_records_without_leaving_the_device(ctx, k, n_bins):
    """reads -> stage-1 kernels -> bins in HBM -> kmc_hip_process_bins_device: per bin the suffix records, LUT and tallies of the stage-2
    oracle run on the stage-1 oracle's bin (reference order)"""
    import oracle_py as O

    m = 9
    rng = np.random.default_rng(k * 3 + n_bins)
    reads = _genome_reads(rng, 30_000, 150_000)
    codes = _stream(reads)
    smap = _sig_map(m, n_bins, 9)
    r = _split_on_device(ctx, codes, k, m, smap, n_bins)
    p = capi.make_params(k, lut_prefix_len=3, cutoff_min=2)
    rec, nl = ctx.out_rec_bytes(p), ctx.lut_entries(p)
    descs = (capi.BinDesc * n_bins)()
    allocs = []
    try:
        for b in range(n_bins):
            nk = int(r["kmers"][b])
            cap = (nk // 2 + 1) * rec
            d_out, d_lut, d_small = ctx.malloc(cap + 256), ctx.malloc(max(nl, 1) * 8), ctx.malloc(64)
            allocs.append((d_out, d_lut, d_small, cap))
            descs[b] = capi.BinDesc(r["d_bins"] + int(r["base"][b]), int(r["bytes"][b]), nk, r["d_pack_start"] + 8 * int(r["pack_base"][b]),
                                    int(r["pack_base"][b + 1] - r["pack_base"][b]) - 1, d_out, cap, d_small + 32, d_lut, d_small)
        ctx.process_bins_device(p, descs, 4)
        ctx.synchronize()
        want, _ = _oracle_bins(reads, k, m, smap, n_bins)
        po = O.make_params(k, lut_prefix_len=3, cutoff_min=2)
        for b in range(n_bins):
            d_out, d_lut, d_small, cap = allocs[b]
            small = np.zeros(8, dtype=np.uint64)
            ctx.d2h(small, d_small)
            out = np.zeros(int(small[4]), dtype=np.uint8)
            if out.size:
                ctx.d2h(out, d_out)
            lut = np.zeros(nl, dtype=np.uint64)
            ctx.d2h(lut, d_lut)
            img = np.frombuffer(b"".join(want[b]), dtype=np.uint8)
            w_out, w_lut, w_st = O.process_bin(po, img, int(r["kmers"][b]))
            assert np.array_equal(small[:4], w_st) and np.array_equal(out, w_out) and np.array_equal(lut, w_lut), b
    finally:
        for a in allocs:
            for d in a[:3]:
                ctx.free(d)
        ctx.free(r["d_bins"])
        ctx.free(r["d_pack_start"])

"""Groundwork for SURVEY.md §8f rank 2 (stage 1 on the GPU): the stage-1 ORACLE (oracle/stage1_oracle.c — minimizer signatures, super-k-mer
cutting, bin record format) pinned to the real reference. A reference run (oracle/_ref/kmc_oracle = reference stage 1 + reference pipeline,
one splitter thread) dumps every signature bin it produced; the oracle must reproduce every bin image from the reads alone, byte for byte:
the super-k-mers whose signatures a bin holds, in read order. CPU only; skipped where oracle/_ref is not built."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import golden_io
import oracle_s1 as S1
from kmc_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KMC_ORACLE = os.path.join(ROOT, "oracle", "_ref", "kmc_oracle")
needs_ref = pytest.mark.skipif(not os.path.exists(KMC_ORACLE), reason="oracle/_ref not built (needs /root/reference)")


def test_norm_table_properties():
    """mmer.h:39-95: a signature is the smaller of an m-mer and its reverse complement among the ALLOWED ones, else 4^len"""
    for m in (5, 7, 9):
        t = S1.norm_table(m)
        special = 1 << (2 * m)
        L = S1.lib()
        for x in (0, 1, special - 1, 0b000001000000000000 % special, 12345 % special):
            rc = 0
            y = x
            for _ in range(m):
                rc = (rc << 2) | (3 - (y & 3))
                y >>= 2
            want = min(x if L.oracle_s1_is_allowed(x, m) else special, rc if L.oracle_s1_is_allowed(rc, m) else special)
            assert t[x] == want
        assert t[0] == special  # AAAA... is never a signature
        # both strands of an m-mer share their signature
        idx = np.arange(special, dtype=np.uint32)
        rc = np.zeros(special, dtype=np.uint32)
        y = idx.copy()
        for _ in range(m):
            rc = (rc << np.uint32(2)) | (np.uint32(3) - (y & np.uint32(3)))
            y >>= np.uint32(2)
        assert np.array_equal(t, t[rc])


def _unpack(rec, k):
    """bin record -> symbols"""
    n = k + int(rec[0])
    b = rec[1:]
    sym = np.zeros(4 * b.size, dtype=np.int8)
    for s in range(4):
        sym[s::4] = (b >> (6 - 2 * s)) & 3
    return sym[:n]


def _walk(image, k):
    pos = 0
    while pos < image.size:
        e = int(image[pos])
        ln = 1 + (k + e + 3) // 4
        yield image[pos:pos + ln]
        pos += ln
    assert pos == image.size


def _special_reads(k):
    rng = np.random.default_rng(7)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rnd = lambda n: acgt[rng.integers(0, 4, size=n)].tobytes()
    per = rnd(11)
    reads = [
        rnd(150) + b"N" + rnd(80) + b"NN" + rnd(k - 1) + b"N" + rnd(k) + b"n" + rnd(200),  # N's: runs shorter than k vanish
        (per * 80)[:700],                      # periodic: one minimizer value for hundreds of k-mers -> the 255-extra-symbol cap
        rnd(k), rnd(k - 1), rnd(k + 1), b"N" * 40, rnd(5000),
        b"A" * 300, (b"AC" * 200), b"T" * (k + 300),  # no allowed m-mer at all: the special signature
        rnd(120).lower(),
    ]
    return reads


@needs_ref
@pytest.mark.parametrize("k", [27, 21, 55])
def test_stage1_oracle_reproduces_every_reference_bin(k, tmp_path):
    fq = str(tmp_path / "in.fq")
    synth.make_fastq(fq, seed=2027 + k, genome_len=60_000, n_reads=6_000, read_len=150)
    with open(fq, "ab") as f:
        for i, r in enumerate(_special_reads(k)):
            f.write(b"@s%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    dump = str(tmp_path / "bins.dump")
    tmp = tmp_path / "tmp"
    tmp.mkdir()
    env = dict(os.environ, KMC_BIN_DUMP=dump)
    r = subprocess.run([KMC_ORACLE, f"-k{k}", "-ci1", "-sf1", "-sp1", "-sr1", "-m2", fq, str(tmp_path / "db"), str(tmp)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
    bins = [b for b in golden_io.read_bins(dump) if b["size"]]
    assert len(bins) > 10  # a small input fills few of the 512 bins: the mapper packs the signatures it saw in its sample, most frequent first

    seqs = S1.read_fastq_sequences(fq)
    sig, rec_off, recs = S1.split(seqs, k)
    assert int(np.sum(rec_off[1:] - rec_off[:-1])) == recs.size
    # the reference's statistics agree with the oracle's cut: every k-mer is in exactly one super-k-mer
    n_kmers_oracle = int(np.sum(recs[rec_off[:-1].astype(np.int64)].astype(np.int64) + 1))
    assert n_kmers_oracle == sum(b["n_rec"] for b in bins)

    # signature -> super-k-mers of the oracle, in emission (= read) order
    order = np.argsort(sig, kind="stable")
    sig_sorted = sig[order]
    starts = np.flatnonzero(np.r_[True, sig_sorted[1:] != sig_sorted[:-1]])
    groups = {int(sig_sorted[s]): order[s:e] for s, e in zip(starts, np.r_[starts[1:], sig_sorted.size])}

    norm = S1.norm_table(9)
    seen = set()
    for b in bins:
        img = b["image"]
        # the signatures this bin holds, from its own records (all k-mers of a super-k-mer share the minimizer value)
        sigs = set()
        for rec in _walk(img, k):
            sym = _unpack(rec, k).astype(np.uint32)
            m = np.zeros(sym.size - 8, dtype=np.uint32)
            for j in range(9):
                m = (m << np.uint32(2)) | sym[j:j + m.size]
            sigs.add(int(norm[m[: k - 8]].min()))
        assert not (sigs & seen), "a signature in two bins"
        seen |= sigs
        mine = np.sort(np.concatenate([groups[s] for s in sigs]))  # read order across the bin's signatures
        want = np.concatenate([recs[int(rec_off[i]):int(rec_off[i + 1])] for i in mine])
        assert np.array_equal(want, img), (k, len(sigs), want.size, img.size)
    assert seen == set(groups), "a signature of the oracle reached no bin"

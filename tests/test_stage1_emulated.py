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


@pytest.mark.parametrize("k,m", [(27, 9), (21, 9), (55, 9), (14, 7), (200, 9), (28, 11)])
def test_emulated_stage1_signatures_and_cut_match_the_oracle(k, m):
    rng = np.random.default_rng(k * 10 + m)
    codes = _stream(_reads(rng, k, 40, 150))
    norm = S1.norm_table(m)
    err, sig, pos, ln, sg = emu.s1_split(codes, k, norm, m)
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
    for pad in (0, 1, 1000, 1023, 1024, 1025, 2047):
        codes = _stream([acgt[rng.integers(0, 4, size=pad)].tobytes() if pad else b"", (per * 400)[:4000], b"A" * 3000])
        norm = S1.norm_table(m)
        err, sig, pos, ln, sg = emu.s1_split(codes, k, norm, m)
        w_pos, w_len, w_sig = S1.split_stream(codes, k, m)
        assert err == 0 and np.array_equal(pos, w_pos.astype(np.uint64)) and np.array_equal(ln, w_len) and np.array_equal(sg, w_sig), pad

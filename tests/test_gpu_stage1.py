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

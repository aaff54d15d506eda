"""Deterministic synthetic inputs (SURVEY.md §8d "Synthetic inputs").

FASTQ shape: genome = i.i.d. uniform ACGT of length G; reads = L-bp windows at uniform random
starts, each base substituted with probability `err` by a different base, reverse-complemented
with probability 0.5; constant quality 'I'; no N. NumPy ``default_rng(seed)``.

Used by tests (to drive the reference binaries in oracle/_ref) and by bench.py's cpu_baseline
leg. The large device-side workloads of bench.py come from kmc_amd/csrc/synth_bins.cpp instead
(same read model, generated straight into super-k-mer bin images).
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_reads(seed: int, genome_len: int, n_reads: int, read_len: int = 150, err: float = 0.01) -> np.ndarray:
    """Return reads as a (n_reads, read_len) uint8 array of 2-bit symbols (A=0 C=1 G=2 T=3)."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    starts = rng.integers(0, genome_len - read_len + 1, size=n_reads)
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    step = 1 << 16
    ar = np.arange(read_len)
    for lo in range(0, n_reads, step):
        hi = min(n_reads, lo + step)
        r = genome[starts[lo:hi, None] + ar[None, :]]
        mask = rng.random(r.shape) < err
        sub = rng.integers(1, 4, size=r.shape, dtype=np.uint8)
        r = np.where(mask, (r + sub) & 3, r).astype(np.uint8)
        flip = rng.random(hi - lo) < 0.5
        r[flip] = (3 - r[flip])[:, ::-1]
        out[lo:hi] = r
    return out


def write_fastq(path: str, reads: np.ndarray) -> int:
    """Write reads as 4-line FASTQ records ("@r<9 digits>", sequence, "+", quality 'I'); returns bytes written."""
    n, L = reads.shape
    rec = 12 + L + 3 + L + 1
    total = 0
    with open(path, "wb") as f:
        step = 1 << 16
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            m = hi - lo
            buf = np.empty((m, rec), dtype=np.uint8)
            buf[:, 0] = ord("@")
            buf[:, 1] = ord("r")
            ids = np.arange(lo, hi, dtype=np.int64)
            for d in range(9):
                buf[:, 10 - d] = (ids // 10**d) % 10 + ord("0")
            buf[:, 11] = ord("\n")
            buf[:, 12 : 12 + L] = _ACGT[reads[lo:hi]]
            buf[:, 12 + L] = ord("\n")
            buf[:, 13 + L] = ord("+")
            buf[:, 14 + L] = ord("\n")
            buf[:, 15 + L : 15 + 2 * L] = ord("I")
            buf[:, 15 + 2 * L] = ord("\n")
            f.write(buf.tobytes())
            total += buf.size
    return total


def make_fastq(path: str, seed: int, genome_len: int, n_reads: int, read_len: int = 150, err: float = 0.01) -> int:
    return write_fastq(path, make_reads(seed, genome_len, n_reads, read_len, err))


# Named configurations of BASELINE.json / SURVEY.md §8d
CONFIGS = {
    "C1": dict(seed=12345, genome_len=400_000, n_reads=33_000),          # 10 MB FASTQ plumbing case
    "C2": dict(seed=2026, genome_len=66_000_000, n_reads=13_300_000),    # ~2 Gbp
    "C3": dict(seed=2026, genome_len=1_000_000_000, n_reads=200_000_000),  # ~30 Gbp
}

if __name__ == "__main__":
    import sys

    name, path = sys.argv[1], sys.argv[2]
    print(make_fastq(path, **CONFIGS[name]))

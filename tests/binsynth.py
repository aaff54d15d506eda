"""Small random bin images for tests: super-k-mers [e][packed k+e symbols] (kb_collector.cpp:57-71)."""
from __future__ import annotations

import numpy as np


def pack_superkmer(symbols: np.ndarray, k: int) -> bytes:
    n = symbols.size
    e = n - k
    assert 0 <= e <= 255
    pad = (-n) % 4
    s = np.concatenate([symbols.astype(np.uint8), np.zeros(pad, dtype=np.uint8)]).reshape(-1, 4)
    packed = (s[:, 0] << 6) | (s[:, 1] << 4) | (s[:, 2] << 2) | s[:, 3]
    return bytes([e]) + packed.astype(np.uint8).tobytes()


def random_bin(rng, k: int, n_super: int, max_extra: int = 40, genome: np.ndarray | None = None, pack_size: int = 4096):
    """Returns (image uint8 ndarray, n_kmers, pack_bytes uint64 ndarray). With `genome`, super-k-mers are windows
    of it (so k-mers repeat and counts > 1 occur); otherwise i.i.d. symbols."""
    chunks, packs = [], []
    n_k, cur, in_pack = 0, 0, 0
    for _ in range(n_super):
        e = int(rng.integers(0, max_extra + 1))
        if genome is not None:
            st = int(rng.integers(0, genome.size - (k + e) + 1))
            sym = genome[st : st + k + e].copy()
            if rng.random() < 0.5:
                sym = (3 - sym)[::-1]
        else:
            sym = rng.integers(0, 4, size=k + e, dtype=np.uint8)
        b = pack_superkmer(sym, k)
        if in_pack >= pack_size:
            packs.append(cur)
            cur, in_pack = 0, 0
        chunks.append(b)
        cur += len(b)
        in_pack += 1
        n_k += e + 1
    if cur:
        packs.append(cur)
    img = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if chunks else np.zeros(0, dtype=np.uint8)
    return img, n_k, np.array(packs, dtype=np.uint64)

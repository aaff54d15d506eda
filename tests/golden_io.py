"""Reader for the per-bin fixture files under tests/golden/ (format written by the oracle engine of
kmc_amd/host/kb_sorter_plugin.h when $KMC_BIN_DUMP is set; see tests/golden/make_golden.py)."""
from __future__ import annotations

import struct

import numpy as np

MAGIC = 0x4B4D4342494E3031  # "KMCBIN01"
PARAMS_FMT = "<IIIIQQII"  # struct kmc_hip_bin_params


def read_bins(path: str):
    """Yield dicts: params(tuple), size, n_rec, pack_bytes, image, out, lut, stats."""
    data = open(path, "rb").read()
    pos = 0
    while pos < len(data):
        magic, size, n_rec, n_packs, out_bytes, lut_n, _, _ = struct.unpack_from("<8Q", data, pos)
        assert magic == MAGIC, "bad fixture"
        pos += 64
        params = struct.unpack_from(PARAMS_FMT, data, pos)
        pos += struct.calcsize(PARAMS_FMT)
        stats = np.frombuffer(data, dtype=np.uint64, count=4, offset=pos).copy()
        pos += 32
        packs = np.frombuffer(data, dtype=np.uint64, count=n_packs, offset=pos).copy()
        pos += 8 * n_packs
        image = np.frombuffer(data, dtype=np.uint8, count=size, offset=pos).copy()
        pos += size
        out = np.frombuffer(data, dtype=np.uint8, count=out_bytes, offset=pos).copy()
        pos += out_bytes
        lut = np.frombuffer(data, dtype=np.uint64, count=lut_n, offset=pos).copy()
        pos += 8 * lut_n
        yield dict(params=params, size=size, n_rec=n_rec, pack_bytes=packs, image=image, out=out, lut=lut, stats=stats)


def write_bins(path: str, bins) -> None:
    with open(path, "wb") as f:
        for b in bins:
            f.write(struct.pack("<8Q", MAGIC, b["size"], b["n_rec"], len(b["pack_bytes"]), len(b["out"]), len(b["lut"]), 0, 0))
            f.write(struct.pack(PARAMS_FMT, *b["params"]))
            f.write(b["stats"].astype("<u8").tobytes())
            f.write(b["pack_bytes"].astype("<u8").tobytes())
            f.write(b["image"].tobytes())
            f.write(b["out"].tobytes())
            f.write(b["lut"].astype("<u8").tobytes())

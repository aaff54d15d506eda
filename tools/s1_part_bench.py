"""Timing of kmc_hip_split_part (include/kmc_hip.h): one part of FASTQ text in host memory -> its bin records and collector sums in host
memory, i.e. what the stage-1 worker plug-in pays per part (H2D of the text, the kernel chain of kmc_amd/csrc/stage1_chain.h, D2H of the
records). NOT RUN YET when this was committed: the first GPU session of the next round starts here. numpy + the C-ABI only."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmc_amd import capi  # noqa: E402


class SplitParams(C.Structure):
    _fields_ = [("kmer_len", C.c_uint32), ("signature_len", C.c_uint32), ("n_bins", C.c_uint32), ("max_x", C.c_uint32), ("both_strands", C.c_uint32),
                ("file_type", C.c_uint32), ("line_cap", C.c_uint64)]


def make_fastq(n_reads, read_len, seed):
    rng = np.random.default_rng(seed)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n_reads, read_len))]
    title = np.frombuffer(b"@read/0123456789 len=150\n", dtype=np.uint8)
    rec = np.empty((n_reads, title.size + read_len + 1 + 2 + read_len + 1), dtype=np.uint8)
    o = 0
    rec[:, o:o + title.size] = title
    o += title.size
    rec[:, o:o + read_len] = seq
    o += read_len
    rec[:, o] = 10
    rec[:, o + 1] = ord("+")
    rec[:, o + 2] = 10
    o += 3
    rec[:, o:o + read_len] = ord("I")
    rec[:, o + read_len] = 10
    return np.ascontiguousarray(rec.reshape(-1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbytes", type=int, default=32, help="size of the part (the reference's readers cut parts of 2^23 .. 2^25 bytes)")
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--m", type=int, default=9)
    ap.add_argument("--bins", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    rec_bytes = 25 + 150 + 1 + 2 + 150 + 1
    text = make_fastq((a.mbytes << 20) // rec_bytes, 150, 1)
    smap = np.random.default_rng(2).integers(0, a.bins, size=(1 << (2 * a.m)) + 1).astype(np.int32)
    ctx = capi.Context((0,))
    L, h = ctx.L, ctx.h
    L.kmc_hip_split_set_map.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
    L.kmc_hip_split_part.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64] + [C.c_void_p] * 7
    ctx._chk(L.kmc_hip_split_set_map(h, 0, smap.ctypes.data, a.m))
    p = SplitParams(a.k, a.m, a.bins, 3, 1, 1, 131080)
    recs = np.zeros(text.size + 256 * (a.bins + 1), dtype=np.uint8)
    arr = np.zeros((5, a.bins), dtype=np.uint64)
    need, n_reads = C.c_uint64(0), C.c_uint64(0)
    best = None
    for rep in range(a.reps + 1):
        t0 = time.perf_counter()
        rc = L.kmc_hip_split_part(h, 0, 0, C.byref(p), text.ctypes.data, text.size, recs.ctypes.data, recs.size, C.byref(need), arr[0].ctypes.data, arr[1].ctypes.data,
                                  arr[2].ctypes.data, arr[3].ctypes.data, arr[4].ctypes.data, C.byref(n_reads))
        dt = time.perf_counter() - t0
        ctx._chk(rc)
        if rep and (best is None or dt < best):
            best = dt
    symbols = int(n_reads.value) * 150
    print(json.dumps(dict(what="kmc_hip_split_part: one FASTQ part, host text -> host records (wall-clock of the synchronous C-ABI call, best of %d)" % a.reps,
                          text_bytes=int(text.size), reads=int(n_reads.value), symbols=symbols, k=a.k, bins=a.bins, record_bytes=int(arr[1].sum()),
                          kmers=int(arr[2].sum()), superkmers=int(arr[3].sum()), plus_x=int(arr[4].sum()), seconds=best, text_GBs=text.size / best / 1e9,
                          gsymbols_per_s=symbols / best / 1e9)))
    ctx.close()


if __name__ == "__main__":
    main()

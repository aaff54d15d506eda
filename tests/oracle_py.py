"""ctypes binding of oracle/liboracle_stage2.so — TEST INFRASTRUCTURE (the checker, never the product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


class OracleParams(C.Structure):
    _fields_ = [
        ("kmer_len", C.c_uint32),
        ("both_strands", C.c_uint32),
        ("cutoff_min", C.c_uint32),
        ("without_output", C.c_uint32),
        ("cutoff_max", C.c_uint64),
        ("counter_max", C.c_uint64),
        ("lut_prefix_len", C.c_uint32),
        ("output_type", C.c_uint32),
    ]


def build_oracle() -> str:
    so = os.path.join(ORACLE_DIR, "liboracle_stage2.so")
    src = os.path.join(ORACLE_DIR, "stage2_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", so, src])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build_oracle())
        u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
        L.oracle_words.restype = C.c_uint32
        L.oracle_counter_size.restype = C.c_uint32
        L.oracle_counter_size.argtypes = [C.c_uint64, C.c_uint64]
        L.oracle_out_rec_bytes.restype = C.c_uint32
        L.oracle_scan.argtypes = [C.c_uint32, u8p, C.c_uint64, u64p, u64p]
        L.oracle_expand.argtypes = [C.POINTER(OracleParams), u8p, C.c_uint64, u64p, C.c_uint64, u64p]
        L.oracle_sort.argtypes = [u64p, C.c_uint64, C.c_uint32]
        L.oracle_sort.restype = None
        L.oracle_compact.argtypes = [C.POINTER(OracleParams), u64p, C.c_uint64, u8p, C.c_uint64, u64p, u64p, u64p]
        L.oracle_process_bin.argtypes = [C.POINTER(OracleParams), u8p, C.c_uint64, C.c_uint64, u8p, C.c_uint64, u64p, u64p, u64p]
        _LIB = L
    return _LIB


def _p8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def make_params(k, both_strands=1, cutoff_min=2, cutoff_max=10**9, counter_max=255, lut_prefix_len=3, output_type=0,
                without_output=0) -> OracleParams:
    return OracleParams(k, both_strands, cutoff_min, without_output, cutoff_max, counter_max, lut_prefix_len, output_type)


def words(k):
    return (k + 31) // 32


def scan(k, data: np.ndarray):
    ns, nk = C.c_uint64(), C.c_uint64()
    rc = lib().oracle_scan(k, _p8(data), data.size, C.byref(ns), C.byref(nk))
    if rc:
        raise ValueError("ragged super-k-mer stream")
    return ns.value, nk.value


def expand(p: OracleParams, data: np.ndarray) -> np.ndarray:
    _, nk = scan(p.kmer_len, data)
    out = np.zeros((max(nk, 1), words(p.kmer_len)), dtype=np.uint64)
    n = C.c_uint64()
    rc = lib().oracle_expand(C.byref(p), _p8(data), data.size, _p64(out), nk, C.byref(n))
    assert rc == 0 and n.value == nk
    return out[:nk]


def sort(recs: np.ndarray) -> np.ndarray:
    r = np.ascontiguousarray(recs.copy())
    lib().oracle_sort(_p64(r), r.shape[0], r.shape[1])
    return r


def lut_entries(p: OracleParams) -> int:
    return (1 << (2 * p.lut_prefix_len)) if p.lut_prefix_len else 0


def process_bin(p: OracleParams, data: np.ndarray, n_rec=None):
    """Returns (out_bytes ndarray, lut ndarray, stats[4])."""
    if n_rec is None:
        _, n_rec = scan(p.kmer_len, data)
    rec_bytes = lib().oracle_out_rec_bytes(C.byref(p))
    cap = (n_rec + 1) * max(rec_bytes, 1)
    out = np.zeros(cap, dtype=np.uint8)
    lut = np.zeros(max(lut_entries(p), 1), dtype=np.uint64)
    stats = np.zeros(4, dtype=np.uint64)
    ob = C.c_uint64()
    d = data if data.size else np.zeros(1, dtype=np.uint8)
    rc = lib().oracle_process_bin(C.byref(p), _p8(d), data.size, n_rec, _p8(out), cap, C.byref(ob), _p64(lut), _p64(stats))
    if rc:
        raise RuntimeError(f"oracle_process_bin rc={rc}")
    return out[: ob.value].copy(), lut[: lut_entries(p)].copy(), stats

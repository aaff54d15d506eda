"""ctypes binding of oracle/liboracle_stage1.so — TEST INFRASTRUCTURE: the C restatement of KMC's stage-1 splitter (minimizer signatures,
super-k-mer cutting, bin record format). See oracle/stage1_oracle.c and tests/test_stage1_oracle.py."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None
_CODES = np.full(256, -1, dtype=np.int8)  # splitter.cpp:42-47
for _c, _v in zip("ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
    _CODES[ord(_c)] = _v


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "liboracle_stage1.so")
        src = os.path.join(ORACLE_DIR, "stage1_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", so, src])
        L = C.CDLL(so)
        L.oracle_s1_norm.argtypes = [C.c_uint32, C.c_void_p]
        L.oracle_s1_is_allowed.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_s1_split.restype = C.c_uint64
        L.oracle_s1_split.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64]
        L.oracle_s1_split_all.restype = C.c_int64
        L.oracle_s1_split_all.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.oracle_s1_kxmer_recs.restype = C.c_uint32
        L.oracle_s1_kxmer_recs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.oracle_s1_parse_part.restype = C.c_int64
        L.oracle_s1_parse_part.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.oracle_s1_parse_long_read_part.restype = C.c_int64
        L.oracle_s1_parse_long_read_part.argtypes = L.oracle_s1_parse_part.argtypes
        _LIB = L
    return _LIB


def norm_table(sig_len: int) -> np.ndarray:
    t = np.zeros(1 << (2 * sig_len), dtype=np.uint32)
    assert lib().oracle_s1_norm(sig_len, t.ctypes.data) == 0
    return t


def encode(seqs):
    """list of ASCII sequences (bytes/str) -> (codes int8 back to back, offsets uint64[n+1])"""
    raw = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    off = np.zeros(len(raw) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in raw])
    codes = _CODES[np.frombuffer(b"".join(raw), dtype=np.uint8)] if raw else np.zeros(0, dtype=np.int8)
    return np.ascontiguousarray(codes), off


def split(seqs, k: int, sig_len: int = 9):
    """-> (signature uint32[n_sk], rec_off uint64[n_sk+1], recs uint8[...]) in emission order (CSplitter::ProcessReads, read by read)"""
    codes, off = encode(seqs)
    total = int(off[-1])
    cap_sk = total + len(seqs) + 16
    cap_bytes = total + 2 * cap_sk + 64
    sig = np.zeros(cap_sk, dtype=np.uint32)
    rec_off = np.zeros(cap_sk + 1, dtype=np.uint64)
    recs = np.zeros(cap_bytes, dtype=np.uint8)
    n = lib().oracle_s1_split_all(codes.ctypes.data, off.ctypes.data, len(seqs), k, sig_len, sig.ctypes.data, rec_off.ctypes.data, recs.ctypes.data, cap_sk, cap_bytes)
    assert n >= 0, n
    return sig[:n].copy(), rec_off[: n + 1].copy(), recs[: int(rec_off[n])].copy()


def read_fastq_sequences(path):
    with open(path, "rb") as f:
        return [ln.rstrip(b"\r\n") for i, ln in enumerate(f) if i % 4 == 1]


def split_stream(codes: np.ndarray, k: int, sig_len: int = 9):
    """one code stream (reads joined by a negative separator) -> (pos, len, signature) of every super-k-mer, emission order"""
    codes = np.ascontiguousarray(codes, dtype=np.int8)
    norm = norm_table(sig_len)
    cap = codes.size + 8
    out = np.zeros((cap, 3), dtype=np.uint32)
    n = lib().oracle_s1_split(codes.ctypes.data, codes.size, k, sig_len, norm.ctypes.data, out.ctypes.data, cap)
    assert n <= cap
    return out[:n, 0].copy(), out[:n, 1].copy(), out[:n, 2].copy()


def parse_part(text: bytes, file_type: int, k: int, line_cap: int = 131080, long_read: bool = False):
    """CSplitter::GetSeq over one part (0 = FASTA, 1 = FASTQ; long_read: a part the reader labelled ReadType::long_read, GetSeqLongRead) -> (list of code arrays, n_reads)"""
    t = np.frombuffer(text, dtype=np.uint8)
    codes = np.zeros(t.size + (t.size // max(line_cap - k + 1, 1) + 2) * k + 16, dtype=np.int8)  # the pieces of an over-long line overlap by k - 1
    off = np.zeros(t.size // 2 + 16, dtype=np.uint64)
    nr = C.c_uint64(0)
    fn = lib().oracle_s1_parse_long_read_part if long_read else lib().oracle_s1_parse_part
    n = fn(t.ctypes.data, t.size, file_type, k, line_cap, codes.ctypes.data, off.ctypes.data, off.size - 1, C.addressof(nr))
    assert n >= 0
    return [codes[int(off[i]):int(off[i + 1])].copy() for i in range(n)], nr.value


def kxmer_recs(seq: np.ndarray, k: int, max_x: int, both_strands: bool) -> int:
    s = np.ascontiguousarray(seq, dtype=np.int8)
    return int(lib().oracle_s1_kxmer_recs(s.ctypes.data, s.size, k, max_x, 1 if both_strands else 0))

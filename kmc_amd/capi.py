"""ctypes binding of the C-ABI in include/kmc_hip.h (libkmc_hip.so).

This is the only way Python code (tests, bench.py) reaches the product: through the same `extern "C"` entry points
the C++ worker (kmc_amd/host/kb_sorter_plugin.h) binds. There is no CPU fallback: a missing library or a missing
GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)


class KmcHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"kmc_hip error {code}: {msg}")
        self.code = code


class HostBin(C.Structure):
    """struct kmc_hip_host_bin: one bin of a kmc_hip_process_bins_submit call (host pointers)"""

    _fields_ = [
        ("superkmers", C.c_void_p),
        ("size", C.c_uint64),
        ("n_rec", C.c_uint64),
        ("pack_bytes", C.c_void_p),
        ("n_packs", C.c_uint64),
        ("out_suffix", C.c_void_p),
        ("out_capacity", C.c_uint64),
        ("lut", C.c_void_p),
    ]


class BinParams(C.Structure):
    """struct kmc_hip_bin_params (mirrors the CKMCParams fields of kb_sorter.h:165-200)."""

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


class BinDesc(C.Structure):
    """struct kmc_hip_bin_desc: one device-resident bin of kmc_hip_process_bins_device."""

    _fields_ = [
        ("d_superkmers", C.c_void_p),
        ("size", C.c_uint64),
        ("n_rec", C.c_uint64),
        ("d_pack_start", C.c_void_p),
        ("n_packs", C.c_uint64),
        ("d_out", C.c_void_p),
        ("out_capacity", C.c_uint64),
        ("d_out_bytes", C.c_void_p),
        ("d_lut", C.c_void_p),
        ("d_stats", C.c_void_p),
    ]


def make_params(k, both_strands=1, cutoff_min=2, cutoff_max=10**9, counter_max=255, lut_prefix_len=3, output_type=0,
                without_output=0) -> BinParams:
    return BinParams(k, both_strands, cutoff_min, without_output, cutoff_max, counter_max, lut_prefix_len, output_type)


# every symbol include/kmc_hip.h declares (tests check the library exports them all)
SYMBOLS = [
    "kmc_hip_init", "kmc_hip_destroy", "kmc_hip_last_error", "kmc_hip_abi_version", "kmc_hip_backend_kind", "kmc_hip_num_devices", "kmc_hip_num_slots",
    "kmc_hip_device_count",
    "kmc_hip_words", "kmc_hip_counter_size", "kmc_hip_out_rec_bytes", "kmc_hip_lut_entries",
    "kmc_hip_sort_records", "kmc_hip_sort_records_into", "kmc_hip_sort_records_device",
    "kmc_hip_process_bin", "kmc_hip_process_bin_multi", "kmc_hip_process_bins_submit", "kmc_hip_process_bins_wait", "kmc_hip_process_bin_submit", "kmc_hip_process_bin_wait", "kmc_hip_process_bin_device",
    "kmc_hip_process_bins_device", "kmc_hip_order_database_device",
    "kmc_hip_allreduce_stats", "kmc_hip_last_timings", "kmc_hip_scatter_totals", "kmc_hip_local_sort_totals", "kmc_hip_set_hybrid", "kmc_hip_path_counters", "kmc_hip_host_boundary_times", "kmc_hip_reserve_slot",
    "kmc_hip_malloc", "kmc_hip_free", "kmc_hip_memcpy_h2d", "kmc_hip_memcpy_d2h",
    "kmc_hip_host_register", "kmc_hip_host_unregister", "kmc_hip_host_alloc", "kmc_hip_host_free", "kmc_hip_synchronize",
    "kmc_hip_debug_expand", "kmc_hip_debug_compact", "kmc_hip_debug_split_reads",
    "kmc_hip_split_reads_plan", "kmc_hip_split_reads_emit", "kmc_hip_split_reads_free",
    "kmc_hip_split_set_map", "kmc_hip_split_part",
]

_LIB = None


def lib_path() -> str:
    """In-tree libkmc_hip.so; $KMC_HIP_LIB overrides (tuning variants built by tools/build_variants.py)."""
    return os.environ.get("KMC_HIP_LIB") or _build.LIB_HIP


def backend_kind() -> int:
    """0 = the GPU library, 1 = the CPU emulation of the host library (tests), 2 = the mock (tests)"""
    return int(load().kmc_hip_backend_kind())


def require_gpu_backend():
    """bench.py / smoke(): refuse a test build of the library (CPU emulation or mock, see kmc_hip_backend_kind) — numbers and certificates are
    for the GPU library only."""
    kind = load().kmc_hip_backend_kind()
    if kind != 0:
        raise RuntimeError(f"{lib_path()} is a test build of libkmc_hip (backend kind {kind}: CPU emulation / mock), not the GPU library")


def load():
    """dlopen libkmc_hip.so (in-tree). Raises FileNotFoundError when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)")
    L = C.CDLL(p, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    L.kmc_hip_init.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.kmc_hip_destroy.argtypes = [vp]
    L.kmc_hip_destroy.restype = None
    L.kmc_hip_last_error.argtypes = [vp]
    L.kmc_hip_last_error.restype = C.c_char_p
    L.kmc_hip_num_devices.argtypes = [vp]
    L.kmc_hip_words.argtypes = [C.c_uint32]
    L.kmc_hip_words.restype = C.c_uint32
    L.kmc_hip_counter_size.argtypes = [C.c_uint64, C.c_uint64]
    L.kmc_hip_counter_size.restype = C.c_uint32
    L.kmc_hip_out_rec_bytes.argtypes = [C.POINTER(BinParams)]
    L.kmc_hip_out_rec_bytes.restype = C.c_uint32
    L.kmc_hip_lut_entries.argtypes = [C.POINTER(BinParams)]
    L.kmc_hip_lut_entries.restype = C.c_uint64
    L.kmc_hip_sort_records.argtypes = [vp, C.c_int, vp, C.c_uint64, C.c_uint32, C.c_uint32]
    L.kmc_hip_sort_records_into.argtypes = [vp, C.c_int, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32]
    L.kmc_hip_sort_records_device.argtypes = [vp, C.c_int, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.kmc_hip_process_bin.argtypes = [vp, C.c_int, C.POINTER(BinParams), vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64,
                                      u64p, vp, u64p]
    L.kmc_hip_process_bin_multi.argtypes = [vp, C.POINTER(BinParams), vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, u64p, vp, u64p]
    L.kmc_hip_process_bins_submit.argtypes = [vp, C.c_int, C.c_int, C.POINTER(BinParams), C.POINTER(HostBin), C.c_uint32]
    L.kmc_hip_process_bins_wait.argtypes = [vp, C.c_int, C.c_int, u64p, u64p]
    L.kmc_hip_process_bin_submit.argtypes = [vp, C.c_int, C.c_int, C.POINTER(BinParams), vp, C.c_uint64, C.c_uint64, vp, C.c_uint64,
                                             vp, C.c_uint64, vp]
    L.kmc_hip_process_bin_wait.argtypes = [vp, C.c_int, C.c_int, u64p, u64p]
    L.kmc_hip_process_bin_device.argtypes = [vp, C.c_int, C.POINTER(BinParams), vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp,
                                             C.c_uint64, vp, vp, vp, C.c_int]
    L.kmc_hip_allreduce_stats.argtypes = [vp, u64p]
    L.kmc_hip_last_timings.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.kmc_hip_scatter_totals.argtypes = [vp, C.c_int, C.c_int, u64p, C.POINTER(C.c_double), u64p]
    L.kmc_hip_local_sort_totals.argtypes = [vp, C.c_int, C.c_int, u64p, C.POINTER(C.c_double), u64p, u64p, u64p]
    L.kmc_hip_process_bins_device.argtypes = [vp, C.c_int, C.POINTER(BinParams), C.POINTER(BinDesc), C.c_uint64, C.c_int]
    L.kmc_hip_order_database_device.argtypes = [vp, C.c_int, C.POINTER(BinParams), C.POINTER(BinDesc), C.c_uint64, C.c_uint32, vp, C.c_uint64, vp, u64p]
    L.kmc_hip_host_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.kmc_hip_host_free.argtypes = [vp, vp]
    L.kmc_hip_malloc.argtypes = [vp, C.c_int, C.c_uint64, C.POINTER(vp)]
    L.kmc_hip_free.argtypes = [vp, C.c_int, vp]
    L.kmc_hip_memcpy_h2d.argtypes = [vp, C.c_int, vp, vp, C.c_uint64]
    L.kmc_hip_memcpy_d2h.argtypes = [vp, C.c_int, vp, vp, C.c_uint64]
    L.kmc_hip_host_register.argtypes = [vp, vp, C.c_uint64]
    L.kmc_hip_host_unregister.argtypes = [vp, vp]
    L.kmc_hip_synchronize.argtypes = [vp, C.c_int]
    L.kmc_hip_debug_expand.argtypes = [vp, C.c_int, C.POINTER(BinParams), vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp]
    L.kmc_hip_debug_compact.argtypes = [vp, C.c_int, C.POINTER(BinParams), vp, C.c_uint64, vp, C.c_uint64, u64p, vp, u64p]
    L.kmc_hip_debug_split_reads.argtypes = [vp, C.c_int, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, vp, vp, C.c_uint64, u64p]
    _LIB = L
    return L


def _vp(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """kmc_hip_ctx: one per process, `devices` = HIP ordinals."""

    def __init__(self, devices=(0,)):
        self.L = load()
        ids = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.L.kmc_hip_init(ids, len(devices), C.byref(h))
        if rc:
            raise KmcHipError(rc, self.L.kmc_hip_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.kmc_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise KmcHipError(rc, self.L.kmc_hip_last_error(self.h).decode())

    # ---- derived sizes
    def out_rec_bytes(self, p: BinParams) -> int:
        return self.L.kmc_hip_out_rec_bytes(C.byref(p))

    def lut_entries(self, p: BinParams) -> int:
        return self.L.kmc_hip_lut_entries(C.byref(p))

    # ---- narrow boundary
    def sort_records(self, recs: np.ndarray, key_bytes: int, dev: int = 0) -> np.ndarray:
        """Sort (n, words) uint64 records ascending by their low key_bytes bytes; returns a sorted copy."""
        r = np.ascontiguousarray(recs, dtype=np.uint64).copy()
        if r.ndim == 1:
            r = r.reshape(-1, 1)
        self._chk(self.L.kmc_hip_sort_records(self.h, dev, _vp(r), r.shape[0], r.shape[1], key_bytes))
        return r

    # ---- full boundary
    def process_bin(self, p: BinParams, image: np.ndarray, n_rec: int, pack_bytes=None, out_capacity=None, dev: int = 0, multi: bool = False):
        """One bin through kmc_hip_process_bin — or, multi=True, through kmc_hip_process_bin_multi: the same bin over every device of the context.
        Returns (suffix_bytes ndarray, lut ndarray, stats ndarray[4])."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        rec = self.out_rec_bytes(p)
        if out_capacity is None:
            out_capacity = ((n_rec + 1) // max(p.cutoff_min, 1)) * rec  # kb_reader.h:141-150
        out = np.zeros(max(out_capacity, 1), dtype=np.uint8)
        nl = self.lut_entries(p)
        lut = np.zeros(max(nl, 1), dtype=np.uint64)
        stats = np.zeros(4, dtype=np.uint64)
        ob = C.c_uint64()
        if pack_bytes is None:
            pb, npk = None, 0
        else:
            pack_bytes = np.ascontiguousarray(pack_bytes, dtype=np.uint64)
            pb, npk = _vp(pack_bytes), pack_bytes.size
        src = image if image.size else np.zeros(1, dtype=np.uint8)
        if multi:
            self._chk(self.L.kmc_hip_process_bin_multi(self.h, C.byref(p), _vp(src), image.size, n_rec, pb, npk, _vp(out), out_capacity, C.byref(ob), _vp(lut),
                                                       stats.ctypes.data_as(u64p)))
        else:
            self._chk(self.L.kmc_hip_process_bin(self.h, dev, C.byref(p), _vp(src), image.size, n_rec, pb, npk, _vp(out), out_capacity,
                                                 C.byref(ob), _vp(lut), stats.ctypes.data_as(u64p)))
        return out[: ob.value].copy(), lut[:nl].copy(), stats

    def process_bins_host(self, p: BinParams, bins, out_capacity=None, slot: int = 0, dev: int = 0):
        """Up to 16 bins [(image, n_rec, pack_bytes or None), ...] through ONE kmc_hip_process_bins_submit / _wait pair (the bins are sorted together
        where the key has spare bits). Returns [(suffix_bytes, lut, stats[4]), ...] in the order of the bins."""
        n = len(bins)
        rec, nl = self.out_rec_bytes(p), self.lut_entries(p)
        arr = (HostBin * max(n, 1))()
        keep, outs, luts = [], [], []
        for i, (image, n_rec, pack_bytes) in enumerate(bins):
            image = np.ascontiguousarray(image, dtype=np.uint8)
            cap = ((n_rec + 1) // max(p.cutoff_min, 1)) * rec if out_capacity is None else out_capacity
            out, lut = np.zeros(max(cap, 1), dtype=np.uint8), np.zeros(max(nl, 1), dtype=np.uint64)
            src = image if image.size else np.zeros(1, dtype=np.uint8)
            pb = None if pack_bytes is None else np.ascontiguousarray(pack_bytes, dtype=np.uint64)
            keep += [src, pb]
            outs.append(out)
            luts.append(lut)
            arr[i] = HostBin(src.ctypes.data, image.size, n_rec, None if pb is None else pb.ctypes.data, 0 if pb is None else pb.size, out.ctypes.data, cap,
                             lut.ctypes.data)
        self._chk(self.L.kmc_hip_process_bins_submit(self.h, dev, slot, C.byref(p), arr, n))
        ob, st = np.zeros(max(n, 1), dtype=np.uint64), np.zeros((max(n, 1), 4), dtype=np.uint64)
        self._chk(self.L.kmc_hip_process_bins_wait(self.h, dev, slot, ob.ctypes.data_as(u64p), st.ctypes.data_as(u64p)))
        return [(outs[i][: int(ob[i])].copy(), luts[i][:nl].copy(), st[i].copy()) for i in range(n)]

    # ---- stage-isolating test hooks
    def debug_expand(self, p: BinParams, image: np.ndarray, n_rec: int, pack_bytes: np.ndarray, dev: int = 0) -> np.ndarray:
        words = (p.kmer_len + 31) // 32
        out = np.zeros((n_rec, words), dtype=np.uint64)
        pack_bytes = np.ascontiguousarray(pack_bytes, dtype=np.uint64)
        self._chk(self.L.kmc_hip_debug_expand(self.h, dev, C.byref(p), _vp(image), image.size, n_rec, _vp(pack_bytes), pack_bytes.size, _vp(out)))
        return out

    def debug_compact(self, p: BinParams, sorted_recs: np.ndarray, out_capacity=None, dev: int = 0):
        r = np.ascontiguousarray(sorted_recs, dtype=np.uint64)
        n = r.shape[0]
        rec = self.out_rec_bytes(p)
        if out_capacity is None:
            out_capacity = (n + 1) * rec
        out = np.zeros(max(out_capacity, 1), dtype=np.uint8)
        nl = self.lut_entries(p)
        lut = np.zeros(max(nl, 1), dtype=np.uint64)
        stats = np.zeros(4, dtype=np.uint64)
        ob = C.c_uint64()
        self._chk(self.L.kmc_hip_debug_compact(self.h, dev, C.byref(p), _vp(r), n, _vp(out), out_capacity, C.byref(ob), _vp(lut),
                                               stats.ctypes.data_as(u64p)))
        return out[: ob.value].copy(), lut[:nl].copy(), stats

    def debug_split_reads(self, codes: np.ndarray, k: int, sig_len: int = 9, dev: int = 0):
        """stage-1 test hook: code stream (int8, negative = N / read boundary) -> (sig per position, sk_pos, sk_len, sk_sig)"""
        codes = np.ascontiguousarray(codes, dtype=np.int8)
        n = codes.size
        sig = np.zeros(max(n, 1), dtype=np.uint32)
        cap = n + 8
        pos = np.zeros(cap, dtype=np.uint64)
        ln = np.zeros(cap, dtype=np.uint32)
        sg = np.zeros(cap, dtype=np.uint32)
        nsk = C.c_uint64()
        self._chk(self.L.kmc_hip_debug_split_reads(self.h, dev, _vp(codes), n, k, sig_len, _vp(sig), _vp(pos), _vp(ln), _vp(sg), cap, C.byref(nsk)))
        j = nsk.value
        return sig[:n], pos[:j].copy(), ln[:j].copy(), sg[:j].copy()

    def split_reads_device(self, d_codes: int, n: int, k: int, sig_len: int, d_sig_to_bin: int, n_bins: int, dev: int = 0):
        """stage 1 on the device (groundwork, include/kmc_hip.h): device code stream -> bins in HBM. Returns dict(d_bins, d_pack_start, base, bytes,
        superkmers, kmers, pack_base); the caller frees d_bins / d_pack_start with free()."""
        base = np.zeros(n_bins + 1, dtype=np.uint64)
        pbase = np.zeros(n_bins + 1, dtype=np.uint64)
        by, sk, km = (np.zeros(n_bins, dtype=np.uint64) for _ in range(3))
        plan = C.c_void_p()
        self.L.kmc_hip_split_reads_free.restype = None
        self._chk(self.L.kmc_hip_split_reads_plan(self.h, dev, C.c_void_p(d_codes), C.c_uint64(n), C.c_uint32(k), C.c_uint32(sig_len), C.c_void_p(d_sig_to_bin),
                                                  C.c_uint32(n_bins), C.byref(plan), _vp(base), _vp(by), _vp(sk), _vp(km), _vp(pbase)))
        d_bins = d_ps = None
        try:
            d_bins = self.malloc(int(base[n_bins]), dev)
            d_ps = self.malloc(int(pbase[n_bins]) * 8, dev)
            self._chk(self.L.kmc_hip_split_reads_emit(self.h, plan, C.c_void_p(d_bins), C.c_void_p(d_ps)))
        except Exception:
            for q in (d_bins, d_ps):
                if q:
                    self.free(q, dev)
            raise
        finally:
            self.L.kmc_hip_split_reads_free(self.h, plan)
        return dict(d_bins=d_bins, d_pack_start=d_ps, base=base, bytes=by, superkmers=sk, kmers=km, pack_base=pbase)

    # ---- device memory helpers
    def malloc(self, nbytes: int, dev: int = 0) -> int:
        p = C.c_void_p()
        self._chk(self.L.kmc_hip_malloc(self.h, dev, nbytes, C.byref(p)))
        return p.value

    def free(self, dptr: int, dev: int = 0):
        self._chk(self.L.kmc_hip_free(self.h, dev, C.c_void_p(dptr)))

    def h2d(self, dptr: int, a: np.ndarray, dev: int = 0):
        a = np.ascontiguousarray(a)
        self._chk(self.L.kmc_hip_memcpy_h2d(self.h, dev, C.c_void_p(dptr), _vp(a), a.nbytes))

    def d2h(self, a: np.ndarray, dptr: int, dev: int = 0):
        self._chk(self.L.kmc_hip_memcpy_d2h(self.h, dev, _vp(a), C.c_void_p(dptr), a.nbytes))

    def synchronize(self, dev: int = 0):
        self._chk(self.L.kmc_hip_synchronize(self.h, dev))

    def process_bin_device(self, p: BinParams, d_image: int, size: int, n_rec: int, d_pack_start: int, n_packs: int, d_out: int,
                           out_capacity: int, d_out_bytes: int, d_lut: int, d_stats: int, sync: bool = True, dev: int = 0):
        self._chk(self.L.kmc_hip_process_bin_device(self.h, dev, C.byref(p), C.c_void_p(d_image), size, n_rec, C.c_void_p(d_pack_start),
                                                    n_packs, C.c_void_p(d_out), out_capacity, C.c_void_p(d_out_bytes),
                                                    C.c_void_p(d_lut), C.c_void_p(d_stats), 1 if sync else 0))

    def sort_records_device(self, d_recs: int, d_tmp: int, n: int, words: int, key_bytes: int, dev: int = 0) -> int:
        res = C.c_void_p()
        self._chk(self.L.kmc_hip_sort_records_device(self.h, dev, C.c_void_p(d_recs), C.c_void_p(d_tmp), n, words, key_bytes, C.byref(res)))
        return res.value

    def last_timings(self, dev: int = 0):
        ms = (C.c_float * 6)()
        self._chk(self.L.kmc_hip_last_timings(self.h, dev, ms))
        return dict(zip(("index", "expand", "hist", "scatter", "compact", "total"), [float(x) for x in ms]))

    def scatter_totals(self, reset: bool = True, dev: int = 0):
        """(launches, summed ms, records moved) of all k_onesweep launches completed on `dev` since the last reset."""
        n, t, k = C.c_uint64(), C.c_double(), C.c_uint64()
        self._chk(self.L.kmc_hip_scatter_totals(self.h, dev, 1 if reset else 0, C.byref(n), C.byref(t), C.byref(k)))
        return n.value, t.value, k.value

    def set_hybrid(self, mode: int) -> int:
        """process-wide sort selection (see include/kmc_hip.h kmc_hip_set_hybrid); returns the previous mode"""
        self.L.kmc_hip_set_hybrid.restype = C.c_int
        return int(self.L.kmc_hip_set_hybrid(C.c_int(mode)))

    def local_sort_totals(self, reset: bool = True, dev: int = 0):
        """dict(launches, ms, records) of the LDS half of the hybrid sort on `dev` since the last reset + process-wide hybrid / redo group counts."""
        n, t, k, h, r = C.c_uint64(), C.c_double(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.kmc_hip_local_sort_totals(self.h, dev, 1 if reset else 0, C.byref(n), C.byref(t), C.byref(k), C.byref(h), C.byref(r)))
        return dict(launches=n.value, ms=t.value, records=k.value, hybrid_groups=h.value, redo_groups=r.value)

    def path_counters(self, dev: int = 0):
        """group counts by path since the last set_hybrid (process-wide) + the tiles / records k_giant_tiles took on `dev` since the context was made:
        dict(rank_count, rank_compact, bucket_count, lsd, giant_tiles, giant_records, indirect); `indirect`: the groups of rank_count whose HBM passes moved
        (key top, record number) pairs instead of records (three words and more)"""
        c = (C.c_uint64 * 8)()
        self._chk(self.L.kmc_hip_path_counters(self.h, dev, c))
        return dict(rank_count=c[0], rank_compact=c[1], bucket_count=c[2], lsd=c[3], giant_tiles=c[4], giant_records=c[5], indirect=c[6])

    def process_bins_device(self, p: BinParams, descs, n_streams: int = 0, dev: int = 0):
        """Enqueue many device-resident bins (ctypes array of BinDesc); returns after enqueueing — call synchronize()."""
        self._chk(self.L.kmc_hip_process_bins_device(self.h, dev, C.byref(p), descs, len(descs), n_streams))

    def order_database_device(self, p: BinParams, descs, out_lut_prefix_len: int, d_out: int, out_capacity: int, d_lut_out: int, dev: int = 0) -> int:
        """All counted k-mers of the bins (device-resident outputs of process_bins_device) as ONE ascending sequence; returns the number of records."""
        n = C.c_uint64()
        self._chk(self.L.kmc_hip_order_database_device(self.h, dev, C.byref(p), descs, len(descs), out_lut_prefix_len, d_out, out_capacity, d_lut_out, C.byref(n)))
        return n.value

    def host_alloc(self, nbytes: int) -> np.ndarray:
        """Pinned host memory as a uint8 array (free with host_free(arr))."""
        p = C.c_void_p()
        self._chk(self.L.kmc_hip_host_alloc(self.h, nbytes, C.byref(p)))
        a = np.ctypeslib.as_array(C.cast(p, u8p), shape=(max(nbytes, 1),))
        return a

    def host_free(self, arr: np.ndarray):
        self._chk(self.L.kmc_hip_host_free(self.h, C.c_void_p(arr.ctypes.data)))

    def device_count(self) -> int:
        return self.L.kmc_hip_device_count()

    def allreduce_stats(self, per_dev: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(per_dev, dtype=np.uint64).copy()
        self._chk(self.L.kmc_hip_allreduce_stats(self.h, a.ctypes.data_as(u64p)))
        return a


# ---- synthetic stage-2 inputs (libkmc_synth.so)
_SYN = None


def _syn():
    global _SYN
    if _SYN is None:
        p = _build.LIB_SYNTH
        if not os.path.exists(p):
            _build.build_synth()
        _SYN = C.CDLL(p)
        _SYN.kmc_synth_bins_range.restype = C.c_void_p
        _SYN.kmc_synth_bins_range.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_int, C.POINTER(C.POINTER(u8p)), C.POINTER(u64p), C.POINTER(u64p),
                                              C.POINTER(C.POINTER(u64p)), C.POINTER(u64p), C.POINTER(u64p)]
        _SYN.kmc_synth_free.argtypes = [C.c_void_p]
        _SYN.kmc_synth_free.restype = None
        _SYN.kmc_synth_chunk_reads.restype = C.c_uint64
        _SYN.kmc_synth_fastq.restype = C.c_uint64
        _SYN.kmc_synth_fastq.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_char_p, C.c_int]
    return _SYN


def synth_chunk_reads() -> int:
    """read ranges given to synth_bins(read_begin=...) must start on a multiple of this"""
    return int(_syn().kmc_synth_chunk_reads())


class SynthBins:
    """Bin images made by libkmc_synth.so, as zero-copy views into its buffers (valid until close())."""

    def __init__(self, handle, bins):
        self._h = handle
        self.bins = bins  # list of (image uint8 view, n_rec, pack_bytes uint64 view, n_super)

    def close(self):
        if self._h:
            self.bins = []
            _syn().kmc_synth_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_bins(seed: int, genome_len: int, n_reads: int, k: int, n_bins: int = 1, read_len: int = 150, err: float = 0.01,
               sig_len: int = 9, n_threads: int = 0, read_begin: int = 0, read_end=None, copy: bool = True):
    """Bins of reads [read_begin, read_end) (default: all n_reads) of the model. copy=True returns a list of
    (image uint8 ndarray, n_rec, pack_bytes uint64 ndarray, n_super) per bin; copy=False returns a SynthBins holding views."""
    S = _syn()
    if read_end is None:
        read_end = n_reads
    imgs, sizes, nrec, packs, npacks, nsup = C.POINTER(u8p)(), u64p(), u64p(), C.POINTER(u64p)(), u64p(), u64p()
    h = S.kmc_synth_bins_range(seed, genome_len, read_begin, read_end, read_len, err, k, sig_len, n_bins, n_threads, C.byref(imgs),
                               C.byref(sizes), C.byref(nrec), C.byref(packs), C.byref(npacks), C.byref(nsup))
    if not h:
        raise ValueError("kmc_synth_bins: bad arguments or out of memory")
    out = []
    for b in range(n_bins):
        sz = sizes[b]
        img = np.ctypeslib.as_array(imgs[b], shape=(sz,)) if sz else np.zeros(0, dtype=np.uint8)
        pk = np.ctypeslib.as_array(packs[b], shape=(npacks[b],)) if npacks[b] else np.zeros(0, dtype=np.uint64)
        if copy:
            img, pk = img.copy(), pk.copy()
        out.append((img, int(nrec[b]), pk, int(nsup[b])))
    if copy:
        S.kmc_synth_free(h)
        return out
    return SynthBins(h, out)


def synth_fastq(path: str, seed: int, genome_len: int, n_reads: int, read_len: int = 150, err: float = 0.01, n_threads: int = 0,
                read_begin: int = 0) -> int:
    """The same reads synth_bins() cuts into bins, as FASTQ (for the reference kmc). Returns bytes written."""
    n = _syn().kmc_synth_fastq(seed, genome_len, read_begin, n_reads, read_len, err, path.encode(), n_threads)
    if not n:
        raise OSError(f"kmc_synth_fastq could not write {path}")
    return int(n)

"""ctypes binding of tests/hipemu/libkmc_emu.so — TEST INFRASTRUCTURE: the kernels of kmc_amd/csrc/kernels.hip.h executed on the
CPU under the emulation of tests/hipemu (no GPU, no libkmc_hip.so). See tests/test_kernels_emulated.py."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")


# Two builds of the same kernel source: the product's tile geometry (512-thread workgroups: hundreds of OS threads per emulated workgroup,
# slow) and a small one (128/256-thread workgroups, 4 KB expand slices) that runs the same code paths ~10x faster. Tests use the small one
# unless they ask for "product".
GEOMETRY_FLAGS = {"small": ["-DCP_BLOCK_THREADS=128", "-DEXP_BLOCK_THREADS=256", "-DEXP_CHUNK_BYTES=4096", "-DRS_BLOCK_THREADS=256", "-DCP_FOLD_CHUNK=256", "-DS1_SK_TILE_N=64", "-DS1_PACK_BYTES_N=16384", "-DS1_SUB_N=2", "-DBS_BLOCK_THREADS=128", "-DBC_BLOCK_THREADS=128", "-DBR_THREADS=128", "-DGT_THREADS=256", "-DGT_MAX_RECORDS_LOG2=11", "-DAR_THREADS=256", "-DBR_MID=192", "-DBD_STRIDE_N=88"], "product": []}
_LIBS = {}


def build(geometry="small", force=False) -> str:
    so = os.path.join(EMU_DIR, f"libkmc_emu_{geometry}.so")
    srcs = [os.path.join(EMU_DIR, "emu_kernels.cpp"), os.path.join(EMU_DIR, "include", "hip", "hip_runtime.h"),
            os.path.join(ROOT, "kmc_amd", "csrc", "kernels.hip.h"), os.path.join(ROOT, "kmc_amd", "csrc", "kmer_ops.h"),
            os.path.join(ROOT, "kmc_amd", "csrc", "stage1_kernels.hip.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        # -fno-gnu-unique / hidden visibility / -Bsymbolic: the two geometries are two builds of the same templates; their static "LDS" arrays and
        # inline variables must not be merged across the libraries when one process loads both
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-w", "-fno-gnu-unique", "-fvisibility=hidden",
                               "-Wl,-Bsymbolic", *GEOMETRY_FLAGS[geometry],
                               "-I", os.path.join(EMU_DIR, "include"), srcs[0], "-o", so])
    return so


def build_mock(force=False) -> str:
    """tests/hipemu/libkmc_hip_mock.so: the entry points the product's dlopen loaders bind, over the stage-2 oracle and the emulated stage-1
    chain (see mock_hip_lib.cpp). Loaded only through an explicit KMC_HIP_LIB."""
    so = os.path.join(EMU_DIR, "libkmc_hip_mock.so")
    srcs = [os.path.join(EMU_DIR, "mock_hip_lib.cpp"), os.path.join(EMU_DIR, "include", "hip", "hip_runtime.h"), os.path.join(ROOT, "oracle", "stage2_oracle.c"),
            os.path.join(ROOT, "oracle", "stage2_oracle.h"), os.path.join(ROOT, "include", "kmc_hip.h")] + [os.path.join(ROOT, "kmc_amd", "csrc", f) for f in
                                                                                                       ("kernels.hip.h", "kmer_ops.h", "stage1_kernels.hip.h", "stage1_chain.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs):
        obj = os.path.join(EMU_DIR, "stage2_oracle_mock.o")
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-c", srcs[2], "-o", obj])
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-w", "-fvisibility=hidden", "-I", os.path.join(EMU_DIR, "include"),
                               srcs[0], obj, "-o", so])
    return so


def rewrite_launches(src: str) -> str:
    """`kernel<tmpl><<<grid, block, lds, stream>>>(args)` -> `HIPEMU_LAUNCH((kernel<tmpl>), grid, block, lds, stream, args)`: the one construct of
    kmc_amd/csrc/kmc_hip.hip a host compiler cannot parse. Everything else of the product's host source is compiled as it is."""
    out, pos = [], 0
    while True:
        i = src.find("<<<", pos)
        if i < 0:
            out.append(src[pos:])
            return "".join(out)
        # the kernel expression: back over an identifier with optional template arguments
        j, depth = i, 0
        while j > 0:
            c = src[j - 1]
            if c == ">":
                depth += 1
            elif c == "<":
                depth -= 1
            elif depth == 0 and not (c.isalnum() or c in "_:"):
                break
            j -= 1
        e = src.find(">>>", i)
        assert e > 0 and src[e + 3] == "(", "launch syntax the rewriter does not know"
        k, d = e + 4, 1
        while d:
            d += {"(": 1, ")": -1}.get(src[k], 0)
            k += 1
        args = src[e + 4:k - 1].strip()
        out.append(src[pos:j] + "HIPEMU_LAUNCH((" + src[j:i] + "), " + src[i + 3:e].strip() + (", " + args if args else "") + ")")
        pos = k


def build_hostlib(geometry="small", force=False) -> str:
    """tests/hipemu/libkmc_hip_emu_<geometry>.so: THE PRODUCT'S HOST LIBRARY (kmc_amd/csrc/kmc_hip.hip: every C-ABI entry point, buffer
    management, launch sequences, error handling) compiled with g++ over the emulated kernels and an emulated HIP runtime
    (tests/hipemu/include/hip/hip_host_api.h). Load it with KMC_HIP_LIB=<path>: kmc_amd.capi, the worker plug-ins' loaders and the -m gpu tests
    then run on a box without a GPU (small inputs: one OS thread per GPU thread)."""
    so = os.path.join(EMU_DIR, f"libkmc_hip_emu_{geometry}.so")
    csrc = os.path.join(ROOT, "kmc_amd", "csrc")
    host_parts = sorted(f for f in os.listdir(csrc) if f.startswith("host_") and f.endswith(".hip.h"))  # kmc_hip.hip's own parts (they hold kernel launches)
    srcs = [os.path.join(csrc, f) for f in ["kmc_hip.hip", "kernels.hip.h", "bucket_sort.hip.h", "arena_sort.hip.h", "order_db.hip.h", "kmer_ops.h", "stage1_kernels.hip.h", "stage1_chain.h"] + host_parts] + [
        os.path.join(ROOT, "include", "kmc_hip.h"), os.path.join(EMU_DIR, "include", "hip", "hip_runtime.h"), os.path.join(EMU_DIR, "include", "hip", "hip_host_api.h"),
        os.path.join(EMU_DIR, "include", "rccl", "rccl.h"), os.path.abspath(__file__)]
    if force or not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs):
        gen_dir = os.path.join(EMU_DIR, "_gen")
        os.makedirs(gen_dir, exist_ok=True)
        gen = os.path.join(gen_dir, f"kmc_hip_emu_{geometry}.cpp")
        with open(srcs[0]) as f:
            text = f.read()
        for part in host_parts:  # inline the parts again: the launch rewriter works on one text
            inc = '#include "%s"' % part
            assert text.count(inc) == 1, part
            line = text[text.index(inc):].split("\n", 1)[0]
            with open(os.path.join(csrc, part)) as f:
                text = text.replace(line, f.read())
        text = rewrite_launches(text)
        with open(gen, "w") as f:
            f.write("/* GENERATED by tests/emu.py build_hostlib from kmc_amd/csrc/kmc_hip.hip: only the kernel launches were rewritten */\n" + text)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-w", "-fno-gnu-unique", "-Wl,-Bsymbolic", "-DHIPEMU_HOST_API",
                               *GEOMETRY_FLAGS[geometry], "-I", os.path.join(EMU_DIR, "include"), "-I", csrc, gen, "-o", so])
    return so


def lib(geometry="small"):
    if geometry not in _LIBS:
        L = C.CDLL(build(geometry))
        L.emu_run.restype = C.c_int
        L.emu_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.emu_lookback.restype = C.c_uint64
        L.emu_lookback.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.emu_group_front.restype = C.c_int
        L.emu_group_front.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.emu_group_compact.restype = C.c_int
        L.emu_group_compact.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.emu_s1_split.restype = C.c_int
        L.emu_s1_split.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.emu_s1_norm_all.argtypes = [C.c_uint32, C.c_void_p]
        L.emu_s1_scatter.restype = C.c_int
        L.emu_s1_scatter.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_uint64]
        L.emu_s1_geometry.argtypes = [C.c_void_p]
        L.emu_s1_scatter_sorted.restype = C.c_int
        L.emu_s1_scatter_sorted.argtypes = L.emu_s1_scatter.argtypes
        L.emu_s1_text_to_codes.restype = C.c_int
        L.emu_s1_text_to_codes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.emu_s1_plus_x.restype = None
        L.emu_s1_plus_x.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        _LIBS[geometry] = L
    return _LIBS[geometry]


def lookback(status: np.ndarray, tile: int, aggregate: int):
    """lookback64 of kernels.hip.h for `tile` over a prepared uint64 status array; returns (exclusive prefix, device error word)"""
    err = C.c_uint32(0)
    r = lib().emu_lookback(status.ctypes.data, tile, aggregate, C.addressof(err))
    return int(r), err.value


def _params(p):
    return np.array([p.kmer_len, p.both_strands, p.cutoff_min, p.without_output, p.cutoff_max, p.counter_max, p.lut_prefix_len, p.output_type, 0, 0],
                    dtype=np.uint32)


def run(p, stage_mask, img=None, n_rec=0, pack_bytes=None, recs=None, out_capacity=None, geometry="small"):
    """p: any struct with the kmc_hip_bin_params fields. Returns dict(err, recs, sorted, out, lut, stats)."""
    words = (p.kmer_len + 31) // 32
    if recs is not None:
        n_rec = recs.shape[0]
    buf = np.zeros((2, max(n_rec, 1), words), dtype=np.uint64)
    if recs is not None:
        buf[0, :n_rec] = recs
    img = np.zeros(0, dtype=np.uint8) if img is None else np.ascontiguousarray(img)
    ps = np.concatenate([[0], np.cumsum(pack_bytes)]).astype(np.uint64) if pack_bytes is not None else np.zeros(1, dtype=np.uint64)
    lut_n = (1 << (2 * p.lut_prefix_len)) if (p.lut_prefix_len and p.output_type == 0) else 0
    rb = max(1, 40 + 8 * words)
    cap = out_capacity if out_capacity is not None else (n_rec + 1) * rb
    out = np.zeros(cap + 64, dtype=np.uint8)
    lut = np.zeros(max(lut_n, 1), dtype=np.uint64)
    stats = np.zeros(4, dtype=np.uint64)
    ob = C.c_uint64(0)
    si = C.c_int(0)
    pr = _params(p)
    err = lib(geometry).emu_run(pr.ctypes.data, stage_mask, img.ctypes.data if img.size else None, img.size, n_rec, ps.ctypes.data, ps.size - 1,
                        buf.ctypes.data, C.addressof(si), out.ctypes.data, cap, C.addressof(ob), lut.ctypes.data, stats.ctypes.data)
    return dict(err=err, recs=buf[0, :n_rec], sorted=buf[si.value, :n_rec], out=out[: ob.value].copy(), lut=lut[:lut_n].copy(), stats=stats)


def _ptr_array(arrays):
    return (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])


def group_front(p, bins, n_pass, geometry="small"):
    """bins: list of (image, n_rec, pack_bytes). Parse + expand of the whole group in one launch each. Returns (err, records [N, words])."""
    words = (p.kmer_len + 31) // 32
    imgs = [np.ascontiguousarray(b[0]) for b in bins]
    sizes = np.array([b[0].size for b in bins], dtype=np.uint64)
    nrec = np.array([b[1] for b in bins], dtype=np.uint64)
    ps = [np.concatenate([[0], np.cumsum(b[2])]).astype(np.uint64) for b in bins]
    npk = np.array([b[2].size for b in bins], dtype=np.uint64)
    recs = np.zeros((int(nrec.sum()), words), dtype=np.uint64)
    pr = _params(p)
    err = lib(geometry).emu_group_front(pr.ctypes.data, len(bins), _ptr_array(imgs), sizes.ctypes.data, nrec.ctypes.data, _ptr_array(ps), npk.ctypes.data,
                                        recs.ctypes.data, n_pass)
    return err, recs


def group_compact(p, sorted_recs, n_recs, geometry="small"):
    """compaction + fold of a group on the bin-major sorted record array. Returns (err, [(out, lut, stats) per bin])."""
    g = len(n_recs)
    words = (p.kmer_len + 31) // 32
    lut_n = (1 << (2 * p.lut_prefix_len)) if (p.lut_prefix_len and p.output_type == 0) else 0
    cap = (max(n_recs) + 1) * (40 + 8 * words)
    outs = [np.zeros(cap + 64, dtype=np.uint8) for _ in range(g)]
    ob = np.zeros(g, dtype=np.uint64)
    luts = np.zeros((g, max(lut_n, 1)), dtype=np.uint64)
    stats = np.zeros((g, 4), dtype=np.uint64)
    nr = np.array(n_recs, dtype=np.uint64)
    srt = np.ascontiguousarray(sorted_recs)
    pr = _params(p)
    err = lib(geometry).emu_group_compact(pr.ctypes.data, g, srt.ctypes.data, nr.ctypes.data, _ptr_array(outs), cap, ob.ctypes.data, luts.ctypes.data, stats.ctypes.data)
    return err, [(outs[i][: int(ob[i])].copy(), luts[i, :lut_n].copy(), stats[i].copy()) for i in range(g)]


def s1_split(codes: np.ndarray, k: int, m: int = 9, geometry="small", fused=True):
    """stage-1 kernels on a code stream (int8: 0..3, negative = invalid/separator). Returns (err, sig per position, sk_pos, sk_len, sk_sig).
    fused: the cutting kernel computes its signatures itself (kmc_hip_split_reads_plan); otherwise it reads k_s1_signatures' output (the test hook)."""
    codes = np.ascontiguousarray(codes, dtype=np.int8)
    n = codes.size
    sig = np.zeros(max(n, 1), dtype=np.uint32)
    cap = n + 8
    pos = np.zeros(cap, dtype=np.uint64)
    ln = np.zeros(cap, dtype=np.uint32)
    sg = np.zeros(cap, dtype=np.uint32)
    nsk = C.c_uint64(0)
    err = lib(geometry).emu_s1_split(codes.ctypes.data, n, k, m, sig.ctypes.data, pos.ctypes.data, ln.ctypes.data,
                                     sg.ctypes.data, cap, C.addressof(nsk), 1 if fused else 0)
    j = nsk.value
    return err, sig[:n], pos[:j].copy(), ln[:j].copy(), sg[:j].copy()


def s1_norm_all(m: int, geometry="small") -> np.ndarray:
    """the kernels' computed normalisation of every m-mer (s1_norm of stage1_kernels.hip.h)"""
    out = np.zeros(1 << (2 * m), dtype=np.uint32)
    lib(geometry).emu_s1_norm_all(m, out.ctypes.data)
    return out


def s1_geometry(geometry="small"):
    """(super-k-mers per scatter tile, pack size in bytes, bin alignment) of the build"""
    g = (C.c_uint32 * 3)()
    lib(geometry).emu_s1_geometry(g)
    return tuple(g)


def s1_scatter(codes, sk_pos, sk_len, sk_sig, k: int, sig_to_bin: np.ndarray, n_bins: int, geometry="small", through_sort=False):
    """stage-1 bin scatter on the super-k-mer list of s1_split. Returns dict(err, base uint64[n_bins+1], totals uint64[3][n_bins] = bytes /
    super-k-mers / k-mers, pack_base uint64[n_bins+1], out bytes, pack_start uint64[...])."""
    codes = np.ascontiguousarray(codes, dtype=np.int8)
    sk_pos = np.ascontiguousarray(sk_pos, dtype=np.uint64)
    sk_len = np.ascontiguousarray(sk_len, dtype=np.uint32)
    sk_sig = np.ascontiguousarray(sk_sig, dtype=np.uint32)
    m = np.ascontiguousarray(sig_to_bin, dtype=np.int32)
    n = sk_pos.size
    _, pack_bytes, align = s1_geometry(geometry)
    base = np.zeros(n_bins + 1, dtype=np.uint64)
    pbase = np.zeros(n_bins + 1, dtype=np.uint64)
    tot = np.zeros((3, n_bins), dtype=np.uint64)
    payload = int(n + ((sk_len.astype(np.uint64) + 3) // 4).sum())
    cap = payload + (n_bins + 2) * align
    out = np.full(cap, 0xEE, dtype=np.uint8)
    pcap = payload // pack_bytes + 2 * n_bins + 2
    pack_start = np.full(pcap, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    fn = lib(geometry).emu_s1_scatter_sorted if through_sort else lib(geometry).emu_s1_scatter
    err = fn(codes.ctypes.data, sk_pos.ctypes.data, sk_len.ctypes.data, sk_sig.ctypes.data, n, k, m.ctypes.data, n_bins, base.ctypes.data,
                                       pbase.ctypes.data, tot.ctypes.data, out.ctypes.data, cap, pack_start.ctypes.data, pcap)
    return dict(err=err, base=base, pack_base=pbase, totals=tot, out=out[: int(base[n_bins])].copy(), pack_start=pack_start[: int(pbase[n_bins])].copy())


def s1_text_to_codes(text: bytes, lines_per_record: int, geometry="small"):
    """one part of FASTA (2) / FASTQ (4) text -> (err, code stream int8, positions of the line ends)"""
    t = np.frombuffer(text, dtype=np.uint8)
    codes = np.zeros(t.size + 16, dtype=np.int8)
    nl = np.zeros(t.size + 16, dtype=np.uint64)
    tot = np.zeros(2, dtype=np.uint64)
    err = lib(geometry).emu_s1_text_to_codes(t.ctypes.data, t.size, lines_per_record, codes.ctypes.data, nl.ctypes.data, nl.size, tot.ctypes.data)
    return err, codes[: int(tot[1])].copy(), nl[: int(tot[0])].copy()


def s1_plus_x(codes, sk_pos, sk_len, sk_sig, k, max_x, both_strands, sig_to_bin, n_bins, geometry="small"):
    codes = np.ascontiguousarray(codes, dtype=np.int8)
    sk_pos = np.ascontiguousarray(sk_pos, dtype=np.uint64)
    sk_len = np.ascontiguousarray(sk_len, dtype=np.uint32)
    sk_sig = np.ascontiguousarray(sk_sig, dtype=np.uint32)
    m = np.ascontiguousarray(sig_to_bin, dtype=np.int32)
    out = np.zeros(n_bins, dtype=np.uint64)
    lib(geometry).emu_s1_plus_x(codes.ctypes.data, sk_pos.ctypes.data, sk_len.ctypes.data, sk_sig.ctypes.data, sk_pos.size, k, max_x, 1 if both_strands else 0,
                                m.ctypes.data, n_bins, out.ctypes.data)
    return out

#!/usr/bin/env python3
"""bench.py — KMC stage-2 (bin sort & count) throughput on MI355X.

Metric (BASELINE.json): stage-2 Gk-mers/s (+ unique k-mers/s), k=27, bit-exact database.

Workload = BASELINE.json configs[2] (the shape the 1/2/4/8-GPU metric is quoted on, SURVEY.md §8d "C3"): k=27, 150 bp synthetic
reads, ~30 Gbp (200 M reads of a 1 Gbp random genome, seed 2026, 1 % substitutions, random strand), cut into 512 signature
bins of minimizer super-k-mers with KMC's parameters for that input (cutoff_min 2, counter_max 255, lut_prefix_len 7 by the
rule of kmc.h:1434-1469) -> ~24.8 G k-mers. One "step" = ALL bins through the whole hot path (parse -> expand (+histograms) ->
7 onesweep passes -> compaction) with the bin images already resident in HBM. Consecutive bins share their scatter passes in groups of 4
(kmc_hip.hip run_group_device_t: the top radix digit of a 27-mer has 2 spare bits that carry the bin's number inside the group).

  --gpus 1 : all 512 bins on one GPU (configs[2]).
  --gpus N : the SAME 512 bins sharded over N ranks (one process per GPU) by LPT on their k-mer counts (the order KMC hands
             bins to sorters, queues.h:499-558), no data-path collective; the four tallies are summed with ONE all-reduce
             (RCCL) inside the timed region. value = all k-mers of the bin set / max-over-ranks time => STRONG scaling (configs[3]).
             Each rank generates 1/N of the reads; bin pieces are exchanged through a scratch directory before the timed region.

What is printed (round 5): the LAST line of stdout is a SHORT record (short_line(): < 6000 bytes — the driver keeps the tail of stdout) with the contract's keys, `roofline`,
`cpu_baseline` and the headline numbers of the legs below; the whole record (prose, per-leg records, worker reports) goes to bench_detail.json next to this file (and under
gpurun_out/ when that exists), a few human lines to stderr.

Keys of the N=1 record (each measured after the timed region, none inside it):
  value_two_streams   : the same step with two bins in flight (tails and launch gaps of one bin filled by the other)
  value_host_boundary : the same bins from pinned host memory (PCIe inclusive) through kmc_hip_process_bin_submit/_wait (one bin per call) and through
                        kmc_hip_process_bins_submit/_wait (4 bins per call): the better of the two, both in host_boundary.legs
  secondary.single_bin: configs[1] — 2 Gbp, all k-mers as ONE bin (the kernel-level datum of round 1)
  secondary.bins512_2gbp: the 2 Gbp sample cut into 512 bins (3.2 M k-mers per bin), tallies checked against the reference
  secondary.stage1_groundwork: NOT stage 2 — the splitter groundwork of DESIGN.md 9 (codes in HBM -> bins in HBM), timed by tools/s1_bench.py
  e2e_stage1          : NOT stage 2 — "1st stage" seconds of the reference pipeline with the splitter worker swapped too (kmc_hip_s1)
  secondary.skew_quarter / skew_spectrum_quarter: the quarter workload with one repeat family / a spectrum of families planted in the genome ($KMC_SYNTH_REPEATS), value on one
                        stream and value_two_streams
  secondary.k55_full / k127_full: configs[4]'s record widths at FULL size on one GPU (200 M reads, 512 bins); their `roofline` is the kernel that dominates them — the finisher
                        k_bucket_rank<2|4> (pair read + record gathered by number) —, the scatter passes' own in roofline_scatter; 16 bins each against the oracle
  cpu_baseline_skew / cpu_baseline_spectrum: the reference's "2nd stage" on a FASTQ of the skew legs' reads (one run each), the GPU's value on the same reads beside it;
                        gpu_over_cpu: the three ratios (uniform, skew, spectrum) side by side
  moved_bytes_per_kmer: {"pmc": from the committed rocprofv3 --pmc run of this workload (profiles/r06/pmc_hbm_traffic.json: every kernel of one step), "design": what the path
                        is designed to move}; moved_frac_of_hbm_peak likewise. Neither is a counter of THIS run (counters cannot be read from inside it); roofline.traffic idem.
  self_check.oracle_bins: 16 bins of the timed run, stratified by size, byte for byte against the oracle (one host thread per bin)
  logical_devices     : (--logical 8; off by default) the resident bins LPT-sharded over 8 LOGICAL devices of the one GPU (a context naming it 8 times), one enqueueing host thread each: the control flow of
                        --gpus 8 inside one process; value ~ `value` means the scheduling around the kernels costs nothing; enqueue_ms_per_bin = host time per bin. Not a scaling number.
  e2e_large           : ONE FASTQ of --e2e-gbp Gbp (default 8): reference vs drop-in "2nd stage" in RAM-only mode, and the reference's own stage-1 bins (dumped by the drop-in's
                        worker) device-resident, tallies against the reference's statistics
  cpu_baseline / e2e  : the REAL reference (oracle/_ref/kmc, built from /root/reference by oracle/Makefile) and the drop-in
                        (kmc_amd/bin/kmc_hip = reference pipeline + this library) on a FASTQ of the SAME reads as the 2 Gbp sample:
                        "2nd stage" seconds of each, the five statistics compared.
`roofline`: dominant kernel k_onesweep (one launch = one 8-bit LSD pass over one bin); achieved = algorithmic bytes of the
launches in the timed region (2*W = 16 B per record and pass) / their summed duration (HIP events around
every launch, on the launch's own stream); peak = 8 TB/s (MI355X_MICROARCH.md).
"""
import argparse
import ctypes as C
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from kmc_amd import capi, sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
PMC_PROFILE = os.path.join(ROOT, "profiles", "r06", "pmc_hbm_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes

# BASELINE.json configs -> generator parameters
CONFIGS = {
    "configs[2]": dict(reads=200_000_000, genome=1_000_000_000, bins=512,
                       desc="configs[2]: k=%d, 150 bp synthetic reads, ~30 Gbp (200 M reads of a 1 Gbp genome), all 512 signature bins"),
    "configs[1]": dict(reads=13_300_000, genome=66_000_000, bins=1,
                       desc="configs[1]: k=%d, 150 bp synthetic reads, ~2 Gbp, all k-mers as a single bin"),
    "2gbp-512bins": dict(reads=13_300_000, genome=66_000_000, bins=512,
                         desc="2 Gbp sample of the configs[2] read model (13.3 M reads of a 66 Mbp genome, 30x), k=%d, 512 signature bins"),
    "quarter": dict(reads=50_000_000, genome=250_000_000, bins=128,
                    desc="a quarter of configs[2]/[4] (50 M reads of a 250 Mbp genome, 30x: 128 signature bins of the same size as the full run's), k=%d"),
}
SEED = 2026
SKEW_REPEATS = "10000:2000:10"
# a spectrum of repeat families (round 5; kmc_amd/csrc/synth_bins.cpp): an Alu-like 300 bp unit x 100 000 copies at 12 % divergence, an L1-like 6 kbp unit x 5 000 at 2 %, a
# 171 bp satellite x 100 000 at 2 %, one poly-A run of 20 kbp: ~77 Mbp of the quarter workload's 250 Mbp genome
SKEW_SPECTRUM = "300:100000:120,6000:5000:20,171:100000:20,H20000"


def kmc_lut_prefix_len(k: int, n_reads: int, n_bins: int) -> int:
    """Params.lut_prefix_len as CKMC::ProcessStage2_impl picks it without a histogram estimate (kmc.h:1434-1469)."""
    n_est = 4 * n_reads
    best, best_mem = 0, 1 << 62
    for p in range(2, 16):
        if (k - p) % 4 or p >= k:
            continue
        mem = n_est * (k - p) // 4 + n_bins * (1 << (2 * p)) * 8
        if mem < best_mem:
            best, best_mem = p, mem
    return best


def pmc_traffic(kernel: str, records_per_launch: float):
    """HBM bytes per launch of `kernel` from the committed PMC profile of THIS workload (null for any other workload:
    counters cannot be collected from inside the timed run)."""
    try:
        d = json.load(open(PMC_PROFILE))
        if abs(d["records_per_launch_avg"] - records_per_launch) > 0.02 * records_per_launch:
            return None
        return d["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def pmc_moved_bytes_per_kmer(kernel: str, records_per_launch: float, kmers: int):
    """HBM bytes per k-mer that ALL kernels of one step moved, from the same committed PMC profile (one step of this workload under rocprofv3 --pmc: every launch counted):
    sum over kernels of bytes per launch x launches / the k-mers of the step. None for any other workload."""
    try:
        d = json.load(open(PMC_PROFILE))
        if abs(d["records_per_launch_avg"] - records_per_launch) > 0.02 * records_per_launch or kernel not in d["kernels"]:
            return None
        return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in d["kernels"].values()) / kmers
    except Exception:
        return None


def scratch_dir(need_bytes: int):
    cands = [d for d in ("/dev/shm", os.environ.get("TMPDIR") or "/tmp", "/tmp", os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)]
    space = {d: shutil.disk_usage(d).free for d in cands}
    for d in cands:
        if space[d] >= need_bytes:
            return d, space[d]
    best = max(cands, key=lambda d: space[d])
    return best, space[best]


# ---------------------------------------------------------------------------------------------------------------- workload
class Workload:
    """This rank's share of the bin set, resident in HBM, as kmc_hip_bin_desc records."""

    def __init__(self, ctx, p, k):
        self.ctx, self.p, self.k = ctx, p, k
        self.bins = []  # (bin id, size, n_rec, n_packs, n_super) of own bins
        self.allocs = []
        self.total_kmers_all = 0  # over ALL ranks
        self.n_bins_all = 0
        self.setup_s = {}

    def free(self):
        for a in self.allocs:
            self.ctx.free(a)
        self.allocs = []


class DumpedBins:
    """The bins of a real run as kmc_hip's stage-2 worker received them from the reference's stage 1 ($KMC_HIP_BIN_DUMP_DIR, kb_sorter_plugin.h): the interface
    build_workload wants from a bin set. Real bin-size spread (s_mapper.h:143-233 balances by statistics and still leaves a tail), real pack lists."""

    def __init__(self, directory):
        self.timings = {}
        t = time.time()
        self.meta = {}
        for fn in sorted(os.listdir(directory)):
            if fn.endswith(".meta"):
                with open(os.path.join(directory, fn)) as f:
                    b, size, n_rec, n_packs = (int(x) for x in f.readline().split())
                    packs = np.array([int(x) for x in f.read().split()], dtype=np.uint64)
                assert packs.size == n_packs and int(packs.sum()) == size, fn
                self.meta[b] = (size, n_rec, packs, os.path.join(directory, fn[:-5] + ".img"))
        self.own = [b for b in sorted(self.meta) if self.meta[b][1] > 0]  # empty bins carry nothing to sort
        nb = max(self.meta) + 1 if self.meta else 0
        self.size = np.zeros(nb, dtype=np.int64)
        self.n_rec = np.zeros(nb, dtype=np.int64)
        self.n_packs = np.zeros(nb, dtype=np.int64)
        self.n_super = np.zeros(nb, dtype=np.int64)  # not recorded by the dump
        for b, (size, n_rec, packs, _) in self.meta.items():
            self.size[b], self.n_rec[b], self.n_packs[b] = size, n_rec, packs.size
        self.pieces = {b: [(self.image(b), None)] for b in self.own}
        self.timings["load_dump"] = time.time() - t

    def image(self, b):
        return np.fromfile(self.meta[b][3], dtype=np.uint8)

    def packs(self, b):
        return self.meta[b][2]

    def close(self):
        self.pieces = {}


def build_workload(ctx, args, k, p, rank, world, keep_host=False, sb=None):
    """Generate + shard the bin set (kmc_amd/sharding.py), upload this rank's bins. Returns Workload. sb: a prepared bin set instead (DumpedBins)."""
    w = Workload(ctx, p, k)
    n_threads = max(2, 2 * sharding.effective_cpus() // max(world, 1))  # the box may grant fewer CPUs (cgroup quota) than it shows
    if sb is None:
        sb = sharding.generate_sharded_bins(SEED, args.genome, args.reads, k, args.bins, rank, world, n_threads, cache_dir=args.cache or None)
    w.setup_s.update(sb.timings)
    own = sb.own

    t = time.time()
    rec_bytes = ctx.out_rec_bytes(p)
    lut_n = ctx.lut_entries(p)
    in_off, ps_off, out_off = [], [], []
    a = b_ = c = 0
    for b in own:
        in_off.append(a)
        a += (int(sb.size[b]) + 256 + 255) & ~255
        ps_off.append(b_)
        b_ += (int(sb.n_packs[b]) + 1) * 8
        out_off.append(c)
        c += ((((int(sb.n_rec[b]) + 1) // max(p.cutoff_min, 1)) * rec_bytes) + 256 + 255) & ~255
    d_in = ctx.malloc(max(a, 256))
    d_ps = ctx.malloc(max(b_, 256))
    d_out = ctx.malloc(max(c, 256))
    d_lut = ctx.malloc(max(lut_n, 1) * 8 * max(len(own), 1))
    d_small = ctx.malloc(64 * max(len(own), 1))
    w.allocs = [d_in, d_ps, d_out, d_lut, d_small]
    w.d_small, w.d_out, w.d_lut, w.out_off, w.lut_n, w.rec_bytes = d_small, d_out, d_lut, out_off, lut_n, rec_bytes
    w.d_in, w.in_off = d_in, in_off
    descs = (capi.BinDesc * max(len(own), 1))()
    zeros = np.zeros(256, dtype=np.uint8)
    host_imgs = []
    for i, b in enumerate(own):
        off = 0
        for img, _ in sb.pieces[b]:
            if img.size:
                ctx.h2d(d_in + in_off[i] + off, np.ascontiguousarray(img))
            off += int(img.size)
        assert off == int(sb.size[b])
        ctx.h2d(d_in + in_off[i] + off, zeros)
        pk_all = sb.packs(b)
        ps = np.concatenate([[0], np.cumsum(pk_all)]).astype(np.uint64)
        ctx.h2d(d_ps + ps_off[i], ps)
        cap = ((int(sb.n_rec[b]) + 1) // max(p.cutoff_min, 1)) * rec_bytes
        descs[i] = capi.BinDesc(d_in + in_off[i], int(sb.size[b]), int(sb.n_rec[b]), d_ps + ps_off[i], int(sb.n_packs[b]), d_out + out_off[i], cap,
                                d_small + 64 * i + 32, d_lut + lut_n * 8 * i, d_small + 64 * i)
        w.bins.append((b, int(sb.size[b]), int(sb.n_rec[b]), int(sb.n_packs[b]), int(sb.n_super[b])))
        if keep_host:
            host_imgs.append((sb.image(b), pk_all))
    w.setup_s["upload"] = time.time() - t
    w.setup_s["upload_GBs"] = float(sum(x[1] for x in w.bins)) / max(w.setup_s["upload"], 1e-9) / 1e9
    w.descs, w.n_own = descs, len(own)
    w.total_kmers_all = int(np.sum(sb.n_rec))
    w.total_super_all = int(np.sum(sb.n_super))
    w.total_bytes_all = int(np.sum(sb.size))
    w.n_bins_all = args.bins if args is not None else len(own)
    w.own_kmers = int(sum(x[2] for x in w.bins))
    w.host_imgs = host_imgs
    w.sb = sb
    if not keep_host:
        sb.close()
        w.sb = None
    return w


def run_step(ctx, w, n_streams):
    if w.n_own:
        ctx.L.kmc_hip_process_bins_device(ctx.h, 0, C.byref(w.p), w.descs, w.n_own, n_streams)
    ctx.synchronize()


def read_results(ctx, w):
    """per-bin (stats[4], out_bytes) of the last step"""
    small = np.zeros(8 * max(w.n_own, 1), dtype=np.uint64)
    ctx.d2h(small, w.d_small)
    return small.reshape(-1, 8)[: w.n_own]


def output_digest(ctx, w, res):
    """Order-independent digest of everything the step wrote (suffix/counter records + LUTs of every own bin):
    sum over bins of a position-weighted 64-bit checksum, mod 2^64. Equal for any --gpus N."""
    dig = 0
    for i in range(w.n_own):
        ob = int(res[i, 4])
        pad = (-ob) % 8
        buf = np.zeros(ob + pad, dtype=np.uint8)
        if ob:
            ctx.d2h(buf[:ob], w.d_out + w.out_off[i])
        lut = np.zeros(max(w.lut_n, 1), dtype=np.uint64)
        if w.lut_n:
            ctx.d2h(lut, w.d_lut + w.lut_n * 8 * i)
        for arr in (buf.view(np.uint64), lut):
            if arr.size:
                idx = np.arange(1, arr.size + 1, dtype=np.uint64)
                dig = (dig + int(arr.sum(dtype=np.uint64)) + int((arr * idx).sum(dtype=np.uint64)) * 31 + (w.bins[i][0] + 1) * arr.size) & ((1 << 64) - 1)
    return dig


def oracle_check(ctx, w, res, n_check=16):
    """The timed run's own output against the oracle, at the scale that was timed: `n_check` bins of this rank, stratified by size (the smallest, the largest and the
    quantiles between them), are copied back (image, suffix records, LUT, tallies) and compared byte for byte with oracle/stage2_oracle.c's process_bin on the same
    image, one host thread per bin. The oracle is the CHECKER here (tests/oracle_py.py, ctypes over oracle/liboracle_stage2.so); nothing of it is timed or shipped."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O

    order = sorted(range(w.n_own), key=lambda i: w.bins[i][2])
    n_check = max(1, min(n_check, len(order), 2 * sharding.effective_cpus()))
    picks = sorted({order[round(q * (len(order) - 1) / max(n_check - 1, 1))] for q in range(n_check)}, key=lambda i: -w.bins[i][2]) if order else []
    p = w.p
    op = O.make_params(p.kmer_len, p.both_strands, p.cutoff_min, p.cutoff_max, p.counter_max, p.lut_prefix_len, p.output_type, p.without_output)
    verdicts = [None] * len(picks)

    def one(j, i):
        b, size, n_rec = w.bins[i][0], w.bins[i][1], w.bins[i][2]
        img = np.zeros(size, dtype=np.uint8)
        ctx.d2h(img, w.d_in + w.in_off[i])
        ob = int(res[i, 4])
        out = np.zeros(ob, dtype=np.uint8)
        if ob:
            ctx.d2h(out, w.d_out + w.out_off[i])
        lut = np.zeros(max(w.lut_n, 1), dtype=np.uint64)
        if w.lut_n:
            ctx.d2h(lut, w.d_lut + w.lut_n * 8 * i)
        t = time.time()
        w_out, w_lut, w_st = O.process_bin(op, img, n_rec)
        verdicts[j] = {"bin": int(b), "kmers": int(n_rec), "out_bytes": ob, "oracle_s": round(time.time() - t, 2),
                       "equal": bool(np.array_equal(out, w_out) and np.array_equal(lut[: w.lut_n], w_lut) and [int(x) for x in res[i, :4]] == [int(x) for x in w_st])}

    th = [threading.Thread(target=one, args=(j, i)) for j, i in enumerate(picks)]  # the oracle is single-threaded C behind ctypes (the GIL is released)
    for t in th:
        t.start()
    for t in th:
        t.join()
    return verdicts


# ---------------------------------------------------------------------------------------------------------------- logical devices
def logical_devices_leg(dev_index, w, n_logical, tallies, steps=2):
    """configs[3]'s host side on ONE GPU: a context that names this GPU `n_logical` times — each logical device with streams, slots and buffers of its own,
    as physical devices would have —, the resident bins sharded over them by LPT (kmc_amd/sharding.py), one host thread per logical device enqueueing its shard through
    kmc_hip_process_bins_device at the same time as the others. The kernels of all shards share the one GPU, so `value` says what the scheduling around them costs (nothing, if it
    equals the one-device value), enqueue_ms_per_bin what a bin costs its host thread; tallies must equal the timed run's. Not a scaling number."""
    c = capi.Context((dev_index,) * n_logical)
    try:
        shards = sharding.lpt_assign([b[2] for b in w.bins], n_logical)
        descs = []
        for sh in shards:
            arr = (capi.BinDesc * max(len(sh), 1))()
            for j, i in enumerate(sh):
                arr[j] = w.descs[i]
            descs.append(arr)
        enq = [0.0] * n_logical
        errs = []

        def rank(r):
            try:
                t = time.perf_counter()
                if shards[r]:
                    rc = c.L.kmc_hip_process_bins_device(c.h, r, C.byref(w.p), descs[r], len(shards[r]), 0)
                    if rc:
                        raise RuntimeError(c.L.kmc_hip_last_error(c.h).decode())
                enq[r] += time.perf_counter() - t
            except Exception as e:  # noqa: BLE001
                errs.append("logical device %d: %r" % (r, e))

        def step():
            ths = [threading.Thread(target=rank, args=(r,)) for r in range(n_logical)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            for r in range(n_logical):
                c.synchronize(r)

        step()
        enq = [0.0] * n_logical
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        dt = (time.perf_counter() - t0) / steps
        if errs:
            return {"error": "; ".join(errs)[-400:]}
        res = read_results(c, w)
        got = res[:, :4].sum(axis=0, dtype=np.uint64)
        loads = [sum(w.bins[i][2] for i in sh) for sh in shards]
        return {"n_logical": n_logical, "value": w.total_kmers_all / dt / 1e9, "ms_per_step": dt * 1e3, "tallies_equal_timed_run": [int(x) for x in got] == [int(x) for x in tallies],
                "enqueue_ms_per_bin": 1e3 * sum(enq) / steps / max(w.n_own, 1), "enqueue_ms_per_step_max_over_threads": 1e3 * max(enq) / steps,
                "lpt_imbalance": max(loads) / (sum(loads) / n_logical) if sum(loads) else None,
                "what": "the resident bins LPT-sharded over %d logical devices of this one GPU, one enqueueing host thread each (the control flow of --gpus %d inside one process); "
                        "the kernels share the GPU: value ~ the one-device value means the scheduling costs nothing" % (n_logical, n_logical)}
    finally:
        c.close()


# ---------------------------------------------------------------------------------------------------------------- host boundary
def host_boundary_pass(ctx, w, n_threads=8, passes=2, group=1):
    """All own bins through the host boundary from PINNED host memory: thread t owns stream slots 2t, 2t+1 and keeps two calls in flight (H2D of call
    j+1 under the kernels of call j, D2H at wait time). group == 1: kmc_hip_process_bin_submit/_wait, one bin per call; group > 1: `group` consecutive
    bins per kmc_hip_process_bins_submit/_wait call (sorted together on the device). Returns (best seconds, tallies, seconds to pin, threads)."""
    from kmc_amd.capi import HostBin

    p, L = w.p, ctx.L
    n_slots = L.kmc_hip_num_slots()
    n_threads = max(1, min(n_threads, n_slots // 2))
    total = sum(((x[1] + 255) & ~255) for x in w.bins) + 256
    t = time.time()
    pin = ctx.host_alloc(total)
    offs, o = [], 0
    for (img, pk), meta in zip(w.host_imgs, w.bins):
        pin[o:o + img.size] = img
        offs.append(o)
        o += (img.size + 255) & ~255
    cap_max = max([((x[2] + 1) // max(p.cutoff_min, 1)) * w.rec_bytes for x in w.bins] + [1])
    outs = [ctx.host_alloc(cap_max + 256) for _ in range(2 * n_threads * group)]
    luts = [ctx.host_alloc(max(w.lut_n, 1) * 8) for _ in range(2 * n_threads * group)]
    t_pin = time.time() - t
    errors = []

    def worker(tid, acc):
        calls = [list(range(w.n_own))[c:c + group] for c in range(tid * group, w.n_own, n_threads * group)]
        inflight = []  # (slot 0/1, bins of the call) in submission order, at most two
        ob, st = (C.c_uint64 * group)(), (C.c_uint64 * (4 * group))()

        def wait(sl, n):
            if group == 1:
                rc = L.kmc_hip_process_bin_wait(ctx.h, 0, 2 * tid + sl, ob, st)
            else:
                rc = L.kmc_hip_process_bins_wait(ctx.h, 0, 2 * tid + sl, ob, st)
            if rc:
                raise RuntimeError(L.kmc_hip_last_error(ctx.h).decode())
            for j in range(n):
                for q in range(4):
                    acc[q] += st[4 * j + q]
                acc[4] += ob[j]

        try:
            for j, mine in enumerate(calls):
                sl = j & 1
                if len(inflight) == 2:
                    wait(*inflight.pop(0))  # == sl: the slot this call is about to reuse
                arr = (HostBin * group)()
                for q, i in enumerate(mine):
                    b, size, n_rec, n_packs, _ = w.bins[i]
                    pk = w.host_imgs[i][1]
                    cap = ((n_rec + 1) // max(p.cutoff_min, 1)) * w.rec_bytes
                    buf = (2 * tid + sl) * group + q
                    arr[q] = HostBin(pin.ctypes.data + offs[i], size, n_rec, pk.ctypes.data, pk.size, outs[buf].ctypes.data, cap, luts[buf].ctypes.data)
                if group == 1:
                    h = arr[0]
                    rc = L.kmc_hip_process_bin_submit(ctx.h, 0, 2 * tid + sl, C.byref(p), C.c_void_p(h.superkmers), h.size, h.n_rec, C.c_void_p(h.pack_bytes), h.n_packs,
                                                      C.c_void_p(h.out_suffix), h.out_capacity, C.c_void_p(h.lut))
                else:
                    rc = L.kmc_hip_process_bins_submit(ctx.h, 0, 2 * tid + sl, C.byref(p), arr, len(mine))
                if rc:
                    raise RuntimeError(L.kmc_hip_last_error(ctx.h).decode())
                inflight.append((sl, len(mine)))
            while inflight:
                wait(*inflight.pop(0))
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d, call %d of %d (bins %s), %.2f s into the pass: %r" % (tid, j, len(calls), mine, time.perf_counter() - t_pass[0], e))

    best, tallies = None, None
    t_pass = [0.0]
    for _ in range(passes):
        t_pass[0] = time.perf_counter()
        accs = [[0, 0, 0, 0, 0] for _ in range(n_threads)]
        ths = [threading.Thread(target=worker, args=(t_, accs[t_])) for t_ in range(n_threads)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        if errors:
            break
        if best is None or dt < best:
            best = dt
        tallies = [sum(a[q] for a in accs) for q in range(5)]
    for a in outs + luts:
        ctx.host_free(a)
    ctx.host_free(pin)
    if errors:
        raise RuntimeError("; ".join(errors))
    return best, tallies, t_pin, n_threads



# ---------------------------------------------------------------------------------------------------------------- the line
SHORT_LINE_LIMIT = 6000  # bytes: the driver reads the tail of stdout; round 4's 22 KB line did not fit it and the record was parsed: null


def _pick(d, keys):
    return {k_: d[k_] for k_ in keys if isinstance(d, dict) and k_ in d}


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def short_line(out):
    """The ONE line of stdout: the contract's keys + roofline + cpu_baseline and a handful of headline numbers; everything else (prose, secondary legs, worker
    reports, e2e timelines) is in bench_detail.json. Built by selection, then checked against SHORT_LINE_LIMIT; optional blocks go first when it does not fit."""
    s = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    s["metric"] = s["metric"].split(" (")[0]
    if out.get("rehearsal"):
        s["rehearsal"] = True
        s["backend_kind"] = out.get("backend_kind")
    s["config"] = _pick(out["config"], ("workload", "kmers", "bins", "bins_rank0", "kmers_rank0", "record_bytes", "lut_prefix_len", "bins_per_sort", "parallelism"))
    s["roofline"] = _pick(out["roofline"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"))
    s["roofline"]["launches"] = out["roofline"].get("launches_in_timed_region")
    s["roofline"]["records_per_launch"] = out["roofline"].get("records_per_launch")
    cb = out.get("cpu_baseline")
    if cb:
        s["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "stage2_s", "input_kmers", "runs", "statistic"))
        s["cpu_baseline"]["gpu_input_kmers"] = out["config"]["kmers"]
    else:
        s["cpu_baseline"] = None
    s["unique_kmers_per_s"] = out.get("unique_kmers_per_s")
    s["tallies"] = out.get("tallies")
    s["value_host_boundary"] = out.get("value_host_boundary")
    s["value_two_streams"] = out.get("value_two_streams") if isinstance(out.get("value_two_streams"), float) else None
    sc = out.get("self_check", {})
    s["self_check"] = _pick(sc, ("per_bin_total_and_out_bytes_consistent", "oracle_bins_equal", "output_digest", "gpu_tallies_equal_reference_on_2gbp_sample"))
    sp = out.get("sort_path", {})
    s["moved_bytes_per_kmer"] = {"pmc": sp.get("hbm_bytes_per_kmer_moved_pmc"), "design": sp.get("hbm_bytes_per_kmer_moved_by_design")}
    s["moved_frac_of_hbm_peak"] = {"pmc": sp.get("moved_frac_of_hbm_peak_pmc"), "design": sp.get("moved_frac_of_hbm_peak")}
    s["stage2_algorithmic_bytes_per_kmer"] = out.get("stage2_algorithmic_bytes_per_kmer")
    s["stage2_frac_of_hbm_peak"] = out.get("stage2_frac_of_hbm_peak")
    s["groups_by_path"] = sp.get("groups_by_path")
    ls = out.get("local_sort", {})
    s["local_sort"] = _pick(ls, ("avg_launch_ms", "records_per_launch", "redo_groups"))
    if out["n_gpus"] == 1:
        s["multi_gpu"] = "no scaling curve measured in this run (n_gpus == 1); the N-rank body is rehearsed over gloo in tests/test_sharding_cpu.py"
    else:
        s["per_rank"] = out.get("per_rank")
        s["lpt_imbalance"] = out.get("lpt_imbalance")
    opt = {}
    sec = out.get("secondary") or {}
    for name_ in sec:
        v = sec[name_]
        if isinstance(v, dict):
            e = {"value": v.get("value"), "ms_per_step": v.get("ms_per_step")}
            if isinstance(v.get("value_two_streams"), float):
                e["value_two_streams"] = v["value_two_streams"]
            if v.get("error"):
                e = {"error": str(v["error"])[-120:]}
            eq = (v.get("self_check") or {}).get("oracle_bins_equal")
            if eq is not None:
                e["oracle_bins_equal"] = eq
            if isinstance(v.get("roofline"), dict):
                e["roofline_frac"] = v["roofline"].get("frac")
                if v["roofline"].get("kernel") != "k_onesweep<1>":
                    e["roofline_kernel"] = v["roofline"].get("kernel")
            opt[name_] = e
    if opt:
        s["secondary"] = opt
    for kk in ("cpu_baseline_k55", "cpu_baseline_k127"):
        if isinstance(out.get(kk), dict):
            s[kk] = _pick(out[kk], ("value", "cores", "stage2_s", "error"))
    for kk in ("cpu_baseline_skew", "cpu_baseline_spectrum"):
        if isinstance(out.get(kk), dict):
            s[kk] = _pick(out[kk], ("value", "cores", "stage2_s", "input_kmers", "gpu_value_same_reads", "gpu_over_cpu", "error"))
    if isinstance(out.get("logical_devices"), dict):
        s["logical_devices"] = _pick(out["logical_devices"], ("n_logical", "value", "ms_per_step", "enqueue_ms_per_bin", "tallies_equal_timed_run", "lpt_imbalance", "error"))
    if isinstance(out.get("gpu_over_cpu"), dict):
        s["gpu_over_cpu"] = _pick(out["gpu_over_cpu"], ("uniform", "skew", "spectrum"))
    if isinstance(out.get("e2e"), dict):
        s["e2e"] = _pick(out["e2e"], ("ref_stage2_s", "hip_stage2_s", "speedup", "hip_Gkmers_per_s", "stats_equal"))
    if isinstance(out.get("e2e_large"), dict):
        s["e2e_large"] = _pick(out["e2e_large"], ("kmers", "ref_stage2_s", "hip_stage2_s", "speedup", "hip_Gkmers_per_s", "ref_Gkmers_per_s", "stats_equal", "error"))
        if "input" in out["e2e_large"]:
            s["e2e_large"]["input"] = out["e2e_large"]["input"].split(",")[0]
        dr = out["e2e_large"].get("device_resident_on_reference_bins")
        if isinstance(dr, dict):
            s["e2e_large"]["device_resident_on_reference_bins"] = _pick(dr, ("value", "ms_per_step", "bins", "tallies_equal_reference_statistics", "bin_kmers_min_median_max"))
    s["detail"] = "bench_detail.json (next to bench.py; also gpurun_out/ when that directory exists)"

    def rnd(o):
        if isinstance(o, dict):
            return {k_: rnd(v) for k_, v in o.items()}
        if isinstance(o, list):
            return [rnd(v) for v in o]
        if isinstance(o, float):
            return float("%.6g" % o)
        return o

    s = rnd(s)
    for drop in ("logical_devices", "e2e", "cpu_baseline_k127", "cpu_baseline_k55", "e2e_large", "secondary", "cpu_baseline_spectrum", "cpu_baseline_skew", "groups_by_path", "local_sort", "tallies", "per_rank"):
        if len(json.dumps(s)) < SHORT_LINE_LIMIT:
            break
        s.pop(drop, None)
        s["dropped_for_length"] = s.get("dropped_for_length", []) + [drop]
    return s


def emit(out):
    """Rank 0: the whole record -> bench_detail.json (+ gpurun_out/), a few human lines -> stderr, then the short line as the LAST line of stdout."""
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    json.dump(out, f, indent=1)
            except OSError:
                pass
    s = short_line(out)
    line = json.dumps(s)
    assert len(line) < SHORT_LINE_LIMIT, len(line)
    if not out.get("rehearsal"):
        rf = s["roofline"]
        print("[bench] %s: %.2f %s, %.1f ms/step; %s %.0f GB/s = %.3f of the HBM peak (%.1f us per launch)" % (
            s["config"]["workload"][:60], s["value"], s["unit"], s["ms_per_step"], rf["kernel"], rf["achieved"], rf["frac"], 1e3 * (rf["avg_launch_ms"] or 0)), file=sys.stderr)
        for name_, v in (s.get("secondary") or {}).items():
            print("[bench]   %s: %s" % (name_, json.dumps(v)), file=sys.stderr)
        if s.get("cpu_baseline"):
            print("[bench]   cpu_baseline %s Gk-mers/s on %s cores" % (s["cpu_baseline"].get("value"), s["cpu_baseline"].get("cores")), file=sys.stderr)
    sys.stderr.flush()
    print(line, flush=True)


# ---------------------------------------------------------------------------------------------------------------- reference legs
_STAT_PATTERNS = {
    "below_min": r"No\. of k-mers below min\. threshold\s*:\s*(\d+)",
    "above_max": r"No\. of k-mers above max\. threshold\s*:\s*(\d+)",
    "unique": r"No\. of unique k-mers\s*:\s*(\d+)",
    "unique_counted": r"No\. of unique counted k-mers\s*:\s*(\d+)",
    "total": r"Total no\. of k-mers\s*:\s*(\d+)",
}


def _run_kmc(exe, flags, fq, td, tag, env=None, timeout=None):
    tmp = os.path.join(td, "tmp_" + tag)
    os.makedirs(tmp, exist_ok=True)
    r = subprocess.run([exe, *flags, fq, os.path.join(td, "db_" + tag), tmp], capture_output=True, text=True, env=env, timeout=timeout)
    shutil.rmtree(tmp, ignore_errors=True)
    if r.returncode != 0:
        raise RuntimeError(f"{os.path.basename(exe)} failed: {(r.stdout + r.stderr)[-600:]}")
    m1 = re.search(r"1st stage:\s*([0-9.eE+-]+)s", r.stdout)
    m2 = re.search(r"2nd stage:\s*([0-9.eE+-]+)s", r.stdout)
    stats = {k: int(re.search(v, r.stdout).group(1)) for k, v in _STAT_PATTERNS.items() if re.search(v, r.stdout)}
    verbose = [ln for ln in r.stderr.splitlines() if ln.startswith("[kmc_hip")]
    return float(m1.group(1)), float(m2.group(1)), stats, verbose


def reference_legs(k: int, reads: int, genome: int, runs: int = 3):
    """cpu_baseline + e2e on a FASTQ of the SAME reads as the 2 Gbp sample (kmc_amd/csrc/synth_bins.cpp writes both)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "kmc")
    hip = os.path.join(ROOT, "kmc_amd", "bin", "kmc_hip")
    cores = sharding.effective_cpus()  # what the container may really use (cgroup quota), not the hardware threads it shows
    threads = min(os.cpu_count() or 1, 128)
    ram_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") >> 30
    mem = max(2, min(128, ram_gb // 2))
    need = int(reads * 316 * 2.0) + (1 << 28)
    d, free = scratch_dir(need)
    if free < need:
        reads = max(int(reads * free / need * 0.9), 100_000)
    out = {}
    with tempfile.TemporaryDirectory(dir=d) as td:
        fq = os.path.join(td, "s.fq")
        t = time.time()
        capi.synth_fastq(fq, seed=SEED, genome_len=genome, n_reads=reads)
        t_fq = time.time() - t
        sample = f"{reads} reads x150bp of a {genome} bp genome (seed {SEED}; the reads of the 2 Gbp sample), FASTQ written in {t_fq:.1f} s"
        # 3 runs, median. The thread count is picked first (one run with every hardware thread, one with the CPUs the cgroup quota really grants);
        # the better count then gets its three runs — the first of them is the probe run itself.
        probe = {threads: _run_kmc(ref, [f"-k{k}", f"-t{threads}", f"-m{mem}", "-hp"], fq, td, "ref_p0")}
        t_alt = max(2, min(threads, cores))
        if t_alt != threads:
            probe[t_alt] = _run_kmc(ref, [f"-k{k}", f"-t{t_alt}", f"-m{mem}", "-hp"], fq, td, "ref_p1")
        threads_used = min(probe, key=lambda t_: probe[t_][1])
        ref_runs = [probe[threads_used]] + [_run_kmc(ref, [f"-k{k}", f"-t{threads_used}", f"-m{mem}", "-hp"], fq, td, f"ref{i}") for i in range(2)]
        s1, s2, st, _ = sorted(ref_runs, key=lambda x: x[1])[1]
        out["cpu_baseline"] = {"value": st["total"] / s2 / 1e9, "unit": "Gk-mers/s", "cores": cores, "kind": "reference",
                               "sample": f"reference kmc 3.2.4 -k{k} -t{threads_used} -m{mem} ({os.cpu_count()} hw threads visible, {cores} usable), "
                                         f"'2nd stage' wall, median of 3; {reads} reads x150bp of a {genome} bp genome = {st['total']} k-mers "
                                         f"(the GPU value is on the full workload: see gpu_input_kmers)",
                               "input_kmers": st["total"], "runs": 3, "statistic": "median", "all_stage2_s": [x[1] for x in ref_runs],
                               "stage2_s": s2, "stage1_s": s1, "unique_kmers_per_s": st["unique"] / s2, "stats": st, "fastq": sample}
        # the record widths of configs[4] beside their own reference timing on this host (same FASTQ, one run each: kmc_CLI/kmc.cpp:343-344 prints "2nd stage")
        for kk in ((55, 127) if k == 27 else ()):
            try:
                w1, w2, wst, _ = _run_kmc(ref, [f"-k{kk}", f"-t{threads_used}", f"-m{mem}", "-hp"], fq, td, f"ref_k{kk}", timeout=600)
                out[f"cpu_baseline_k{kk}"] = {"value": wst["total"] / w2 / 1e9, "unit": "Gk-mers/s", "cores": cores, "kind": "reference",
                                              "sample": f"reference kmc 3.2.4 -k{kk} -t{threads_used} -m{mem}, '2nd stage' wall, one run; {sample} = {wst['total']} k-mers",
                                              "stage2_s": w2, "stage1_s": w1, "unique_kmers_per_s": wst["unique"] / w2}
            except Exception as e:  # noqa: BLE001
                out[f"cpu_baseline_k{kk}"] = {"error": repr(e)[-300:]}
        if os.path.exists(hip):
            env = dict(os.environ, KMC_HIP_LIB=capi.lib_path(), KMC_HIP_VERBOSE="1")
            hip_runs = [_run_kmc(hip, [f"-k{k}", f"-t{threads}", f"-m{mem}", "-sr16", "-hp"], fq, td, f"hip{i}", env) for i in range(runs)]
            h1, h2, hst, verbose = sorted(hip_runs, key=lambda x: x[1])[len(hip_runs) // 2]  # median, like the reference's
            out["e2e"] = {"what": "'2nd stage' seconds of the reference's own pipeline on the same FASTQ: unmodified (oracle/_ref/kmc) vs with the "
                                  "stage-2 worker and bin reader swapped for this library (kmc_amd/bin/kmc_hip -sr16); stage 1, arena, completer and "
                                  "database writer are the reference's in both",
                          "ref_stage2_s": s2, "hip_stage2_s": h2, "speedup": s2 / h2, "hip_Gkmers_per_s": hst["total"] / h2 / 1e9,
                          "ref_Gkmers_per_s": st["total"] / s2 / 1e9, "stats_equal": hst == st, "hip_stats": hst, "hip_stage1_s": h1,
                          "all_hip_stage2_s": [x[1] for x in hip_runs], "all_ref_stage2_s": [x[1] for x in ref_runs], "worker_report": verbose}
            # Stage 1 on the GPU as well: informative, in its own try — kmc_hip_split_part had run under emulation only when this
            # was committed, and nothing here may cost the stage-2 line.
            hip_s1 = os.path.join(ROOT, "kmc_amd", "bin", "kmc_hip_s1")
            if os.path.exists(hip_s1):
                try:
                    g1, g2, gst, gverbose = _run_kmc(hip_s1, [f"-k{k}", f"-t{threads}", f"-m{mem}", "-sr16", "-hp"], fq, td, "hips1", env, timeout=300)
                    out["e2e_stage1"] = {"what": "'1st stage' seconds of the same pipeline with the splitter worker swapped too (kmc_amd/bin/kmc_hip_s1: parts of "
                                                 "FASTQ text through kmc_hip_split_part); readers, storer and bin files are the reference's",
                                         "ref_stage1_s": s1, "hip_stage1_s": g1, "speedup": s1 / g1 if g1 else None, "hip_stage2_s": g2, "stats_equal": gst == st,
                                         "workers": sum(1 for ln in gverbose if "stage 1" in ln),
                                         "parts_through_the_engine": sum(int(m.group(1)) for ln in gverbose for m in [re.search(r"worker: (\d+) parts", ln)] if m),
                                         "engine_s_summed_over_workers": sum(float(m.group(1)) for ln in gverbose for m in [re.search(r"\(([0-9.]+) s inside\)", ln)] if m),
                                         "uncovered_parts": sum(int(m.group(1)) for ln in gverbose for m in [re.search(r"(\d+) uncovered parts", ln)] if m),
                                         "worker_report": [ln for ln in gverbose if "stage 1" in ln and "worker: 0 parts" not in ln][:3]}
                except Exception as e:  # noqa: BLE001
                    out["e2e_stage1"] = {"error": repr(e)[-600:]}
    return out


def reference_repeat_legs(k: int, threads: int, gpu_legs: dict):
    """cpu_baseline_skew / cpu_baseline_spectrum: the REAL reference's "2nd stage" (kmc_CLI/kmc.cpp:390-391) on a FASTQ of the reads the skew legs' bins were
    cut from — the quarter workload (50 M reads x 150 bp of a 250 Mbp genome, seed 2026) with the same $KMC_SYNTH_REPEATS planted —, one run each, RAM-only mode off (the
    default KMC run, as cpu_baseline). The GPU's device-resident value on the same reads is put beside it (speed-up on the inputs where the GPU is weakest)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "kmc")
    cfg = CONFIGS["quarter"]
    ram_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") >> 30
    mem = max(2, min(128, ram_gb // 2))
    cores = sharding.effective_cpus()
    out = {}
    for key, spec, gpu_key in (("cpu_baseline_skew", SKEW_REPEATS, "skew_quarter"), ("cpu_baseline_spectrum", SKEW_SPECTRUM, "skew_spectrum_quarter")):
        try:
            need = int(cfg["reads"] * 316 * 1.3) + (1 << 30)
            d, free = scratch_dir(need)
            if free < need:
                raise RuntimeError(f"no scratch directory with {need >> 30} GiB free for the FASTQ")
            with tempfile.TemporaryDirectory(dir=d) as td:
                fq = os.path.join(td, "r.fq")
                os.environ["KMC_SYNTH_REPEATS"] = spec
                try:
                    capi.synth_fastq(fq, seed=SEED, genome_len=cfg["genome"], n_reads=cfg["reads"])
                finally:
                    del os.environ["KMC_SYNTH_REPEATS"]
                s1, s2, st, _ = _run_kmc(ref, [f"-k{k}", f"-t{threads}", f"-m{mem}", "-hp"], fq, td, key, timeout=900)
            gv = (gpu_legs.get(gpu_key) or {}).get("value")
            out[key] = {"value": st["total"] / s2 / 1e9, "unit": "Gk-mers/s", "cores": cores, "kind": "reference", "stage2_s": s2, "stage1_s": s1, "input_kmers": st["total"],
                        "sample": f"reference kmc 3.2.4 -k{k} -t{threads} -m{mem}, '2nd stage' wall, one run; {cfg['reads']} reads x150bp of a {cfg['genome']} bp genome with "
                                  f"$KMC_SYNTH_REPEATS={spec} (the reads of secondary.{gpu_key})",
                        "gpu_value_same_reads": gv, "gpu_over_cpu": (gv / (st["total"] / s2 / 1e9)) if isinstance(gv, float) else None, "stats": st}
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": repr(e)[-300:]}
    return out


def e2e_large_leg(ctx, k: int, gbp: float, budget_s: float = 1500.0):
    """The drop-in inside the reference's pipeline at a size where the pipeline, not the start-up, is what is timed (round 4's e2e ran 2 Gbp: 0.25 s). One FASTQ of
    `gbp` Gbp (30x of a random genome, the read model of configs[2]); ONE run each of the unmodified reference (oracle/_ref/kmc) and of the drop-in (kmc_amd/bin/kmc_hip:
    the reference's stage 1, reader / worker / completer plug-ins over this library), RAM-only mode (-r) for both; then the bins exactly as the reference's stage 1
    produced them (a third run of the drop-in dumps what its worker received: $KMC_HIP_BIN_DUMP_DIR) go through kmc_hip_process_bins_device device-resident —
    real bin-size spread instead of synth_bins' near-equal bins — and their tallies must equal the reference's five statistics."""
    ref = os.path.join(ROOT, "oracle", "_ref", "kmc")
    hip = os.path.join(ROOT, "kmc_amd", "bin", "kmc_hip")
    if not (os.path.exists(ref) and os.path.exists(hip)):
        return {"error": "oracle/_ref/kmc or kmc_amd/bin/kmc_hip not built"}
    t_leg = time.time()
    reads = int(gbp * 1e9 / 150)
    cores = sharding.effective_cpus()
    threads = max(2, min(os.cpu_count() or 1, 128, cores))
    ram_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") >> 30
    mem = max(4, min(512, ram_gb // 3))
    need = int(reads * 316 * 1.1) + int(reads * 150 * 1.4) + (1 << 30)  # FASTQ + the dumped bins
    d, free = scratch_dir(need)
    if free < need:
        reads = max(int(reads * free / need * 0.9), 1_000_000)
    genome = reads * 150 // 30
    out = {"input": f"{reads} reads x150bp = {reads * 150 / 1e9:.1f} Gbp, 30x of a {genome} bp random genome (seed {SEED + 1}), FASTQ in {d}", "kmc_flags": f"-k{k} -t{threads} -m{mem} -r"}
    with tempfile.TemporaryDirectory(dir=d) as td:
        fq = os.path.join(td, "large.fq")
        t = time.time()
        capi.synth_fastq(fq, seed=SEED + 1, genome_len=genome, n_reads=reads)
        out["fastq_s"] = time.time() - t
        flags = [f"-k{k}", f"-t{threads}", f"-m{mem}", "-r", "-hp"]
        r1, r2, rst, _ = _run_kmc(ref, flags, fq, td, "lref", timeout=budget_s)
        env = dict(os.environ, KMC_HIP_LIB=capi.lib_path(), KMC_HIP_VERBOSE="1")
        h1, h2, hst, verbose = _run_kmc(hip, flags + ["-sr16"], fq, td, "lhip", env, timeout=budget_s)
        out.update(kmers=rst["total"], ref_stage1_s=r1, ref_stage2_s=r2, hip_stage1_s=h1, hip_stage2_s=h2, speedup=r2 / h2, ref_Gkmers_per_s=rst["total"] / r2 / 1e9,
                   hip_Gkmers_per_s=hst["total"] / h2 / 1e9, stats_equal=hst == rst, ref_stats=rst)
        rep = " ".join(verbose)
        m = re.search(r"reader: admit ([0-9.]+) s, read ([0-9.]+) s \(summed\), wall ([0-9.]+) s \| workers \(summed over threads\): wait for a bin ([0-9.]+) s, engine ([0-9.]+) s, "
                      r"wait for the turn to push ([0-9.]+) s, push ([0-9.]+) s, wall ([0-9.]+) s", rep)
        if m:
            out["worker_report"] = dict(zip(("reader_admit_s", "reader_read_s_summed", "reader_wall_s", "workers_wait_for_a_bin_s_summed", "engine_s_summed",
                                             "wait_to_push_s_summed", "push_s_summed", "workers_wall_s_summed"), (float(x) for x in m.groups())))
        tl = [ln for ln in verbose if ln.startswith("[kmc_hip timeline]")]
        if tl:
            out["timeline"] = tl[0][:900]
        hb = [ln for ln in verbose if ln.startswith("[kmc_hip host boundary]")]
        if hb:  # kmc_hip_host_boundary_times: where the workers' engine seconds went (device buffers, staging copies, enqueue, wait for the kernels, D2H)
            out["host_boundary_times"] = hb[0][24:400]
        # the same bins, device resident: what is left when reader, host link and completer are taken away
        if time.time() - t_leg < budget_s * 0.6:
            dump = os.path.join(td, "dump")
            os.makedirs(dump)
            _run_kmc(hip, flags + ["-sr16"], fq, td, "ldump", dict(env, KMC_HIP_BIN_DUMP_DIR=dump), timeout=budget_s)
            os.remove(fq)
            sb = DumpedBins(dump)
            pl = kmc_lut_prefix_len(k, reads, 512)
            p = capi.make_params(k, lut_prefix_len=pl)
            w = build_workload(ctx, None, k, p, 0, 1, sb=sb)
            run_step(ctx, w, 0)
            t0 = time.perf_counter()
            for _ in range(2):
                run_step(ctx, w, 0)
            dt = (time.perf_counter() - t0) / 2
            res = read_results(ctx, w)
            tl_ = res[:, :4].sum(axis=0, dtype=np.uint64)
            got = {"unique": int(tl_[0]), "below_min": int(tl_[1]), "above_max": int(tl_[2]), "total": int(tl_[3]), "unique_counted": int(tl_[0]) - int(tl_[1]) - int(tl_[2])}
            nr = np.sort(np.array([x[2] for x in w.bins], dtype=np.int64))
            out["device_resident_on_reference_bins"] = {
                "value": w.total_kmers_all / dt / 1e9, "ms_per_step": dt * 1e3, "bins": len(w.bins), "tallies_equal_reference_statistics": got == rst,
                "bin_kmers_min_median_max": [int(nr[0]), int(nr[nr.size // 2]), int(nr[-1])], "bin_kmers_p90_over_median": float(nr[int(nr.size * 0.9)] / max(nr[nr.size // 2], 1)),
                "groups_by_path": ctx.path_counters()}
            w.free()
    out["leg_s"] = time.time() - t_leg
    return out


def secondary_leg(name: str, k: int, extra, env=None):
    cfg = CONFIGS[name]
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", name, "--k", str(k), "--reads", str(cfg["reads"]), "--genome", str(cfg["genome"]),
           "--bins", str(cfg["bins"]), "--steps", "3", "--warmup", "1", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env) if env else None)
    for ln in reversed(r.stdout.splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    return {"error": (r.stdout + r.stderr)[-500:]}


def stage1_leg(k: int):
    """NOT part of `value`: the stage-1 groundwork timed by tools/s1_bench.py in its own process — reads already in HBM as
    codes -> signature bins in HBM, the hand-over kmc_hip_process_bins_device takes. Informative only; never loses the stage-2 line."""
    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "s1_bench.py"), "--k", str(k)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
        for ln in reversed(r.stdout.splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": (r.stdout + r.stderr)[-500:]}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


# ---------------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--reads", type=int, default=CONFIGS["configs[2]"]["reads"], help="150 bp reads of the whole job")
    ap.add_argument("--genome", type=int, default=CONFIGS["configs[2]"]["genome"])
    ap.add_argument("--bins", type=int, default=CONFIGS["configs[2]"]["bins"])
    ap.add_argument("--lut-prefix", type=int, default=-1, help="-1 = KMC's own choice for this input (kmc.h:1434-1469)")
    ap.add_argument("--streams", type=int, default=0, help="stream slots kmc_hip_process_bins_device fans out over (0 = library default)")
    ap.add_argument("--leg", default="", help="internal: run as a secondary leg with this workload name")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the reference/e2e legs")
    ap.add_argument("--no-host-boundary", action="store_true")
    ap.add_argument("--host-group", type=int, default=4, help="bins per call of the host-boundary leg (1: kmc_hip_process_bin_submit, one bin per call)")
    ap.add_argument("--host-threads", type=int, default=0, help="host threads of the host-boundary leg (0: 4 with several bins per call, 8 with one)")
    ap.add_argument("--host-probe", action="store_true", help="diagnostics: the host-boundary leg for every sort selection and call style, errors reported per leg")
    ap.add_argument("--no-host-single", action="store_true", help="skip the one-bin-per-call comparison of the host-boundary leg")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--logical", type=int, default=0, help="after the timed region: the same bins over this many LOGICAL devices of the one GPU, one host thread each (0/1 = skip; 8 = the control flow of --gpus 8 inside one process)")
    ap.add_argument("--no-full-wide", action="store_true", help="skip the full-size k = 55 / k = 127 legs (configs[4] on one GPU, ~80 s each)")
    ap.add_argument("--no-repeat-baselines", action="store_true", help="skip cpu_baseline_skew / cpu_baseline_spectrum (the reference on the skew legs' reads, ~40 s each)")
    ap.add_argument("--no-two-streams", action="store_true", help="skip the value_two_streams leg (profiling runs: keeps overlapped launches out of the kernel statistics)")
    ap.add_argument("--no-digest", action="store_true")
    ap.add_argument("--no-oracle-check", action="store_true", help="skip the byte-for-byte comparison of three of the timed run's bins with the oracle")
    ap.add_argument("--oracle-bins", type=int, default=0, help="bins of the timed run compared byte for byte with the oracle (0: 16 on the full workload, 3 on a quarter leg)")
    ap.add_argument("--cache", default="", help="directory for the generated bin set (tuning sessions: generate once, reuse)")
    ap.add_argument("--e2e-gbp", type=float, default=8.0, help="size of the e2e_large leg's FASTQ in Gbp (0 = skip; 30 = the full configs[2] shape, ~4 minutes more)")
    ap.add_argument("--also-two-streams", action="store_true", help="secondary legs: time the step with two groups in flight as well (value_two_streams)")
    ap.add_argument("--dry-launch", action="store_true", help="launcher check (runs without a GPU): start the ranks --gpus asks for, rendezvous over gloo, print what they see")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it starts its own N ranks (one process per GPU, rendezvous on 127.0.0.1): the same command
    # line re-executed under torch.distributed.run. Under a launcher (the driver's `python -m torch.distributed.run ... bench.py --gpus N`) WORLD_SIZE
    # is already there — and must agree with --gpus: a line that says n_gpus = N after timing fewer ranks would be a wrong number.
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the launcher's ranks and --gpus must agree")
    import torch

    if args.dry_launch:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = torch.tensor([1], dtype=torch.int64)
        dist.all_reduce(seen)
        if rank == 0:
            print(json.dumps({"dry_launch": True, "gpus_requested": args.gpus, "world_size_env": world, "ranks_seen_by_all_reduce": int(seen.item()),
                              "n_gpus": dist.get_world_size()}))
        dist.destroy_process_group()
        return

    # Rehearsal (tests/test_sharding_cpu.py): $KMC_BENCH_REHEARSAL=1 AND a test build of the library (tests/hipemu: kmc_hip_backend_kind() != 0) run this
    # very body — sharded generation, run_step, the tally all-reduce, the digest gather — over gloo on the CPU, so that the N-rank path has executed before it
    # meets N GPUs. The line it prints carries "rehearsal": true and no value; with the GPU library the switch is refused.
    rehearsal = os.environ.get("KMC_BENCH_REHEARSAL") == "1"
    if rehearsal and capi.backend_kind() == 0:
        raise SystemExit("KMC_BENCH_REHEARSAL is for test builds of the library only (kmc_hip_backend_kind() == 0 here: the GPU library)")
    if not rehearsal and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU path")
    dist = None
    if rehearsal:
        dev_index = 0
        dev = torch.device("cpu")
        if world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        # one rank per GPU; if the launcher already narrowed the visible devices to one per process, that one is ordinal 0
        dev_index = local_rank if torch.cuda.device_count() > local_rank else 0
        torch.cuda.set_device(dev_index)
        if world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        dev = torch.device("cuda", dev_index)

    def device_sync():
        if not rehearsal:
            torch.cuda.synchronize()

    is_main = not args.leg
    name = args.leg or next((n for n, c in CONFIGS.items() if (c["reads"], c["genome"], c["bins"]) == (args.reads, args.genome, args.bins)), "custom")
    k = args.k
    if not rehearsal:
        capi.require_gpu_backend()
    ctx = capi.Context((dev_index,))
    pl = args.lut_prefix if args.lut_prefix >= 0 else kmc_lut_prefix_len(k, args.reads, args.bins)
    p = capi.make_params(k, lut_prefix_len=pl)
    want_host = is_main and world == 1 and not args.no_host_boundary
    w = build_workload(ctx, args, k, p, rank, world, keep_host=want_host)

    for _ in range(args.warmup):
        run_step(ctx, w, args.streams)
    ctx.scatter_totals(reset=True)
    ctx.local_sort_totals(reset=True)
    if dist:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(ctx, w, args.streams)
    res = read_results(ctx, w)
    t_busy = time.perf_counter() - t0  # this rank's own bins done and their results read (before it waits for anybody)
    own_tallies = res[:, :4].sum(axis=0, dtype=np.uint64) if w.n_own else np.zeros(4, dtype=np.uint64)
    tallies = sharding.allreduce_tallies(own_tallies, device=dev)  # the one RCCL collective of the path (32 bytes)
    device_sync()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    n_launch, sc_ms, sc_recs = ctx.scatter_totals(reset=True)
    ls = ctx.local_sort_totals(reset=True)
    dt_own = dt
    per_rank = None
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # what every rank did, for the line: its own clock over the timed region (the barrier at the end makes them nearly equal — the busy time is the useful
        # one: wall until its own bins were done), its bins and k-mers. LPT imbalance = the heaviest rank's k-mers / the mean.
        mine = torch.zeros(world, 4, dtype=torch.float64, device=dev)
        mine[rank] = torch.tensor([dt_own, t_busy, float(w.own_kmers), float(w.n_own)], dtype=torch.float64)
        dist.all_reduce(mine)
        per_rank = [{"rank": r_, "ms_per_step": float(mine[r_, 0]) / args.steps * 1e3, "busy_ms_per_step": float(mine[r_, 1]) / args.steps * 1e3,
                     "kmers": int(mine[r_, 2]), "bins": int(mine[r_, 3])} for r_ in range(world)]
        sc = torch.tensor([float(n_launch), sc_ms, float(sc_recs)], dtype=torch.float64, device=dev)
        dist.all_reduce(sc)
        n_launch, sc_ms, sc_recs = int(sc[0].item()), float(sc[1].item()), int(sc[2].item())

    # ---- self-checks on the timed run's own output (every rank)
    rec_bytes = w.rec_bytes
    ok = True
    for i in range(w.n_own):
        u, bmin, amax, tot, ob = (int(x) for x in res[i, :5])
        ok &= tot == w.bins[i][2] and ob == (u - bmin - amax) * rec_bytes
    ok &= int(tallies[3]) == w.total_kmers_all
    digest = None
    if not args.no_digest:
        dg = output_digest(ctx, w, res)
        if dist:
            parts = [None] * world
            dist.all_gather_object(parts, dg)
            dg = sum(parts) & ((1 << 64) - 1)
        digest = "%016x" % dg
    oracle_bins = None
    if rank == 0 and not args.no_oracle_check and args.leg in ("", "quarter", "configs[2]"):
        try:
            oracle_bins = oracle_check(ctx, w, res, args.oracle_bins or (16 if args.leg in ("", "configs[2]") else 3))
        except Exception as e:  # noqa: BLE001 — the checker must not take the measurement down with it
            oracle_bins = [{"error": repr(e)}]
    if dist:
        okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    timings = ctx.last_timings() if w.n_own else {}
    out = None
    if rank == 0:
        W = 8 * ((k + 31) // 32)
        P = (2 * k + 7) // 8
        # records of three words and more: the HBM passes move one word per record (the key's top four bytes above the record's number), k_bucket_rank gathers the
        # records by number (library counter [6]); the scatter kernel is then k_onesweep<1> on 8-byte pairs whatever k is
        pc = ctx.path_counters()
        indirect = pc.get("indirect", 0) > 0 and pc["indirect"] == pc["rank_count"]
        SW = 8 if indirect else W   # bytes per record of what the scatter passes move
        kern = "k_onesweep<%d>" % (1 if indirect else (k + 31) // 32)
        avg_ms = sc_ms / max(n_launch, 1)
        rpl = sc_recs / max(n_launch, 1)
        achieved = (2 * SW * sc_recs) / (sc_ms * 1e-3) / 1e9 if n_launch else 0.0
        value = w.total_kmers_all * args.steps / dt / 1e9
        # which sort ran: LSD passes over every key byte (P of them), or — hybrid — over the top bytes only + the LDS finisher. The sampled groups carry
        # both kinds of event pairs, so passes per record = scatter records / LDS-sorted records.
        hyb = ls["launches"] > 0 and ls["records"] > 0
        hbm_passes = (sc_recs / ls["records"]) if hyb else float(P)
        # which LDS finisher ran (library's own counters): k_bucket_rank fused (tiles ranked and counted in LDS: one read), k_bucket_rank in place + k_compact
        # (one more read + write)
        rank_fused = hyb and pc["rank_count"] > 0 and pc["rank_compact"] == 0 and pc["bucket_count"] == 0
        by_rank = hyb and not rank_fused and pc["rank_compact"] > 0
        moved = W * (1 + 2 * hbm_passes + (2 if by_rank else 0) + 1) + 1.2  # expand write + passes (read + write) [+ rank in place] + one read by the finisher / k_compact + the bin image
        if indirect:
            moved = W + 8 + 2 * 8 * hbm_passes + 8 + W + 1.2  # expand writes record + pair, the passes move pairs, the finisher reads the pair and gathers the record
        moved_pmc = pmc_moved_bytes_per_kmer(kern, rpl, w.total_kmers_all) if world == 1 else None
        desc = (CONFIGS[name]["desc"] % k) if name in CONFIGS else f"custom: k={k}, {args.reads} reads of a {args.genome} bp genome, {args.bins} bins"
        out = {
            "metric": "stage-2 Gk-mers/s, k=%d (bin sort & count: parse + expand + 8-bit LSD radix sort + compaction over all signature bins)" % k,
            "value": value, "unit": "Gk-mers/s", "n_gpus": dist.get_world_size() if dist else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": desc + (" on 1 MI355X" if world == 1 else f", sharded over {world} MI355X by LPT (per-GPU bin queues, RCCL tally reduce)"),
                       "kmers": w.total_kmers_all, "superkmers": w.total_super_all, "bin_image_bytes": w.total_bytes_all, "bins": w.n_bins_all,
                       "bins_rank0": w.n_own, "kmers_rank0": w.own_kmers, "record_bytes": W, "radix_passes": P, "cutoff_min": 2, "counter_max": 255,
                       "lut_prefix_len": pl, "streams": args.streams or "auto (1 stream when bins average >= 64 MB of records, else 8)",
                       "bins_per_sort": min(1 << (8 * P - 2 * k), 16, int(os.environ.get("KMC_HIP_GROUP", "16"))),
                       "parallelism": "bins sharded over ranks (LPT), 1 process/GPU, tallies all-reduced (RCCL)" if world > 1 else "1 GPU"},
            "unique_kmers_per_s": float(tallies[0]) * args.steps / dt,
            "tallies": {"n_unique": int(tallies[0]), "n_cutoff_min": int(tallies[1]), "n_cutoff_max": int(tallies[2]), "n_total": int(tallies[3])},
            "self_check": {"per_bin_total_and_out_bytes_consistent": bool(ok), "output_digest": digest,
                           "oracle_bins_equal": (all(v.get("equal") for v in oracle_bins) if oracle_bins else None), "oracle_bins": oracle_bins},
            "sort_path": {"what": ("hybrid: 8-bit LSD passes through HBM over the top key bytes only, then every bucket-aligned tile put in order inside LDS (k_bucket_rank: a record's "
                                   "place = the records of its bucket below it, counted pairwise) and counted there — run lengths, cutoffs, (suffix, counter) records, LUT, "
                                   "tallies: the sorted tile never goes back to HBM" +
                                   ("; INDIRECT: the passes move (key top, record number) pairs of 8 bytes, the records stay where k_expand wrote them and are gathered by number" if indirect else "")
                                   if rank_fused else
                                   "hybrid: 8-bit LSD passes through HBM over the top key bytes only, then every bucket-aligned tile put in order inside LDS (k_bucket_rank: a record's "
                                   "place = the records of its bucket below it, counted pairwise), then k_compact" if by_rank else
                                   "hybrid: 8-bit LSD passes through HBM over the top key bytes only, the rest inside LDS on bucket-aligned tiles" if hyb
                                   else "8-bit LSD passes through HBM over every key byte, then k_compact"),
                          "groups_by_path": pc, "hbm_passes_per_record": hbm_passes, "hbm_bytes_per_kmer_moved_by_design": moved, "moved_GBs": moved * value, "moved_frac_of_hbm_peak": moved * value / HBM_PEAK_GBS,
                          "hbm_bytes_per_kmer_moved_pmc": moved_pmc, "moved_frac_of_hbm_peak_pmc": (moved_pmc * value / HBM_PEAK_GBS) if moved_pmc else None,
                          "pmc_source": os.path.relpath(PMC_PROFILE, ROOT) + " (a rocprofv3 --pmc run of this workload on the committed tree, not a counter of THIS run: counters cannot be read from inside it)",
                          "note": "SURVEY 8d: an implementation with fewer passes moves fewer real bytes — stage2_algorithmic_* below is the NORMATIVE 8-bit-LSD figure W(2P+3) "
                                  "(what the reference formulation would have to move for this throughput: it can exceed the HBM peak when passes are skipped), "
                                  "moved_* is what this path is designed to move (PMC-checked per kernel in profiles/r05)"},
            "stage2_algorithmic_bytes_per_kmer": W * (2 * P + 3),
            "stage2_algorithmic_GBs": W * (2 * P + 3) * value,
            "stage2_frac_of_hbm_peak": W * (2 * P + 3) * value / HBM_PEAK_GBS,
            "roofline": {"bound": "hbm", "kernel": kern, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(kern, rpl),
                         "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE of a rocprofv3 run of this workload, profiles/r06/pmc_hbm_traffic.json: a committed constant, not a counter of this run)",
                         "algorithmic_bytes_per_launch": 2 * SW * rpl, "launches_in_timed_region": n_launch, "avg_launch_ms": avg_ms,
                         "records_per_launch": rpl, "algorithmic_bytes_per_record_per_launch": 2 * SW,
                         "note": "consecutive bins of a stream share one sort (bins_per_sort: the bin's number inside the group rides in the spare bits of the "
                                 "top radix digit), so a launch covers that many bins; every 8th group of a stream carries the event pairs (an event costs "
                                 "stream time); big bins run on one stream, so launches do not overlap and the event durations are the kernel's own"},
            "local_sort": {"kernel": ("k_bucket_bounds + k_bucket_rank<%d, fused>" % ((k + 31) // 32) if rank_fused else "k_bucket_bounds + k_bucket_rank<1, in place>" if by_rank
                                      else "k_bucket_bounds + LDS finisher<%d>" % ((k + 31) // 32)) +
                                     " (the key bytes below the HBM passes, resolved inside LDS on bucket-aligned tiles)",
                           "launches_timed": ls["launches"], "avg_launch_ms": ls["ms"] / max(ls["launches"], 1), "records_per_launch": ls["records"] / max(ls["launches"], 1),
                           "GBs_read_plus_written": (2 * W * ls["records"]) / (ls["ms"] * 1e-3) / 1e9 if ls["launches"] else 0.0,
                           "hybrid_groups": ls["hybrid_groups"], "redo_groups": ls["redo_groups"]},
            "phases_ms_last_bin_slot0": timings,
            "phases_note": "event intervals of the last timed GROUP of bins on stream slot 0 (bins_per_sort bins; one launch each of parse+index, expand, "
                           "the scatter passes, compaction + fold + gather)",
            "setup_s": w.setup_s,
        }
        if indirect and ls["launches"]:
            # records of two words and more: the scatter passes move 8-byte pairs and are the smaller part of the step; the kernel that dominates is the finisher
            # (profiles/r05/quarter_k55_kernel_stats.csv: k_bucket_rank<2> 46 %, k_onesweep<1> 35 %). Its algorithmic bytes per record: the pair it reads (8) + the
            # record it gathers by number (W); the (suffix, counter) records it writes are counted from the run's own out_bytes.
            fin_bytes = (8 + W) * ls["records"] + float(res[:, 4].sum()) * ls["records"] / max(w.own_kmers, 1)
            fin_ach = fin_bytes / (ls["ms"] * 1e-3) / 1e9
            out["roofline_scatter"] = out["roofline"]
            out["roofline"] = {"bound": "hbm", "kernel": "k_bucket_rank<%d, true>" % (W // 8), "achieved": fin_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": fin_ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": fin_bytes / ls["launches"],
                               "launches_in_timed_region": ls["launches"], "avg_launch_ms": ls["ms"] / ls["launches"], "records_per_launch": ls["records"] / ls["launches"],
                               "algorithmic_bytes_per_record_per_launch": 8 + W,
                               "note": "the dominant kernel of this record width (k >= 33: indirect sort): HIP events around k_bucket_bounds + k_bucket_rank of every 8th "
                                       "group; algorithmic bytes = (8-byte pair + W-byte record gathered by number) per record + the output records written. The gather "
                                       "pulls a 64-byte request per record (profiles/r05/pmc_hbm_traffic_k55.json): traffic is ~3x this at k = 55; the scatter passes' own "
                                       "roofline is in roofline_scatter"}
        if per_rank:
            out["per_rank"] = per_rank
            km = [pr["kmers"] for pr in per_rank]
            out["lpt_imbalance"] = max(km) / (sum(km) / len(km)) if sum(km) else None
            if rehearsal:
                for pr in per_rank:
                    pr["busy_ms_per_step"] = None
    if rank == 0 and world == 1 and args.leg and args.also_two_streams:
        # a secondary leg on repeat-rich input: what the latency-bound k_giant_tiles costs when another group's kernels can run beside it (the drop-in's normal
        # mode: several slots in flight). The leg's `value` stays the one-stream schedule, comparable with the rounds before.
        try:
            run_step(ctx, w, 2)
            t1 = time.perf_counter()
            for _ in range(2):
                run_step(ctx, w, 2)
            out["value_two_streams"] = w.total_kmers_all * 2 / (time.perf_counter() - t1) / 1e9
        except Exception as e:  # noqa: BLE001
            out["value_two_streams"] = repr(e)
    # ---- after the timed region: overlapped streams, host boundary, secondary workloads, the reference (rank 0 of a 1-GPU run only)
    if rank == 0 and world == 1 and is_main:
        # The timed region runs big bins back to back on ONE stream, so that a k_onesweep launch has the GPU to itself and its
        # duration means something (HIP events and rocprofv3 agree). Two bins in flight fill each other's tails and gaps:
        try:
            if args.no_two_streams:
                raise RuntimeError("skipped (--no-two-streams)")
            run_step(ctx, w, 2)
            t1 = time.perf_counter()
            for _ in range(2):
                run_step(ctx, w, 2)
            out["value_two_streams"] = w.total_kmers_all * 2 / (time.perf_counter() - t1) / 1e9
            ctx.scatter_totals(reset=True)
        except Exception as e:  # noqa: BLE001
            out["value_two_streams"] = repr(e)
        if args.logical > 1 and not rehearsal:
            try:
                out["logical_devices"] = logical_devices_leg(dev_index, w, args.logical, tallies)
            except Exception as e:  # noqa: BLE001
                out["logical_devices"] = {"error": repr(e)[-400:]}
        if want_host and args.host_probe:  # diagnostics: every combination of sort selection and call style, each on its own
            probe = []
            for mode in (1, 1, 1, 0):
                ctx.set_hybrid(mode)
                for grp in (1, 4):
                    t0 = time.perf_counter()
                    try:
                        secs, ht, t_pin, nth = host_boundary_pass(ctx, w, n_threads=args.host_threads or (4 if grp > 1 else 8), group=grp)
                        probe.append({"hybrid_mode": mode, "bins_per_call": grp, "value": w.total_kmers_all / secs / 1e9, "wall_s": time.perf_counter() - t0,
                                      "tallies_equal": [int(x) for x in ht[:4]] == [int(x) for x in tallies]})
                    except Exception as e:  # noqa: BLE001
                        probe.append({"hybrid_mode": mode, "bins_per_call": grp, "error": repr(e)[:600], "wall_s": time.perf_counter() - t0})
                        try:
                            ctx.synchronize()
                        except Exception as e2:  # noqa: BLE001
                            probe[-1]["synchronize"] = repr(e2)[:200]
            ctx.set_hybrid(1)
            out["host_probe"] = probe
        elif want_host:
            try:
                legs_hb = {}
                failed = []
                for grp in ([args.host_group] if args.no_host_single else sorted({1, args.host_group})):
                    entry = "kmc_hip_process_bins_submit/_wait" if grp > 1 else "kmc_hip_process_bin_submit/_wait"
                    try:
                        secs, ht, t_pin, nth = host_boundary_pass(ctx, w, n_threads=args.host_threads or (4 if grp > 1 else 8), group=grp)
                    except Exception as e:  # noqa: BLE001 — a leg that fails is reported, the other one still counts
                        failed.append({"bins_per_call": grp, "entry": entry, "error": repr(e)[:800]})
                        try:
                            ctx.synchronize()
                        except Exception:  # noqa: BLE001
                            pass
                        continue
                    legs_hb[grp] = {"value": w.total_kmers_all / secs / 1e9, "seconds": secs, "bins_per_call": grp, "host_threads": nth, "bytes_out": int(ht[4]),
                                    "pin_and_stage_s": t_pin, "tallies_equal_device_resident": [int(x) for x in ht[:4]] == [int(x) for x in tallies], "entry": entry}
                if not legs_hb:
                    raise RuntimeError("; ".join(f["error"] for f in failed))
                best = max(legs_hb.values(), key=lambda v: v["value"])
                out["value_host_boundary"] = best["value"]
                out["host_boundary"] = {"what": "the same bins from pinned host memory through the host boundary, host threads x 2 stream slots (H2D + kernels + D2H of every "
                                                "bin; PCIe inclusive), best of 2 passes; value_host_boundary = the better of: one bin per call, %d consecutive bins per "
                                                "call (sorted together on the device)" % args.host_group,
                                        "bytes_in": w.total_bytes_all, "best": best["entry"], "seconds": best["seconds"], "bins_per_call": best["bins_per_call"],
                                        "tallies_equal_device_resident": all(v["tallies_equal_device_resident"] for v in legs_hb.values()),
                                        "legs": [legs_hb[g] for g in sorted(legs_hb)], "failed_legs": failed}
            except Exception as e:  # noqa: BLE001
                out["host_boundary"] = {"error": repr(e)}
        w.free()
        w.host_imgs = []
        if w.sb:
            w.sb.close()
        if not args.no_secondary:
            sec = {}
            s1 = secondary_leg("configs[1]", k, [])
            sec["single_bin"] = {kk: s1.get(kk) for kk in ("value", "ms_per_step", "config", "roofline", "tallies", "phases_ms_last_bin_slot0", "stage2_frac_of_hbm_peak", "error") if kk in s1}
            s2 = secondary_leg("2gbp-512bins", k, [])
            sec["bins512_2gbp"] = {kk: s2.get(kk) for kk in ("value", "ms_per_step", "config", "roofline", "tallies", "stage2_frac_of_hbm_peak", "error") if kk in s2}
            sec["stage1_groundwork"] = stage1_leg(k)
            if k == 27:  # configs[4]'s record widths on a quarter of the reads: the hybrid sort (docs/history/DESIGN_rounds_1_to_5.md §4b); full size: bench.py --k 55 / --k 127
                for kk in (55, 127):
                    sk = secondary_leg("quarter", kk, ["--no-digest"])
                    sec["k%d_quarter" % kk] = {x: sk.get(x) for x in ("value", "ms_per_step", "config", "roofline", "sort_path", "local_sort", "tallies", "self_check", "error") if x in sk}
                # skew: the same quarter with a repeat family planted in the genome (a 10 kbp unit x 2000 copies, 1 % diverged: its k-mers occur tens of thousands
                # of times — far beyond a tile of the LDS sort): what do k_giant_tiles and, beyond it, the redo through LSD passes cost on repeat-rich input?
                sk = secondary_leg("quarter", 27, ["--no-digest", "--also-two-streams"], env={"KMC_SYNTH_REPEATS": SKEW_REPEATS})
                sec["skew_quarter"] = dict({x: sk.get(x) for x in ("value", "value_two_streams", "ms_per_step", "sort_path", "local_sort", "tallies", "self_check", "error") if x in sk},
                                           what="quarter workload, k=27, $KMC_SYNTH_REPEATS=%s (unit:copies:per-mille divergence) planted in the genome; sort_path.groups_by_path has "
                                                "the tiles / records k_giant_tiles took, local_sort.redo_groups the groups that went back through LSD passes" % SKEW_REPEATS)
                sk = secondary_leg("quarter", 27, ["--no-digest", "--also-two-streams"], env={"KMC_SYNTH_REPEATS": SKEW_SPECTRUM})
                sec["skew_spectrum_quarter"] = dict({x: sk.get(x) for x in ("value", "value_two_streams", "ms_per_step", "sort_path", "local_sort", "tallies", "self_check", "error") if x in sk},
                                                    what="quarter workload, k=27, $KMC_SYNTH_REPEATS=%s: a spectrum of repeat families (unit:copies:per-mille divergence, H = a "
                                                         "homopolymer run) planted in the genome" % SKEW_SPECTRUM)
                # configs[4]'s record widths at FULL size (200 M reads, 512 bins, one GPU): the legs the quarter numbers above stand in for; roofline = the kernel that
                # dominates THEM (the finisher), 16 bins each against the oracle
                if not args.no_full_wide:
                    for kk in (55, 127):
                        sk = secondary_leg("configs[2]", kk, ["--no-digest"])
                        sec["k%d_full" % kk] = {x: sk.get(x) for x in ("value", "ms_per_step", "config", "roofline", "roofline_scatter", "sort_path", "local_sort", "tallies", "self_check", "error") if x in sk}
            out["secondary"] = sec
        if not args.no_cpu_baseline:
            try:
                legs = reference_legs(k, CONFIGS["2gbp-512bins"]["reads"], CONFIGS["2gbp-512bins"]["genome"])
                out.update(legs)
                s2t = out.get("secondary", {}).get("bins512_2gbp", {}).get("tallies")
                st = legs["cpu_baseline"]["stats"]
                if s2t:
                    out["self_check"]["gpu_tallies_equal_reference_on_2gbp_sample"] = (
                        s2t["n_unique"] == st["unique"] and s2t["n_cutoff_min"] == st["below_min"] and s2t["n_cutoff_max"] == st["above_max"]
                        and s2t["n_total"] == st["total"])
            except Exception as e:  # the baseline is informative; never lose the GPU number over it
                out["cpu_baseline"] = {"value": None, "unit": "Gk-mers/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
            if k == 27 and not args.no_secondary and not args.no_repeat_baselines:
                m = re.search(r"-t(\d+)", (out.get("cpu_baseline") or {}).get("sample") or "")
                out.update(reference_repeat_legs(k, int(m.group(1)) if m else max(2, sharding.effective_cpus()), out.get("secondary") or {}))
                cb = out.get("cpu_baseline") or {}
                if isinstance(cb.get("value"), float):
                    out["gpu_over_cpu"] = {"uniform": out["value"] / cb["value"], "skew": (out.get("cpu_baseline_skew") or {}).get("gpu_over_cpu"),
                                           "spectrum": (out.get("cpu_baseline_spectrum") or {}).get("gpu_over_cpu"),
                                           "note": "device-resident GPU value / the reference's '2nd stage' rate on this host's cores: uniform = configs[2] over the 2 Gbp sample's "
                                                   "baseline; skew / spectrum = the quarter legs over the reference on a FASTQ of the same reads"}
            if args.e2e_gbp > 0:
                try:
                    out["e2e_large"] = e2e_large_leg(ctx, k, args.e2e_gbp)
                except Exception as e:  # noqa: BLE001
                    out["e2e_large"] = {"error": repr(e)[-600:]}
    if rank == 0:
        if rehearsal:  # control flow only: the same record with every timing taken out, so that nothing in it can be taken for a measurement
            out.update(rehearsal=True, backend_kind=capi.backend_kind(), value=None, ms_per_step=None, unique_kmers_per_s=None, stage2_algorithmic_GBs=None,
                       stage2_frac_of_hbm_peak=None, phases_ms_last_bin_slot0=None, setup_s=None)
            out["roofline"].update(achieved=None, frac=None, avg_launch_ms=None)
            out["sort_path"].update(moved_GBs=None, moved_frac_of_hbm_peak=None)
            out["local_sort"].update(avg_launch_ms=None, GBs_read_plus_written=None)
            for pr in out.get("per_rank") or []:
                pr["ms_per_step"] = None
        if args.leg:
            print(json.dumps(out))  # a secondary leg talks to its parent process: the whole record
        else:
            emit(out)
    if dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()

"""Multi-GPU sharding of stage 2: signature bins are independent sort/count problems, so they are
partitioned over ranks (one process per GPU) with NO data-path collective; only the four end-of-run tallies
(n_unique, n_cutoff_min, n_cutoff_max, n_total — the sums the completer keeps, kb_completer.cpp:206-209) are
combined, with a single all-reduce (RCCL over xGMI when the backend is "nccl"; gloo on CPU in tests)."""
from __future__ import annotations

import heapq

import numpy as np


def effective_cpus() -> int:
    """CPUs this process may really use: the smaller of the visible CPUs, the affinity mask and the cgroup CPU quota
    (the GPU boxes show 256 hardware threads but grant a 16-CPU quota: 256 busy threads then run slower than 32)."""
    import math
    import os

    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(p))))
    except Exception:
        pass
    return n


def lpt_assign(sizes, world: int):
    """Longest-processing-time assignment of bins to ranks. `sizes[i]` = cost of bin i (records to sort, the
    quantity CBinDesc::get_sorted_req_sizes orders bins by, queues.h:499-558). Returns world lists of bin ids, each
    in descending size order (the order the reference hands bins to sorters). Deterministic."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    out = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(i)
        heapq.heappush(heap, (load + int(sizes[i]), r))
    return out


def allreduce_tallies(stats, device=None) -> np.ndarray:
    """Sum a length-4 tally vector over all ranks of the default torch.distributed group (identity when the
    process group is not initialised). int64 on the wire: the tallies are < 2^63."""
    import torch
    import torch.distributed as dist

    a = np.asarray(stats, dtype=np.uint64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return a.copy()
    t = torch.from_numpy(a.astype(np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


class ShardedBins:
    """This rank's bins of a synthetic bin set that all ranks generated together (see generate_sharded_bins)."""

    def __init__(self):
        self.own = []            # bin ids of this rank, ascending
        self.pieces = {}         # bin id -> [(image uint8 array, pack_bytes uint64 array), ...] in read order (one piece per source rank)
        self.size = self.n_rec = self.n_packs = self.n_super = None  # per-bin totals over ALL ranks (int64 arrays of length n_bins)
        self.timings = {}
        self._keep = []
        self._cleanup = None

    def image(self, b):
        ps = self.pieces[b]
        return ps[0][0] if len(ps) == 1 else np.concatenate([x[0] for x in ps])

    def packs(self, b):
        ps = self.pieces[b]
        return np.asarray(ps[0][1], dtype=np.uint64) if len(ps) == 1 else np.concatenate([np.asarray(x[1], dtype=np.uint64) for x in ps])

    def close(self):
        self.pieces = {}
        for k in self._keep:
            if hasattr(k, "close"):
                k.close()
        self._keep = []
        if self._cleanup:
            self._cleanup()
            self._cleanup = None


def _load_cached(path_prefix, n_bins):
    import os

    if not (os.path.exists(path_prefix + ".npz") and os.path.exists(path_prefix + ".bin")):
        return None
    m = dict(np.load(path_prefix + ".npz"))
    if len(m["sizes"]) != n_bins:
        return None
    sb = ShardedBins()
    mm = np.memmap(path_prefix + ".bin", dtype=np.uint8, mode="r") if os.path.getsize(path_prefix + ".bin") else np.zeros(0, dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum(m["sizes"])])
    poffs = np.concatenate([[0], np.cumsum(m["npk"])])
    sb.own = list(range(n_bins))
    sb.size, sb.n_rec, sb.n_packs, sb.n_super = m["sizes"], m["nrec"], m["npk"], m["nsup"]
    sb.pieces = {b: [(mm[offs[b]:offs[b + 1]], m["packs"][poffs[b]:poffs[b + 1]])] for b in sb.own}
    sb._keep = [mm]
    return sb


def generate_sharded_bins(seed, genome_len, n_reads, k, n_bins, rank=0, world=1, n_threads=0, scratch_base=None, cache_dir=None) -> ShardedBins:
    """The SAME `n_bins` signature bins whatever `world` is, sharded over ranks (BASELINE configs[3]).

    Every rank generates a chunk-aligned 1/world of the reads into all bins (kmc_amd/csrc/synth_bins.cpp: a bin image is the
    concatenation, in read order, of the pieces of chunk-aligned read ranges), the pieces are exchanged through files in a
    scratch directory (setup, not data path: stage 1 would have written these bins to disk), bins are assigned to ranks by
    LPT on their k-mer counts, and each rank maps the pieces of its own bins. world == 1 needs no torch.distributed."""
    import os
    import shutil
    import tempfile
    import time

    from . import capi

    cache = None
    if cache_dir and world == 1:  # tuning sessions: several bench runs in one gpurun call share the generated bin set
        cache = os.path.join(cache_dir, f"kmcbins_s{seed}_g{genome_len}_r{n_reads}_k{k}_b{n_bins}")
        t = time.time()
        got = _load_cached(cache, n_bins)
        if got is not None:
            got.timings["load_cached"] = time.time() - t
            return got
    sb = ShardedBins()
    cr = capi.synth_chunk_reads()
    n_chunks = (n_reads + cr - 1) // cr
    c0, c1 = n_chunks * rank // world, n_chunks * (rank + 1) // world
    r0, r1 = min(n_reads, c0 * cr), min(n_reads, c1 * cr)
    t = time.time()
    syn = capi.synth_bins(seed=seed, genome_len=genome_len, n_reads=n_reads, k=k, n_bins=n_bins, n_threads=n_threads, read_begin=r0,
                          read_end=r1, copy=False)
    sb.timings["generate"] = time.time() - t
    sizes = np.array([b[0].size for b in syn.bins], dtype=np.int64)
    nrec = np.array([b[1] for b in syn.bins], dtype=np.int64)
    npk = np.array([b[2].size for b in syn.bins], dtype=np.int64)
    nsup = np.array([b[3] for b in syn.bins], dtype=np.int64)
    if world == 1:
        sb.own = list(range(n_bins))
        sb.size, sb.n_rec, sb.n_packs, sb.n_super = sizes, nrec, npk, nsup
        sb.pieces = {b: [(syn.bins[b][0], syn.bins[b][2])] for b in sb.own}
        sb._keep = [syn]
        if cache and shutil.disk_usage(cache_dir).free < int(sizes.sum()) + (1 << 30):
            cache = None  # no room: this run simply does not leave a cache behind
        if cache:
            t = time.time()
            with open(cache + ".bin", "wb") as f:
                for b in range(n_bins):
                    if syn.bins[b][0].size:
                        f.write(memoryview(syn.bins[b][0]))
            np.savez(cache + ".npz", sizes=sizes, nrec=nrec, npk=npk, nsup=nsup,
                     packs=np.concatenate([np.asarray(b[2], dtype=np.uint64) for b in syn.bins]) if npk.sum() else np.zeros(0, dtype=np.uint64))
            sb.timings["save_cache"] = time.time() - t
        return sb

    import torch.distributed as dist

    t = time.time()
    base = [None]
    if rank == 0:
        need = int(n_reads * 150 * 0.4) + (1 << 28)
        cands = [d for d in ([scratch_base] if scratch_base else []) + ["/dev/shm", os.environ.get("TMPDIR") or "/tmp", "/tmp"] if d and os.path.isdir(d)]
        d = next((c for c in cands if shutil.disk_usage(c).free >= need), max(cands, key=lambda c: shutil.disk_usage(c).free))
        base[0] = tempfile.mkdtemp(prefix="kmcbins_", dir=d)
    dist.broadcast_object_list(base, src=0)
    xdir = base[0]
    with open(os.path.join(xdir, f"src{rank}.bin"), "wb") as f:
        for b in range(n_bins):
            if syn.bins[b][0].size:
                f.write(memoryview(syn.bins[b][0]))
    np.savez(os.path.join(xdir, f"src{rank}.npz"), sizes=sizes, nrec=nrec, npk=npk, nsup=nsup,
             packs=np.concatenate([np.asarray(b[2], dtype=np.uint64) for b in syn.bins]) if npk.sum() else np.zeros(0, dtype=np.uint64))
    syn.close()
    dist.barrier()
    metas = [dict(np.load(os.path.join(xdir, f"src{s}.npz"))) for s in range(world)]
    sb.size = sum(m["sizes"] for m in metas)
    sb.n_rec = sum(m["nrec"] for m in metas)
    sb.n_packs = sum(m["npk"] for m in metas)
    sb.n_super = sum(m["nsup"] for m in metas)
    sb.own = sorted(lpt_assign(sb.n_rec, world)[rank])
    mm = []
    for s_ in range(world):
        path = os.path.join(xdir, f"src{s_}.bin")
        mm.append(np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.zeros(0, dtype=np.uint8))
    offs = [np.concatenate([[0], np.cumsum(m["sizes"])]) for m in metas]
    poffs = [np.concatenate([[0], np.cumsum(m["npk"])]) for m in metas]
    sb.pieces = {b: [(mm[s_][offs[s_][b]:offs[s_][b + 1]], metas[s_]["packs"][poffs[s_][b]:poffs[s_][b + 1]]) for s_ in range(world)]
                 for b in sb.own}
    sb.timings["exchange"] = time.time() - t

    def cleanup():
        mm.clear()
        dist.barrier()
        if rank == 0:
            shutil.rmtree(xdir, ignore_errors=True)

    sb._cleanup = cleanup
    return sb

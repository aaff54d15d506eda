"""Multi-GPU sharding of stage 2: signature bins are independent sort/count problems, so they are
partitioned over ranks (one process per GPU) with NO data-path collective; only the four end-of-run tallies
(n_unique, n_cutoff_min, n_cutoff_max, n_total — the sums the completer keeps, kb_completer.cpp:206-209) are
combined, with a single all-reduce (RCCL over xGMI when the backend is "nccl"; gloo on CPU in tests)."""
from __future__ import annotations

import heapq

import numpy as np


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

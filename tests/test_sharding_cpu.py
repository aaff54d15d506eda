"""CPU suite, part 2: the N>1 path (bin sharding + tally all-reduce) with world_size-2 gloo.
Each rank runs the ORACLE on its bins here (tests may use the oracle as the checker); on GPUs the same sharding
drives libkmc_hip (bench.py --gpus N)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

import oracle_py as O
from kmc_amd import capi, sharding


def test_lpt_assign_is_a_balanced_partition():
    rng = np.random.default_rng(3)
    sizes = rng.integers(0, 10**6, size=512).tolist() + [0, 0, 0]
    for world in (1, 2, 4, 8):
        parts = sharding.lpt_assign(sizes, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(sizes)))
        loads = [sum(sizes[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(sizes)
        for p in parts:
            assert [sizes[i] for i in p] == sorted((sizes[i] for i in p), reverse=True)
    assert sharding.lpt_assign(sizes, 4) == sharding.lpt_assign(sizes, 4)


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bins = capi.synth_bins(seed=5, genome_len=50_000, n_reads=3000, k=27, n_bins=16, n_threads=1)
    parts = sharding.lpt_assign([b[1] for b in bins], world)
    p = O.make_params(27)
    mine = np.zeros(4, dtype=np.uint64)
    for i in parts[rank]:
        _, _, st = O.process_bin(p, bins[i][0], bins[i][1])
        mine += st
    tot = sharding.allreduce_tallies(mine)
    q.put((rank, mine.tolist(), tot.tolist()))
    dist.destroy_process_group()


def test_two_rank_tally_allreduce_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    bins = capi.synth_bins(seed=5, genome_len=50_000, n_reads=3000, k=27, n_bins=16, n_threads=1)
    p = O.make_params(27)
    want = np.zeros(4, dtype=np.uint64)
    for img, nrec, _, _ in bins:
        want += O.process_bin(p, img, nrec)[2]
    for rank, mine, tot in res:
        assert tot == want.tolist()
    assert (np.array(res[0][1], dtype=np.uint64) + np.array(res[1][1], dtype=np.uint64)).tolist() == want.tolist()

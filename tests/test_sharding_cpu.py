"""CPU suite, part 2: the N>1 path (bin sharding + tally all-reduce) with world_size-2 gloo.
Each rank runs the ORACLE on its bins here (tests may use the oracle as the checker); on GPUs the same sharding
drives libkmc_hip (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
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


def _shard_worker(rank, world, port, q, scratch):
    import hashlib
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # 70 000 reads = 2 full generator chunks + a partial one; n_bins > world so every rank owns several bins
    sb = sharding.generate_sharded_bins(seed=77, genome_len=120_000, n_reads=70_000, k=27, n_bins=24, rank=rank, world=world, n_threads=2,
                                        scratch_base=scratch)
    own = {}
    for b in sb.own:
        img, pk = sb.image(b), sb.packs(b)
        own[b] = (hashlib.md5(np.ascontiguousarray(img).tobytes()).hexdigest(), hashlib.md5(pk.tobytes()).hexdigest(), int(sb.n_rec[b]), int(sb.size[b]))
    # the oracle over own bins + the one collective of the path
    p = O.make_params(27)
    mine = np.zeros(4, dtype=np.uint64)
    for b in sb.own:
        mine += O.process_bin(p, np.ascontiguousarray(sb.image(b)), int(sb.n_rec[b]))[2]
    tot = sharding.allreduce_tallies(mine)
    sb.close()
    q.put((rank, own, tot.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_generation_reassembles_the_single_process_bin_set(world, tmp_path):
    """bench.py --gpus N: every rank generates 1/N of the reads, pieces are exchanged through a scratch directory, bins go to ranks
    by LPT. The union over ranks must be EXACTLY the bin set one process generates (images, pack lists, k-mer counts), each bin on
    exactly one rank, and the all-reduced tallies must equal the single-process tallies."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import hashlib

    bins = capi.synth_bins(seed=77, genome_len=120_000, n_reads=70_000, k=27, n_bins=24, n_threads=1)
    seen = {}
    for rank, own, tot in res:
        for b, v in own.items():
            assert b not in seen, "a bin landed on two ranks"
            seen[b] = v
    assert sorted(seen) == list(range(24))
    p = O.make_params(27)
    want = np.zeros(4, dtype=np.uint64)
    for b, (img, nrec, pk, _) in enumerate(bins):
        assert seen[b] == (hashlib.md5(img.tobytes()).hexdigest(), hashlib.md5(pk.tobytes()).hexdigest(), nrec, img.size), b
        want += O.process_bin(p, img, nrec)[2]
    for rank, own, tot in res:
        assert tot == want.tolist()
    # LPT balance: no rank holds more than the ideal share + the largest bin
    loads = [sum(v[2] for v in own.values()) for _, own, _ in res]
    assert max(loads) <= sum(loads) / world + max(b[1] for b in bins)
    assert not [f for f in os.listdir(tmp_path) if f.startswith("kmcbins_")], "the exchange directory must be removed"


def test_bench_gpus_n_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (re-exec under torch.distributed.run, rendezvous on 127.0.0.1)
    and must refuse a world size that disagrees with --gpus. --dry-launch stops after the rendezvous (gloo): no GPU needed."""
    import json
    import subprocess
    import sys

    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-800:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["gpus_requested"] == 2 and d["world_size_env"] == 2 and d["ranks_seen_by_all_reduce"] == 2 and d["n_gpus"] == 2
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="3", RANK="0"))
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_bench_multi_rank_body_rehearsed_over_gloo_on_the_emulated_library(tmp_path):
    """The N-rank body of bench.py — sharded generation, LPT, run_step on every rank's own bins, the tally all-reduce, the gather of the output digests — has
    only ever met one real GPU. Here it runs as `python bench.py --gpus 2` (its own launcher, two processes, gloo) over the CPU emulation of the host library
    (KMC_BENCH_REHEARSAL=1 is honoured for test builds only and prints no value): tallies and the order-independent output digest must equal the 1-rank run's,
    and the 2-rank run must have sorted its own share on each rank. The same switch is refused with the GPU library."""
    import json
    import subprocess
    import sys

    import emu

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    lib = emu.build_hostlib("small")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env = dict(base, KMC_HIP_LIB=lib, KMC_BENCH_REHEARSAL="1")
    args = ["--reads", "3000", "--genome", "30000", "--bins", "8", "--steps", "1", "--warmup", "0", "--no-secondary", "--no-cpu-baseline", "--no-host-boundary",
            "--no-two-streams", "--no-oracle-check"]
    out = {}
    for n in (1, 2, 4, 8):
        r = subprocess.run([sys.executable, bench, "--gpus", str(n), *args], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
        assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
        last = r.stdout.rstrip("\n").splitlines()[-1]
        # the line the driver parses: the LAST line of stdout, one JSON object, short enough for the tail the driver keeps (round 4's 22 KB line was cut)
        assert len(last) < 6000, len(last)
        out[n] = json.loads(last)
        assert out[n]["rehearsal"] is True and out[n]["value"] is None and out[n]["ms_per_step"] is None and out[n]["n_gpus"] == n
        for key in ("metric", "unit", "steps", "warmup", "dtype", "config", "roofline", "cpu_baseline", "self_check", "moved_bytes_per_kmer", "scaling"):
            assert key in out[n], key
        assert out[n]["roofline"]["kernel"].startswith("k_onesweep") and out[n]["roofline"]["peak"] == 8000.0
        assert out[n]["self_check"]["per_bin_total_and_out_bytes_consistent"] is True
        if n == 1:
            assert "no scaling curve" in out[n]["multi_gpu"]
        else:  # what every rank did is in the line: its k-mers, its bins, the LPT imbalance
            pr = out[n]["per_rank"]
            assert [x["rank"] for x in pr] == list(range(n)) and sum(x["kmers"] for x in pr) == out[n]["config"]["kmers"]
            assert sum(x["bins"] for x in pr) == out[n]["config"]["bins"] and out[n]["lpt_imbalance"] >= 1.0
            assert out[n]["tallies"] == out[1]["tallies"] and out[n]["tallies"]["n_total"] == out[1]["config"]["kmers"]
            assert out[n]["self_check"]["output_digest"] == out[1]["self_check"]["output_digest"]
            assert 0 < out[n]["config"]["kmers_rank0"] < out[n]["config"]["kmers"] and out[n]["config"]["bins_rank0"] < out[n]["config"]["bins"]
    detail = json.load(open(os.path.join(root, "bench_detail.json")))  # the whole record went to the side file
    assert detail["rehearsal"] is True and "sort_path" in detail and len(json.dumps(detail)) > len(last)
    if os.path.exists(capi._build.LIB_HIP):  # the GPU library refuses the rehearsal switch (dlopen works without a GPU)
        r = subprocess.run([sys.executable, bench, "--gpus", "1", *args], capture_output=True, text=True, timeout=300,
                           env=dict(base, KMC_BENCH_REHEARSAL="1", KMC_HIP_LIB=capi._build.LIB_HIP), cwd=str(tmp_path))
        assert r.returncode != 0 and "test builds of the library only" in (r.stdout + r.stderr)

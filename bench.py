#!/usr/bin/env python3
"""bench.py — KMC stage-2 (bin sort & count) throughput on MI355X.

Metric (BASELINE.json): stage-2 Gk-mers/s (+ unique k-mers/s), k=27. One "step" = one pass of the whole hot path
(index -> expand -> histogram -> 7 LSD scatter passes -> compaction) over one resident bin image.
Workload at N=1 = BASELINE.json configs[1]: k=27, 150 bp synthetic reads, ~2 Gbp, ALL k-mers as a single bin
(13.3 M reads of a 66 Mbp genome -> ~1.65 G k-mers, 13.2 GB of 8-byte records) on one GPU.
N>1: one process per GPU, every rank sorts its own bin of that size (weak scaling, no data-path collective);
the four tallies are summed once with torch.distributed all_reduce (RCCL) inside the timed region.

Inputs are resident in HBM when the timed region starts (the PCIe-inclusive rate is in DESIGN.md, not here).
`roofline`: the dominant kernel is k_onesweep (one launch = one 8-bit pass over <= 2^29 records); achieved =
algorithmic bytes of a pass (2*W = 16 B per record) / its average launch time measured with HIP
events on the library's own stream; peak = 8 TB/s (MI355X_MICROARCH.md). `cpu_baseline`: the REAL reference
(oracle/_ref/kmc, built from /root/reference by oracle/Makefile) timed on this box's host cores on the same
workload (its stage 2 has ~1 s of fixed cost, so a smaller sample would under-report it) — a reported baseline only.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from kmc_amd import capi, sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
PMC_PROFILE = os.path.join(ROOT, "profiles", "r01", "final", "pmc_hbm_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes


def pmc_traffic(kernel: str, records_per_launch: float):
    """HBM bytes per launch of `kernel` from the committed PMC profile of THIS workload (null for any other workload:
    counters cannot be collected from inside the timed run)."""
    try:
        d = json.load(open(PMC_PROFILE))
        if abs(d["k_onesweep_records_per_launch_avg"] - records_per_launch) > 0.01 * records_per_launch:
            return None
        return d["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(k: int, sample_reads: int, genome_len: int, runs: int = 2):
    """Reference KMC stage 2 on the host cores. The default sample is the WHOLE N=1 workload (13.3 M reads): the
    reference's stage 2 carries about a second of fixed cost (sorter calibration, arena initialisation), so a small
    sample under-reports it by 2x; the full workload is ~7 s per kmc run plus ~15 s of FASTQ writing."""
    exe = os.path.join(ROOT, "oracle", "_ref", "kmc")
    cores = os.cpu_count() or 1
    if os.path.exists(exe):
        from kmc_amd import synth

        # the FASTQ (~316 B per read), kmc's temporary bins and its output need room; prefer RAM-backed storage
        need = int(sample_reads * 316 * 1.7) + (1 << 28)
        cands = [d for d in ("/dev/shm", os.environ.get("TMPDIR") or "/tmp", ROOT) if os.path.isdir(d)]
        space = {d: shutil.disk_usage(d).free for d in cands}
        best = next((d for d in cands if space[d] >= need), max(cands, key=lambda d: space[d]))
        if space[best] < need:  # shrink the sample rather than fail
            sample_reads = max(int(sample_reads * space[best] / need * 0.9), 100_000)
        with tempfile.TemporaryDirectory(dir=best) as td:
            fq = os.path.join(td, "s.fq")
            synth.make_fastq(fq, seed=2026, genome_len=genome_len, n_reads=sample_reads)
            times, total, uniq = [], 0, 0
            threads = min(cores, 128)
            ram_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") >> 30
            mem = max(2, min(128, ram_gb // 2))
            for i in range(runs):
                tmp = os.path.join(td, f"t{i}")
                os.makedirs(tmp)
                r = subprocess.run([exe, f"-k{k}", f"-t{threads}", f"-m{mem}", "-hp", fq, os.path.join(td, "o"), tmp], capture_output=True, text=True)
                if r.returncode != 0:
                    break
                m = re.search(r"2nd stage:\s*([0-9.eE+-]+)s", r.stdout)
                t = re.search(r"Total no. of k-mers\s*:\s*(\d+)", r.stdout)
                u = re.search(r"No. of unique k-mers\s*:\s*(\d+)", r.stdout)
                if not (m and t):
                    break
                times.append(float(m.group(1)))
                total, uniq = int(t.group(1)), int(u.group(1)) if u else 0
            if times:
                t2 = min(times)
                return {"value": total / t2 / 1e9, "unit": "Gk-mers/s", "cores": threads, "kind": "reference",
                        "sample": f"reference kmc 3.2.4 -k{k} -t{threads} -m{mem}, '2nd stage' wall, best of {len(times)}; {sample_reads} reads x150bp "
                                  f"of a {genome_len} bp genome = {total} k-mers ({uniq} unique)",
                        "stage2_s": t2, "unique_kmers_per_s": uniq / t2}
    # no reference binary on this box: time the single-threaded C oracle (a port) on a smaller sample
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O

    (img, nrec, _, _), = capi.synth_bins(seed=2026, genome_len=genome_len // 10, n_reads=sample_reads // 10, k=k, n_bins=1)
    t0 = time.time()
    _, _, st = O.process_bin(O.make_params(k), img, nrec)
    dt = time.time() - t0
    return {"value": nrec / dt / 1e9, "unit": "Gk-mers/s", "cores": 1, "kind": "port",
            "sample": f"oracle/stage2_oracle.c single thread on {nrec} k-mers", "stage2_s": dt, "unique_kmers_per_s": float(st[0]) / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--reads", type=int, default=13_300_000, help="reads per GPU (150 bp); default = configs[1] (~2 Gbp)")
    ap.add_argument("--genome", type=int, default=66_000_000)
    ap.add_argument("--lut-prefix", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="reads of the CPU-baseline sample (0 = the whole --reads workload)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU path")
    # one rank per GPU; if the launcher already narrowed the visible devices to one per process, that one is ordinal 0
    dev_index = local_rank if torch.cuda.device_count() > local_rank else 0
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    dev = torch.device("cuda", dev_index)

    ctx = capi.Context((dev_index,))
    k = args.k
    t_gen = time.time()
    n_threads = max(1, (os.cpu_count() or 8) // max(world, 1))
    (img, n_rec, packs, n_super), = capi.synth_bins(seed=2026 + rank, genome_len=args.genome, n_reads=args.reads, k=k, n_bins=1, n_threads=n_threads)
    t_gen = time.time() - t_gen
    p = capi.make_params(k, lut_prefix_len=args.lut_prefix)
    rec_bytes = ctx.out_rec_bytes(p)
    out_cap = ((n_rec + 1) // 2) * rec_bytes
    lut_n = ctx.lut_entries(p)
    pack_start = np.concatenate([[0], np.cumsum(packs)]).astype(np.uint64)

    d_in = ctx.malloc(img.size + 256)
    d_ps = ctx.malloc(pack_start.nbytes)
    d_out = ctx.malloc(out_cap + 256)
    d_lut = ctx.malloc(max(lut_n, 1) * 8)
    d_small = ctx.malloc(64)
    d_stats, d_ob = d_small, d_small + 32
    t_h2d = time.time()
    ctx.h2d(d_in, img)
    ctx.h2d(d_ps, pack_start)
    t_h2d = time.time() - t_h2d

    def step():
        ctx.process_bin_device(p, d_in, img.size, n_rec, d_ps, packs.size, d_out, out_cap, d_ob, d_lut, d_stats, sync=True)

    for _ in range(args.warmup):
        step()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    small = np.zeros(8, dtype=np.uint64)
    ctx.d2h(small, d_small)
    tallies = sharding.allreduce_tallies(small[:4], device=dev)  # the one RCCL collective of the path (32 bytes)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ntot = torch.tensor([n_rec], dtype=torch.int64, device=dev)
        dist.all_reduce(ntot)
        n_total_all = int(ntot.item())
    else:
        n_total_all = n_rec

    timings = ctx.last_timings()
    n_launch, sc_ms, sc_keys = ctx.last_scatter_stats()
    if rank == 0:
        W = 8 * ((k + 31) // 32)
        avg_ms = sc_ms / max(n_launch, 1)
        achieved = (2 * W * sc_keys) / (avg_ms * 1e-3) / 1e9 if n_launch else 0.0
        value = n_total_all * args.steps / dt / 1e9
        res = {
            "metric": "stage-2 Gk-mers/s, k=%d (bin sort & count: expand + 8-bit LSD radix sort + compaction), bit-exact vs reference KMC" % k,
            "value": value, "unit": "Gk-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[1]: k=%d, 150 bp synthetic reads, %.2f Gbp per GPU, all k-mers of a GPU as a single bin (radix sort + count on one MI355X each)" % (k, args.reads * 150 / 1e9),
                       "kmers_per_gpu": n_rec, "superkmers_per_gpu": n_super, "bin_image_bytes": int(img.size), "record_bytes": W,
                       "radix_passes": (2 * k + 7) // 8, "cutoff_min": 2, "counter_max": 255, "lut_prefix_len": args.lut_prefix,
                       "parallelism": "bins sharded, 1 process/GPU, tallies all-reduced (RCCL)" if world > 1 else "1 GPU"},
            "unique_kmers_per_s": float(tallies[0]) * args.steps / dt,
            "tallies": {"n_unique": int(tallies[0]), "n_cutoff_min": int(tallies[1]), "n_cutoff_max": int(tallies[2]), "n_total": int(tallies[3])},
            "phases_ms_last_step": timings,
            "stage2_algorithmic_bytes_per_kmer": W * (2 * ((2 * k + 7) // 8) + 3),
            "stage2_algorithmic_GBs": W * (2 * ((2 * k + 7) // 8) + 3) * n_rec / (timings["total"] * 1e-3) / 1e9,
            "roofline": {"bound": "hbm", "kernel": "k_onesweep<%d>" % ((k + 31) // 32), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("k_onesweep<%d>" % ((k + 31) // 32), sc_keys),
                         "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01/final/pmc_hbm_traffic.json)",
                         "algorithmic_bytes_per_launch": 2 * W * sc_keys, "launches_per_step": n_launch, "avg_launch_ms": avg_ms,
                         "records_per_launch": sc_keys, "algorithmic_bytes_per_record_per_launch": 2 * W},
            "setup_s": {"generate": t_gen, "h2d": t_h2d, "h2d_GBs": img.size / max(t_h2d, 1e-9) / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(k, args.cpu_sample_reads or args.reads, args.genome if not args.cpu_sample_reads else max(args.genome * args.cpu_sample_reads // args.reads, 1_000_000))
            except Exception as e:  # the baseline is informative; never lose the GPU number over it
                res["cpu_baseline"] = {"value": None, "unit": "Gk-mers/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
        print(json.dumps(res))
    if dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Builds tuning variants of libkmc_hip.so into kmc_amd/variants/ (macro overrides of tile geometry) so one gpurun
call can A/B them: `KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_<name>.so python bench.py ...`."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {
    "base": [],
    "trace": ["-DKMC_TRACE"],  # per-tile phase stamps, read by tools/trace_run.py
    "k2": ["-DRS_LOOKBACK_K=2"],
    "k8": ["-DRS_LOOKBACK_K=8"],
    "b512x16": ["-DRS_BLOCK_THREADS=512", "-DRS_WORDS_PER_THREAD=16", "-DRS_MIN_WAVES=6"],
    "w8": ["-DRS_WORDS_PER_THREAD_1=8"],
    "w16s4": ["-DRS_WORDS_PER_THREAD_1=16", "-DRS_STAGES=4"],  # 16 K-record tiles (512-byte runs), 2 workgroups/CU, ~19 VGPRs spilled
    "w16s4_1cu": ["-DRS_WORDS_PER_THREAD_1=16", "-DRS_STAGES=4", "-DRS_MIN_WAVES=4"],  # same tile, 128 VGPRs, 1 workgroup/CU
    # one workgroup per CU (128 VGPRs): bigger tiles = longer runs per digit (tools/ubench_scatter.hip: misaligned 320-byte runs 2.8 TB/s, 640: 3.3, 1024: 4.1)
    "w16s2_1cu": ["-DRS_WORDS_PER_THREAD_1=16", "-DRS_STAGES=2", "-DRS_MIN_WAVES=4"],
    "w16s1_1cu": ["-DRS_WORDS_PER_THREAD_1=16", "-DRS_STAGES=1", "-DRS_MIN_WAVES=4"],
    "w20s4_1cu": ["-DRS_WORDS_PER_THREAD_1=20", "-DRS_STAGES=4", "-DRS_MIN_WAVES=4"],
    "w20s2_1cu": ["-DRS_WORDS_PER_THREAD_1=20", "-DRS_STAGES=2", "-DRS_MIN_WAVES=4"],
    "w24s4_1cu": ["-DRS_WORDS_PER_THREAD_1=24", "-DRS_STAGES=4", "-DRS_MIN_WAVES=4"],
    "w24s2_1cu": ["-DRS_WORDS_PER_THREAD_1=24", "-DRS_STAGES=2", "-DRS_MIN_WAVES=4"],
    "w32s4_1cu": ["-DRS_WORDS_PER_THREAD_1=32", "-DRS_STAGES=4", "-DRS_MIN_WAVES=4"],
    "wm12": ["-DRS_WORDS_PER_THREAD=12"],  # records of 2+ words: 12 words per thread in a scatter tile (k = 55: 6144 records, 384-byte runs)
    "wm16s4": ["-DRS_WORDS_PER_THREAD=16", "-DRS_STAGES=4"],  # 16 words (k = 55: 8192 records / k = 127: 4096, 512-byte runs), staged in 4 slices
    "wm12s3": ["-DRS_WORDS_PER_THREAD=12", "-DRS_STAGES=3"],
    "w12s3": ["-DRS_WORDS_PER_THREAD_1=12", "-DRS_STAGES=3"],  # 12 K-record tiles, 384-byte runs, 2 workgroups/CU
    "w10s2": ["-DRS_WORDS_PER_THREAD_1=10", "-DRS_STAGES=2"],
    "exp16k": ["-DEXP_CHUNK_BYTES=16384"],
    "exp4k": ["-DEXP_CHUNK_BYTES=4096"],
    "exp8k": ["-DEXP_CHUNK_BYTES=8192"],
    "lb1": ["-DLB64_WINDOWS=1"],  # 64-bit look-back (compaction, expand): 64-tile windows per round trip
    "lb4": ["-DLB64_WINDOWS=4"],
    "lb16": ["-DLB64_WINDOWS=16"],
    "parse16": ["-DPARSE_CAND_POS=16"],
    "parse40": ["-DPARSE_CAND_POS=40"],
    "exp8kb256": ["-DEXP_CHUNK_BYTES=8192", "-DEXP_BLOCK_THREADS=256"],
    "cp512": ["-DCP_BLOCK_THREADS=512"],
    "w10s5": ["-DRS_WORDS_PER_THREAD_1=10", "-DRS_STAGES=5"],
    "w9s3": ["-DRS_WORDS_PER_THREAD_1=9", "-DRS_STAGES=3"],
    "w11s1": ["-DRS_WORDS_PER_THREAD_1=11", "-DRS_STAGES=1"],
    "w10s2k2": ["-DRS_WORDS_PER_THREAD_1=10", "-DRS_STAGES=2", "-DRS_LOOKBACK_K=2"],
    "w10s2k8": ["-DRS_WORDS_PER_THREAD_1=10", "-DRS_STAGES=2", "-DRS_LOOKBACK_K=8"],
    "exp16kb1024": ["-DEXP_CHUNK_BYTES=16384", "-DEXP_BLOCK_THREADS=1024"],
    "combo1": ["-DRS_WORDS_PER_THREAD_1=10", "-DRS_STAGES=2", "-DEXP_CHUNK_BYTES=16384", "-DCP_BLOCK_THREADS=512"],
    "exp8k": ["-DEXP_CHUNK_BYTES=8192"],  # k_expand occupancy: 27 KB of LDS per workgroup (4 per CU) instead of 49.5 (3)
    "ind3": ["-DINDIRECT_MIN_WORDS=3"],  # two-word records through the HBM passes themselves (the default sorts them through pairs too: 16-byte gathers, half of every HBM sector wasted, still +7 %)
    "brind1": ["-DBR_INDIRECT_MIN_SIZE=1"],  # k_bucket_rank<1> with the indirect-sort branches decided at run time (as before the compile-time guard)
    "expcut1": ["-DEXP_CUT=1"],  # k_expand without its k-mer loop (phase costs; output garbage)
    "nohist": ["-DEXP_NO_HIST"],  # expand without the fused histograms (sort output is garbage): what do the LDS atomics cost?
    "exp256": ["-DEXP_BLOCK_THREADS=256"],
    "cp_r16": ["-DCP_WORDS_PER_THREAD_1=16"],  # one-word records: 16 rows per wave (8192-record tiles; 64 VGPRs force spills)
    "cp_bidx": ["-DCP_TILE_FROM_BLOCKIDX=1"],  # compaction tiles in blockIdx order (no ticket atomic)
    "rs_bidx": ["-DRS_TILE_FROM_BLOCKIDX=1"],  # scatter tiles in blockIdx order
    "bidx2": ["-DCP_TILE_FROM_BLOCKIDX=1", "-DRS_TILE_FROM_BLOCKIDX=1"],
    "cpmw6": ["-DCP_MIN_WAVES_1=6"],  # k_compact<1> with 80 VGPRs (no spills?) instead of 64 + 30 spilled
    "cpmw5": ["-DCP_MIN_WAVES_1=5"],
    "br512": ["-DBR_THREADS=512"],  # k_bucket_rank geometry: 8 waves (x 8 / 4 / 2 rows by record width)
    "br1024": ["-DBR_THREADS=1024", "-DBR_MIN_WAVES=4"],
    "brslack6": ["-DBR_SLACK_DIV=6"],  # 768 x 8 with windows of 5120 records (1024 of slack)
    "brslack12": ["-DBR_SLACK_DIV=12"],  # windows of 5632 (512 of slack)
    "brslack24": ["-DBR_SLACK_DIV=24"],  # windows of 5888 (256 of slack; longer tiles take a second chunk)
    # round 6: buckets beyond BR_MID records go through the arena (arena_sort.hip.h) instead of the whole workgroup's pairwise walk
    "brmid175": ["-DBR_MID=175"], "brmid383": ["-DBR_MID=383"], "brmid767": ["-DBR_MID=767"], "brmid1535": ["-DBR_MID=1535"],
    "brnl0": ["-DBR_NARROW_LOOP=0"],  # k_bucket_rank, 32-bit pairs: the walk of rounds 3-5 (steps of 4 + a pairwise tail, interleaved by the compiler); the shipped one is 8 / 4 / one masked step
    "brwl0": ["-DBR_WIDE_LOOP=0", "-DBR_NARROW_LOOP=2"],  # k_bucket_rank<1>, 64-bit pairs: steps of 2 + one, and the dealers' count 8 + one by one (rounds 3-5)
    "brwl4": ["-DBR_WIDE_STEP=4"],  # steps of 4, 2, 1
    "br1": ["-DBR_STOP_AFTER=1"],  # k_bucket_rank cut after: 1 loads + bucket starts + table look-ups, 2 + pairs + rank loops, 3 + records placed in order,
    "br2": ["-DBR_STOP_AFTER=2"],  # 4 + run tails, counts, classes, ranks of the counted k-mers (garbage output): phase costs
    "br3": ["-DBR_STOP_AFTER=3"],
    "br4": ["-DBR_STOP_AFTER=4"],
}


def main(names):
    out = os.path.join(ROOT, "kmc_amd", "variants")
    os.makedirs(out, exist_ok=True)
    procs = []
    for n in names or VARIANTS:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", *VARIANTS[n],
               os.path.join(ROOT, "kmc_amd", "csrc", "kmc_hip.hip"), "-o", os.path.join(out, f"libkmc_hip_{n}.so"), "-lrccl"]
        procs.append((n, subprocess.Popen(cmd)))
    for n, p in procs:
        assert p.wait() == 0, n
        print("built", n)


if __name__ == "__main__":
    main(sys.argv[1:])

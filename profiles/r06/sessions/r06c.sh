#!/bin/bash
# round 6, session c: the arena path is right under the emulation and wrong on the device — where? (arena checked on the host between passes and finish; persistent grids of 1)
OUT=gpurun_out/r06c; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export BB_CASES="2000:10:0,3000:20:5 2000:300:10"
run() { tag=$1; shift; env "$@" timeout 300 python tools/debug/bigbucket_gpu.py > $OUT/$tag.txt 2>&1; echo "== $tag"; grep -c True, $OUT/$tag.txt; grep "arena debug" $OUT/$tag.txt | cut -c1-250; tail -1 $OUT/$tag.txt | cut -c1-200; }
run debug KMC_HIP_ARENA_DEBUG=1
run all1 KMC_HIP_ARENA_GRID_GATHER=1 KMC_HIP_ARENA_GRID_SWEEP=1 KMC_HIP_ARENA_GRID_FINISH=1 KMC_HIP_ARENA_GRID_HEAVY=1
run gather1 KMC_HIP_ARENA_GRID_GATHER=1
run sweep1 KMC_HIP_ARENA_GRID_SWEEP=1
run finish1 KMC_HIP_ARENA_GRID_FINISH=1
run heavy1 KMC_HIP_ARENA_GRID_HEAVY=1

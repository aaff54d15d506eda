#!/bin/bash
# round 6, session d: the arena step by step against the host (KMC_HIP_ARENA_DEBUG=2): gather, histograms, every pass
OUT=gpurun_out/r06d; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export BB_CASES="2000:10:0,3000:20:5"
run() { tag=$1; shift; env "$@" timeout 300 python tools/debug/bigbucket_gpu.py > $OUT/$tag.txt 2>&1; echo "== $tag"; grep "arena debug" $OUT/$tag.txt | cut -c1-250 | head -40; tail -1 $OUT/$tag.txt | cut -c1-200; }
run steps KMC_HIP_ARENA_DEBUG=2
run static KMC_HIP_ARENA_DEBUG=3

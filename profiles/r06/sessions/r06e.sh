#!/bin/bash
# round 6, session e: k_onesweep_dyn with the 32-bit tile length (the 64-bit compare had compiled to a select on a stale SCC): the arena step by step, then the repeat-rich bins, then the quarter legs
OUT=gpurun_out/r06e; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
BB_CASES="2000:10:0,3000:20:5" KMC_HIP_ARENA_DEBUG=2 timeout 300 python tools/debug/bigbucket_gpu.py > $OUT/steps.txt 2>&1; grep "arena debug" $OUT/steps.txt | cut -c1-200 | head -12; tail -1 $OUT/steps.txt | cut -c1-200
BB_CASES="5000:6:0 2000:10:0,3000:20:5 300:1500:5,5000:6:0 2000:300:10 H30000,1000:100:10 300:100000:120,171:20000:20,H20000" timeout 600 python tools/debug/bigbucket_gpu.py > $OUT/bigbucket.txt 2>&1; grep -c "True, True, True" $OUT/bigbucket.txt; tail -1 $OUT/bigbucket.txt | cut -c1-300
SPEC=300:100000:120,6000:5000:20,171:100000:20,H20000
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest --also-two-streams"
run() { tag=$1; shift; env "$@" timeout 500 python bench.py $Q > $OUT/$tag.json 2> $OUT/$tag.err; python tools/pj.py $OUT/$tag.json 2>&1 | cut -c1-700; tail -2 $OUT/$tag.err | cut -c1-300; }
run uniform A=1
run skew KMC_SYNTH_REPEATS=10000:2000:10
run spectrum KMC_SYNTH_REPEATS=$SPEC

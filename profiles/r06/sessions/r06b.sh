#!/bin/bash
# round 6, session b: the arena path (buckets beyond BR_MID records sorted together by HBM passes) — correctness on repeat-rich bins at product geometry, then the quarter legs
OUT=gpurun_out/r06b; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
BB_CASES="5000:6:0 2000:10:0,3000:20:5 300:1500:5,5000:6:0 2000:300:10 H30000,1000:100:10 300:100000:120,171:20000:20,H20000" timeout 600 python tools/debug/bigbucket_gpu.py > $OUT/bigbucket.txt 2>&1; tail -4 $OUT/bigbucket.txt | cut -c1-300
SPEC=300:100000:120,6000:5000:20,171:100000:20,H20000
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest --also-two-streams"
run() { tag=$1; shift; env "$@" timeout 500 python bench.py $Q > $OUT/$tag.json 2> $OUT/$tag.err; python tools/pj.py $OUT/$tag.json 2>&1 | cut -c1-600; tail -3 $OUT/$tag.err | cut -c1-300; }
run uniform A=1
run uniform_noarena KMC_HIP_ARENA=0
run skew KMC_SYNTH_REPEATS=10000:2000:10
run spectrum KMC_SYNTH_REPEATS=$SPEC

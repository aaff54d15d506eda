#!/bin/bash
# round 6, session a: this round's box, the tree of round 5's end — quarter workload uniform / skew / spectrum (the baseline the round's finisher work is measured against)
OUT=gpurun_out/r06a; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
SPEC=300:100000:120,6000:5000:20,171:100000:20,H20000
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest --also-two-streams"
run() { tag=$1; shift; env "$@" timeout 500 python bench.py $Q > $OUT/$tag.json 2> $OUT/$tag.err; python tools/pj.py $OUT/$tag.json 2>&1 | cut -c1-400; }
run uniform A=1
run skew KMC_SYNTH_REPEATS=10000:2000:10
run spectrum KMC_SYNTH_REPEATS=$SPEC

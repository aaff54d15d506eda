#!/bin/bash
# round 6, session k: detector stride 176 / BR_MID 351, k_arena_finish with 256-thread workgroups: correctness, the three legs, kernel statistics
OUT=gpurun_out/r06k; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
BB_CASES="5000:6:0 2000:10:0,3000:20:5 300:1500:5,5000:6:0 2000:300:10 H30000,1000:100:10 300:100000:120,171:20000:20,H20000" timeout 600 python tools/debug/bigbucket_gpu.py > $OUT/bigbucket.txt 2>&1; grep -c "True, True, True" $OUT/bigbucket.txt; tail -1 $OUT/bigbucket.txt | cut -c1-300
SPEC=300:100000:120,6000:5000:20,171:100000:20,H20000
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest --also-two-streams"
run() { tag=$1; shift; env "$@" timeout 500 python bench.py $Q > $OUT/$tag.json 2> $OUT/$tag.err; python tools/pj.py $OUT/$tag.json 2>&1 | cut -c1-110; python - $OUT/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("   two streams", round(d.get("value_two_streams") or 0,2), "oracle", d["self_check"].get("oracle_bins_equal"), "local sort ms", round(d["local_sort"]["avg_launch_ms"],3))
PY
}
run uniform A=1
run skew KMC_SYNTH_REPEATS=10000:2000:10
run spectrum KMC_SYNTH_REPEATS=$SPEC
bash tools/gpu_session.sh r06k profk:27:A=1 profk:27:KMC_SYNTH_REPEATS=10000:2000:10 profk:27:KMC_SYNTH_REPEATS=$SPEC > /dev/null 2>&1
mv "$OUT/profk_27_A=1" $OUT/p_uniform; mv "$OUT/profk_27_KMC_SYNTH_REPEATS=10000:2000:10" $OUT/p_skew; mv "$OUT/profk_27_KMC_SYNTH_REPEATS=$SPEC" $OUT/p_spectrum

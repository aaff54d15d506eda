#!/bin/bash
# round 6, session i: buckets beyond BR_MID found by sampling (k_bucket_detect) BEFORE k_bucket_rank, which then runs once; BR_MID 175 / 255 / 383 / 767
OUT=gpurun_out/r06i; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
BB_CASES="5000:6:0 2000:10:0,3000:20:5 300:1500:5,5000:6:0 2000:300:10 H30000,1000:100:10 300:100000:120,171:20000:20,H20000" timeout 600 python tools/debug/bigbucket_gpu.py > $OUT/bigbucket.txt 2>&1; grep -c "True, True, True" $OUT/bigbucket.txt; tail -1 $OUT/bigbucket.txt | cut -c1-300
SPEC=300:100000:120,6000:5000:20,171:100000:20,H20000
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest"
run() { tag=$1; lib=$2; shift 2; env KMC_HIP_LIB=$lib "$@" timeout 500 python bench.py $Q > $OUT/$tag.json 2> $OUT/$tag.err; python tools/pj.py $OUT/$tag.json 2>&1 | cut -c1-110; python - $OUT/$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("   oracle", d["self_check"].get("oracle_bins_equal"), "local sort ms", round(d["local_sort"]["avg_launch_ms"],3), d["sort_path"].get("groups_by_path"))
PY
}
for v in base brmid175 brmid383 brmid767; do
  lib=kmc_amd/variants/libkmc_hip_$v.so; [ $v = base ] && lib=kmc_amd/libkmc_hip.so
  run skew_$v $lib KMC_SYNTH_REPEATS=10000:2000:10
  run spec_$v $lib KMC_SYNTH_REPEATS=$SPEC
done
run uniform kmc_amd/libkmc_hip.so A=1
run uniform_noarena kmc_amd/libkmc_hip.so KMC_HIP_ARENA=0
run uniform2 kmc_amd/libkmc_hip.so A=1
run uniform_noarena2 kmc_amd/libkmc_hip.so KMC_HIP_ARENA=0

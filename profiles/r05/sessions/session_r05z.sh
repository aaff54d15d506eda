#!/bin/bash
# round 5, session z: the new default walk (steps of 8 / 4 / one masked step, also in the dealers' count; masked tail for A/B pairs) against the old ones (brnl0, brpt0), same box
OUT=gpurun_out/r05zz; mkdir -p $OUT /dev/shm/kmccache /dev/shm/kmccache_skew /dev/shm/kmccache_spec
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q0="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 4 --warmup 1 --no-digest"
show() { python - <<PY
import json
try:
    d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("   $1: value %.2f, %.1f ms/step, local_sort %.3f ms, onesweep %.1f us, oracle %s" % (d["value"], d["ms_per_step"], d["local_sort"]["avg_launch_ms"], 1e3*d["roofline"]["avg_launch_ms"], d["self_check"].get("oracle_bins_equal")))
except Exception as e: print("   $1: ", e)
PY
}
run() { tag=$1; lib=$2; k=$3; cache=$4; shift; shift; shift; shift
  env KMC_HIP_LIB=$lib "$@" timeout 600 python bench.py --k $k $Q0 --cache /dev/shm/$cache > $OUT/$tag.json 2> $OUT/$tag.err; show $tag
}
NEW=kmc_amd/libkmc_hip.so; OLD=kmc_amd/variants/libkmc_hip_brnl0.so; OLD2=kmc_amd/variants/libkmc_hip_brpt0.so
SK=KMC_SYNTH_REPEATS=10000:2000:10; SP=KMC_SYNTH_REPEATS=300:100000:120,6000:5000:20,171:100000:20,H20000
run new_a $NEW 27 kmccache A=1; run old_a $OLD 27 kmccache A=1; run new_b $NEW 27 kmccache A=1; run old_b $OLD 27 kmccache A=1
run new_skew $NEW 27 kmccache_skew $SK; run old_skew $OLD 27 kmccache_skew $SK; run new_skew_b $NEW 27 kmccache_skew $SK
run new_spec $NEW 27 kmccache_spec $SP; run old_spec $OLD 27 kmccache_spec $SP; run new_spec_b $NEW 27 kmccache_spec $SP
run new_k55 $NEW 55 kmccache A=1; run old_k55 $OLD2 55 kmccache A=1; run new_k55_b $NEW 55 kmccache A=1; run old_k55_b $OLD2 55 kmccache A=1
run new_skew55 $NEW 55 kmccache_skew $SK; run old_skew55 $OLD2 55 kmccache_skew $SK
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rank_path or repeat or giant or big_buckets" 2>&1 | tail -2

#!/bin/bash
# round 5, session y: more shapes of the same walk (brnl2, 4, 5)
OUT=gpurun_out/r05y; mkdir -p $OUT /dev/shm/kmccache
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 4 --warmup 1 --no-digest --cache /dev/shm/kmccache"
show() { python - <<PY
import json
try:
    d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("   $1: value %.2f, %.1f ms/step, local_sort %.3f ms, onesweep %.1f us, oracle %s" % (d["value"], d["ms_per_step"], d["local_sort"]["avg_launch_ms"], 1e3*d["roofline"]["avg_launch_ms"], d["self_check"].get("oracle_bins_equal")))
except Exception as e: print("   $1: ", e)
PY
}
run() { tag=$1; lib=$2; shift; shift
  env KMC_HIP_LIB=$lib "$@" timeout 600 python bench.py --k 27 $Q > $OUT/$tag.json 2> $OUT/$tag.err; show $tag
}
run base_a kmc_amd/libkmc_hip.so A=1
for v in brnl2 brnl4 brnl5; do run $v kmc_amd/variants/libkmc_hip_$v.so A=1; done
run base_b kmc_amd/libkmc_hip.so A=1
for v in brnl2 brnl4 brnl5; do run ${v}_b kmc_amd/variants/libkmc_hip_$v.so A=1; done
mkdir -p /dev/shm/kmccache_skew; Q="${Q/kmccache/kmccache_skew}"   # (the cache key does not know the repeats)
run base_skew kmc_amd/libkmc_hip.so KMC_SYNTH_REPEATS=10000:2000:10
for v in brnl2 brnl4 brnl5; do run ${v}_skew kmc_amd/variants/libkmc_hip_$v.so KMC_SYNTH_REPEATS=10000:2000:10; done

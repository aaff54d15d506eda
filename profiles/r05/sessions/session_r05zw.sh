#!/bin/bash
# round 5, the last GPU seconds: k_bucket_rank<1>'s walk over 64-bit pairs in walk32's shape (steps of 8 / 4 / 2 / 1) against steps of 2 (brwl0) — k = 32 and the repeat-rich legs
OUT=gpurun_out/r05zw; mkdir -p $OUT /dev/shm/kc32 /dev/shm/kcspec /dev/shm/kcskew /dev/shm/kc27
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
left() { echo $(( 165 - ( $(date +%s) - T0 ) )); }
Q0="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest"
show() { python - <<PY
import json
try:
    d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("   $1: value %.2f, %.1f ms/step, local_sort %.3f ms, oracle %s" % (d["value"], d["ms_per_step"], d["local_sort"]["avg_launch_ms"], d["self_check"].get("oracle_bins_equal")))
except Exception as e: print("   $1: ", e)
PY
}
run() { tag=$1; lib=$2; k=$3; cache=$4; shift; shift; shift; shift
  [ $(left) -gt 12 ] || { echo "   $tag: skipped (time)"; return; }
  env KMC_HIP_LIB=$lib "$@" timeout 60 python bench.py --k $k $Q0 --cache /dev/shm/$cache > $OUT/$tag.json 2> $OUT/$tag.err; show $tag
}
NEW=kmc_amd/libkmc_hip.so; OLD=kmc_amd/variants/libkmc_hip_brwl0.so; W4=kmc_amd/variants/libkmc_hip_brwl4.so
SP=KMC_SYNTH_REPEATS=300:100000:120,6000:5000:20,171:100000:20,H20000; SK=KMC_SYNTH_REPEATS=10000:2000:10
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rank_path or repeat or giant or big_buckets or skew or wide or k32 or 64" 2>&1 | tail -2
run new_spec $NEW 27 kcspec $SP; run old_spec $OLD 27 kcspec $SP
run new_k32 $NEW 32 kc32 A=1; run old_k32 $OLD 32 kc32 A=1; run w4_k32 $W4 32 kc32 A=1
run new_skew $NEW 27 kcskew $SK; run old_skew $OLD 27 kcskew $SK
run w4_spec $W4 27 kcspec $SP
run new_27 $NEW 27 kc27 A=1; run old_27 $OLD 27 kc27 A=1
echo "elapsed $(( $(date +%s) - T0 )) s"

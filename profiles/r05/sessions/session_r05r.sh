#!/bin/bash
OUT=gpurun_out/r05r; mkdir -p $OUT /dev/shm/kmccache
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest --no-oracle-check"
show() { python - <<PY
import json
d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("   $1: value %.2f, %.1f ms/step, local_sort %.3f ms, onesweep %.1f us" % (d["value"], d["ms_per_step"], d["local_sort"]["avg_launch_ms"], 1e3*d["roofline"]["avg_launch_ms"]))
PY
}
for v in big64 big32 big256 gt512; do
  KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_$v.so KMC_SYNTH_REPEATS=10000:2000:10 timeout 600 python bench.py --k 27 $Q > $OUT/skew_$v.json 2> $OUT/skew_$v.err; show skew_$v
done
for v in big64 big32; do
  KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_$v.so timeout 600 python bench.py --cache /dev/shm/kmccache --k 27 $Q > $OUT/uni_$v.json 2> $OUT/uni_$v.err; show uni_$v
done
timeout 600 python bench.py --cache /dev/shm/kmccache --k 27 $Q > $OUT/uni_base.json 2> $OUT/uni_base.err; show uni_base
KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_gt512.so KMC_SYNTH_REPEATS=300:100000:120,6000:5000:20,171:100000:20,H20000 timeout 600 python bench.py --k 27 $Q > $OUT/spec_gt512.json 2> $OUT/spec_gt512.err; show spec_gt512

#!/bin/bash
OUT=gpurun_out/r05h; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 2 --warmup 0 --no-digest"
show() { python - <<PY
import json
d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("   $1: value %.2f, local_sort %.3f ms, %s" % (d["value"], d["local_sort"]["avg_launch_ms"], [(b["bin"], b["equal"]) for b in d["self_check"]["oracle_bins"]]))
PY
}
export KMC_SYNTH_REPEATS=10000:2000:10
timeout 600 python bench.py --k 27 $Q > $OUT/shipped.json 2> $OUT/shipped.err; show shipped
KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_nobig.so timeout 600 python bench.py --k 27 $Q > $OUT/nobig.json 2> $OUT/nobig.err; show nobig
KMC_HIP_GROUP=1 timeout 600 python bench.py --k 27 $Q > $OUT/shipped_g1.json 2> $OUT/shipped_g1.err; show shipped_g1

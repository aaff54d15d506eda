#!/bin/bash
OUT=gpurun_out/r05t; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest --no-oracle-check"
show() { python - <<PY
import json
d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("   $1: value %.2f, %.1f ms/step" % (d["value"], d["ms_per_step"]))
PY
}
for n in 1 2 3; do
KMC_SYNTH_REPEATS=10000:2000:10 timeout 600 python bench.py --k 27 $Q --streams $n > $OUT/skew_s$n.json 2> $OUT/skew_s$n.err; show skew_s$n
done
for n in 1 2; do
KMC_SYNTH_REPEATS=300:100000:120,6000:5000:20,171:100000:20,H20000 timeout 600 python bench.py --k 27 $Q --streams $n > $OUT/spec_s$n.json 2> $OUT/spec_s$n.err; show spec_s$n
done

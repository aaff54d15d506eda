#!/bin/bash
OUT=gpurun_out/r05p; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export BB_CASES="5000:6:0 5000:7:0 5000:9:0 5000:12:0 1000:100:10 2000:40:5,300:230:0"
echo "== shipped"; timeout 300 python tools/debug/bigbucket_gpu.py 2>&1 | tail -30 | cut -c1-130 | grep -v "True, True, True" | tee $OUT/a.txt
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest"
show() { python - <<PY
import json
d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("   $1: value %.2f, %.1f ms/step, local_sort %.3f ms, onesweep %.1f us, oracle %s, redo %s" % (d["value"], d["ms_per_step"], d["local_sort"]["avg_launch_ms"], 1e3*d["roofline"]["avg_launch_ms"], [b["equal"] for b in d["self_check"]["oracle_bins"]], d["local_sort"]["redo_groups"]))
PY
}
KMC_SYNTH_REPEATS=10000:2000:10 timeout 600 python bench.py --k 27 $Q > $OUT/skew.json 2> $OUT/skew.err; show skew
KMC_SYNTH_REPEATS=300:100000:120,6000:5000:20,171:100000:20,H20000 timeout 600 python bench.py --k 27 $Q > $OUT/spec.json 2> $OUT/spec.err; show spec
timeout 600 python bench.py --k 27 $Q > $OUT/uni.json 2> $OUT/uni.err; show uni
KMC_SYNTH_REPEATS=10000:2000:10 timeout 600 python bench.py --k 55 $Q > $OUT/skew55.json 2> $OUT/skew55.err; show skew55

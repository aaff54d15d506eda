#!/bin/bash
OUT=gpurun_out/r05q; mkdir -p $OUT /dev/shm/kmccache
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$PWD
export BB_CASES="5000:6:0 5000:9:0 5000:12:0 1000:100:10 2000:40:5,300:230:0"
echo "== shipped"; timeout 300 python tools/debug/bigbucket_gpu.py 2>&1 | tail -30 | cut -c1-130 | grep -v "True, True, True" | tee $OUT/a.txt
unset BB_CASES
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest"
show() { python - <<PY
import json
d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("   $1: value %.2f, %.1f ms/step, local_sort %.3f ms, onesweep %.1f us, oracle %s, redo %s" % (d["value"], d["ms_per_step"], d["local_sort"]["avg_launch_ms"], 1e3*d["roofline"]["avg_launch_ms"], [b["equal"] for b in d["self_check"]["oracle_bins"]], d["local_sort"]["redo_groups"]))
PY
}
prof() { tag=$1; shift
  cd /tmp; env "$@" timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/prof_$tag -o kt -- python $ROOT/bench.py --k 27 $Q > $ROOT/$OUT/$tag.json 2> $ROOT/$OUT/$tag.err; cd $ROOT
  find $OUT/prof_$tag -name "*kernel_trace.csv" -delete
  find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -r grep "k_bucket_rank\|k_giant" | cut -c1-28,150-240
  show $tag
}
prof skew KMC_SYNTH_REPEATS=10000:2000:10
prof spec KMC_SYNTH_REPEATS=300:100000:120,6000:5000:20,171:100000:20,H20000
prof uni A=1

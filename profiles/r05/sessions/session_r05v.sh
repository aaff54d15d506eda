#!/bin/bash
# round 5, session v: k_giant_tiles takes only the giant bucket (k_bucket_rank keeps the buckets in front of it), odd pass counts count in place
OUT=gpurun_out/r05v; mkdir -p $OUT /dev/shm/kmccache
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$PWD
export BB_CASES="300:1500:0 300:1500:5,5000:6:0 2000:300:10 H30000,1000:100:10"
timeout 300 python tools/debug/bigbucket_gpu.py 2>&1 | tail -24 | cut -c1-130 | grep -v "True, True, True" | tee $OUT/a.txt
unset BB_CASES
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "giant or repeat or rank_path or big_buckets" 2>&1 | tail -3
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest --also-two-streams"
show() { python - <<PY
import json
d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("   $1: value %.2f (two streams %.2f), %.1f ms/step, local_sort %.3f ms, oracle %s, giant %s redo %s" % (d["value"], d.get("value_two_streams") or 0, d["ms_per_step"], d["local_sort"]["avg_launch_ms"], [b["equal"] for b in d["self_check"]["oracle_bins"]], d["sort_path"]["groups_by_path"]["giant_tiles"], d["local_sort"]["redo_groups"]))
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
KMC_SYNTH_REPEATS=10000:2000:10 timeout 600 python bench.py --k 55 $Q > $OUT/skew55.json 2> $OUT/skew55.err; show skew55

#!/bin/bash
# round 5, session f: big buckets ranked by the whole workgroup + per-bin redo: uniform / one-family skew / spectrum quarter workloads (kernel statistics), then the rank / repeat tests
OUT=gpurun_out/r05f; mkdir -p $OUT /dev/shm/kmccache
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$PWD
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest"
show() { python - <<PY
import json
try:
    d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("   $1: value %.2f, %.1f ms/step, local_sort %.3f ms, onesweep %.1f us, oracle %s, giant %s/%s redo %s" % (d["value"], d["ms_per_step"], d["local_sort"]["avg_launch_ms"], 1e3*d["roofline"]["avg_launch_ms"], d["self_check"].get("oracle_bins_equal"), d["sort_path"]["groups_by_path"]["giant_tiles"], d["sort_path"]["groups_by_path"]["giant_records"], d["local_sort"]["redo_groups"]))
except Exception as e: print("   $1: ", e)
PY
}
prof() { tag=$1; shift
  cd /tmp; env "$@" timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/prof_$tag -o kt -- python $ROOT/bench.py --k 27 $Q > $ROOT/$OUT/$tag.json 2> $ROOT/$OUT/$tag.err; cd $ROOT
  find $OUT/prof_$tag -name "*kernel_trace.csv" -delete
  find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -r grep "k_bucket_rank\|k_giant\|k_onesweep\|k_compact<" | cut -c1-28,150-240
  show $tag
}
prof uni A=1
prof skew KMC_SYNTH_REPEATS=10000:2000:10
prof spec KMC_SYNTH_REPEATS=300:100000:120,6000:5000:20,171:100000:20,H20000
timeout 900 python -m pytest tests -m gpu -x -q -k "rank or repeat or giant or skew or many_bins or collaps" > $OUT/pytest_k.txt 2>&1; tail -3 $OUT/pytest_k.txt

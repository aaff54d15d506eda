#!/bin/bash
# round 5, session c: where does the skew leg's finisher time go (kernel statistics, both finishers); the repeat spectrum; the rank kernel at one workgroup per CU beside a second stream
OUT=gpurun_out/r05c; mkdir -p $OUT /dev/shm/kmccache
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$PWD
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest"
show() { python - <<PY
import json
try:
    d=json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("   $1: value %.2f, %.1f ms/step, local_sort %.3f ms, onesweep %.1f us, oracle %s, giant %s/%s" % (d["value"], d["ms_per_step"], d["local_sort"]["avg_launch_ms"], 1e3*d["roofline"]["avg_launch_ms"], d["self_check"].get("oracle_bins_equal"), d["sort_path"]["groups_by_path"]["giant_tiles"], d["sort_path"]["groups_by_path"]["giant_records"]))
except Exception as e: print("   $1: ", e)
PY
}
prof() { # tag env...
  tag=$1; shift
  cd /tmp; env "$@" timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/prof_$tag -o kt -- python $ROOT/bench.py --k 27 $Q --no-oracle-check > $ROOT/$OUT/$tag.json 2> $ROOT/$OUT/$tag.err; cd $ROOT
  find $OUT/prof_$tag -name "*kernel_trace.csv" -delete
  find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-40,150-250
  show $tag
}
prof skew_c0 KMC_HIP_RANK_COLLAPSE=0 KMC_SYNTH_REPEATS=10000:2000:10
prof skew_c1 KMC_HIP_RANK_COLLAPSE=1 KMC_SYNTH_REPEATS=10000:2000:10
SPEC=300:100000:120,6000:5000:20,171:100000:20,H20000
prof spec_c0 KMC_HIP_RANK_COLLAPSE=0 KMC_SYNTH_REPEATS=$SPEC
prof spec_c1 KMC_HIP_RANK_COLLAPSE=1 KMC_SYNTH_REPEATS=$SPEC
run() { tag=$1; n=$2; shift 2; env "$@" timeout 600 python bench.py --cache /dev/shm/kmccache --k 27 $Q --streams $n --no-oracle-check > $OUT/$tag.json 2> $OUT/$tag.err; show $tag; }
run s1_pad0 1 KMC_HIP_RANK_LDS_PAD=0
run s2_pad0 2 KMC_HIP_RANK_LDS_PAD=0
run s1_pad16 1 KMC_HIP_RANK_LDS_PAD=16384
run s2_pad16 2 KMC_HIP_RANK_LDS_PAD=16384
run s3_pad16 3 KMC_HIP_RANK_LDS_PAD=16384
run s2_pad32 2 KMC_HIP_RANK_LDS_PAD=32768

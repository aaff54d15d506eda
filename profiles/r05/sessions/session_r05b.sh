#!/bin/bash
# round 5, session b: the collapsing finisher (k_bucket_rank_c) against round 4's, every record width + the skew leg; kernel statistics of the new default
OUT=gpurun_out/r05b; mkdir -p $OUT /dev/shm/kmccache
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest"
run() { # tag k env...
  tag=$1; k=$2; shift 2
  env "$@" timeout 600 python bench.py --k $k $Q > $OUT/$tag.json 2> $OUT/$tag.err
  python tools/pj.py $OUT/$tag.json 2>&1 | cut -c1-200
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1])
    print("   $tag: value %.2f, local_sort %.3f ms, onesweep %.1f us, oracle %s, paths %s" % (d["value"], d["local_sort"]["avg_launch_ms"], 1e3*d["roofline"]["avg_launch_ms"], d["self_check"].get("oracle_bins_equal"), d["sort_path"]["groups_by_path"]))
except Exception as e: print("   $tag: ", e)
PY
}
run k27_c1 27 KMC_HIP_RANK_COLLAPSE=1
run k27_c0 27 KMC_HIP_RANK_COLLAPSE=0
run skew_c1 27 KMC_HIP_RANK_COLLAPSE=1 KMC_SYNTH_REPEATS=10000:2000:10
run skew_c0 27 KMC_HIP_RANK_COLLAPSE=0 KMC_SYNTH_REPEATS=10000:2000:10
run k55_c1 55 KMC_HIP_RANK_COLLAPSE=1
run k55_c0 55 KMC_HIP_RANK_COLLAPSE=0
run k127_c1 127 KMC_HIP_RANK_COLLAPSE=1
run k127_c0 127 KMC_HIP_RANK_COLLAPSE=0
bash tools/gpu_session.sh r05b profk:27:KMC_HIP_RANK_COLLAPSE=1 tests

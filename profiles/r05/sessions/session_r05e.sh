#!/bin/bash
# round 5, session e: k_bucket_rank cut after its phases (tools/build_variants.py br1 br2 br4), uniform and one-family skew quarter workloads: which phase grows with the repeats?
OUT=gpurun_out/r05e; mkdir -p $OUT /dev/shm/kmccache
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$PWD
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest --no-oracle-check"
prof() { # tag lib env...
  tag=$1; lib=$2; shift 2
  cd /tmp; env KMC_HIP_LIB=$ROOT/$lib "$@" timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/prof_$tag -o kt -- python $ROOT/bench.py --k 27 $Q > $ROOT/$OUT/$tag.json 2> $ROOT/$OUT/$tag.err; cd $ROOT
  find $OUT/prof_$tag -name "*kernel_trace.csv" -delete
  echo "== $tag"; find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -r grep "k_bucket_rank\|k_giant" | cut -c1-28,150-230
}
for v in br1 br2 br4; do
  prof uni_$v kmc_amd/variants/libkmc_hip_$v.so A=1
  prof skew_$v kmc_amd/variants/libkmc_hip_$v.so KMC_SYNTH_REPEATS=10000:2000:10
done
prof skew_base kmc_amd/libkmc_hip.so KMC_SYNTH_REPEATS=10000:2000:10

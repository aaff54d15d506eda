#!/bin/bash
OUT=gpurun_out/r05a; mkdir -p $OUT
bash tools/gpu_session.sh r05a facts tests
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err ) 2> $OUT/bench_driver.time
tail -n 1 $OUT/bench_driver.out | wc -c
tail -n 1 $OUT/bench_driver.out | cut -c1-600
cp bench_detail.json $OUT/bench_detail_driver.json 2>/dev/null
cat $OUT/bench_driver.time
bash tools/gpu_session.sh r05a qs:1:KMC_HIP_GROUP=4 qs:2:KMC_HIP_GROUP=4 qs:1:KMC_HIP_GROUP=2 qs:2:KMC_HIP_GROUP=2 qs:1:KMC_HIP_GROUP=1 qs:2:KMC_HIP_GROUP=1 qs:4:KMC_HIP_GROUP=1

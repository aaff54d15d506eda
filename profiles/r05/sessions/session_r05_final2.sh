#!/bin/bash
# round 5, final session 1: the -m gpu suite, then bench.py exactly as the driver runs it (the line it prints, its length, the detail file)
OUT=gpurun_out/r05z5; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt | head -2
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err ) 2> $OUT/bench_driver.time
tail -n 1 $OUT/bench_driver.out | wc -c
tail -n 1 $OUT/bench_driver.out | cut -c1-400
cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
cat $OUT/bench_driver.time; tail -12 $OUT/bench_driver.err | cut -c1-200

#!/bin/bash
# round 5, last session: the tree with the new walk of k_bucket_rank<1> — the -m gpu suite, bench.py exactly as the driver runs it, then (time permitting) rocprofv3 kernel
# statistics and the SQ instruction counters of the quarter workload
OUT=gpurun_out/r05z6; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 330 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.txt | tail -2
( time timeout 360 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err ) 2> $OUT/bench_driver.time
tail -n 1 $OUT/bench_driver.out | wc -c
tail -n 1 $OUT/bench_driver.out | cut -c1-300
cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
grep real $OUT/bench_driver.time; tail -12 $OUT/bench_driver.err | cut -c1-200
left() { echo $(( 700 - ( $(date +%s) - T0 ) )); }
if [ $(left) -gt 75 ]; then bash tools/gpu_session.sh r05z6 profk:27:A=1 2>&1 | tail -12; fi
if [ $(left) -gt 75 ]; then bash tools/gpu_session.sh r05z6 pmck:27:SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVES:A=1 2>&1 | tail -12; fi
echo "elapsed $(( $(date +%s) - T0 )) s"

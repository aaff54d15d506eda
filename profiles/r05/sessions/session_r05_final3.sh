#!/bin/bash
# round 5, final session 3: the e2e_large leg at the configs[2] size (30 Gbp FASTQ: reference vs drop-in, RAM-only mode; the reference's bins device-resident)
OUT=gpurun_out/r05z3; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1700 python tools/e2e_large_run.py 30 27 > $OUT/e2e_large_30gbp.json 2> $OUT/e2e_large_30gbp.err; tail -c 1500 $OUT/e2e_large_30gbp.json; tail -3 $OUT/e2e_large_30gbp.err

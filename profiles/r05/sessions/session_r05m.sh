#!/bin/bash
OUT=gpurun_out/r05m; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/debug/bigbucket_locate.py 2>&1 | tail -30 | cut -c1-400 | tee $OUT/locate.txt

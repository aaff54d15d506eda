#!/bin/bash
OUT=gpurun_out/r05j; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== shipped"; timeout 300 python tools/debug/bigbucket_gpu.py 2>&1 | tail -32 | cut -c1-200 | tee $OUT/big.txt

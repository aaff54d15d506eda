#!/bin/bash
OUT=gpurun_out/r05i; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/debug/skew_bin_diff.py 2>&1 | tail -40 | tee $OUT/diff.txt

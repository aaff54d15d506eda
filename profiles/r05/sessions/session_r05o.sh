#!/bin/bash
OUT=gpurun_out/r05o; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 BB_QUICK=1
KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_dbgI.so timeout 300 python tools/debug/bigbucket_gpu.py > $OUT/dbg.txt 2>&1; grep -c "HEAD idx" $OUT/dbg.txt; grep "HEAD idx" $OUT/dbg.txt | sort | head -12 | cut -c1-330; tail -3 $OUT/dbg.txt | cut -c1-150

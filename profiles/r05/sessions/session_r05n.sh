#!/bin/bash
OUT=gpurun_out/r05n; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 BB_QUICK=1
for v in dbgS dbgP dbgI; do echo "== $v"; KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_$v.so timeout 300 python tools/debug/bigbucket_gpu.py 2>&1 | tail -6 | cut -c1-160; done | tee $OUT/dbg.txt

#!/bin/bash
OUT=gpurun_out/r05g; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== shipped"; timeout 300 python tools/debug/bigbucket_gpu.py 2>&1 | tail -20 | tee $OUT/big.txt
echo "== BR_BIG off"; KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_nobig.so timeout 300 python tools/debug/bigbucket_gpu.py 2>&1 | tail -20 | tee $OUT/nobig.txt

#!/bin/bash
OUT=gpurun_out/r05l; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export BB_CASES="5000:4:0 5000:5:0 5000:6:0 5000:7:0 5000:8:0 5000:9:0 5000:12:0 1000:6:0 20000:6:0"
echo "== shipped"; timeout 300 python tools/debug/bigbucket_gpu.py 2>&1 | tail -40 | cut -c1-130 | tee $OUT/a.txt
echo "== nobig"; KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_nobig.so timeout 300 python tools/debug/bigbucket_gpu.py 2>&1 | tail -40 | cut -c1-130 | grep -v "True, True, True" | tee $OUT/b.txt
echo "== shipped GROUP=1"; KMC_HIP_GROUP=1 timeout 300 python tools/debug/bigbucket_gpu.py 2>&1 | tail -40 | cut -c1-130 | grep -v "True, True, True"| tee $OUT/c.txt

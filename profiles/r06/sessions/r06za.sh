#!/bin/bash
# round 6, session za: is the 37.3 of session z (three groups in flight, 12 KB of extra dynamic LDS on k_onesweep) real? Repeats and neighbours, one box
OUT=gpurun_out/r06za; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_session.sh r06za qs:1:A=1 qs:3:A=1 qs:3:KMC_HIP_SCATTER_LDS_PAD=12288 qs:3:A=2 qs:3:KMC_HIP_SCATTER_LDS_PAD=12289 qs:3:A=3 qs:3:KMC_HIP_SCATTER_LDS_PAD=12290 qs:2:KMC_HIP_SCATTER_LDS_PAD=12288 qs:1:KMC_HIP_SCATTER_LDS_PAD=12288 qs:3:KMC_HIP_SCATTER_LDS_PAD=8192 qs:3:KMC_HIP_SCATTER_LDS_PAD=16384 qs:3:KMC_HIP_SCATTER_LDS_PAD=20480 qs:1:A=2 2>&1 | grep -v "^\[qs" | cut -c1-150

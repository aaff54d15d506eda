#!/bin/bash
# round 6, session h: rocprofv3 kernel statistics of the spectrum leg, BR_MID 384 (base) and 768
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
SPEC=300:100000:120,6000:5000:20,171:100000:20,H20000
bash tools/gpu_session.sh r06h profk:27:KMC_SYNTH_REPEATS=$SPEC > /dev/null 2>&1
mv gpurun_out/r06h/profk_27_KMC_SYNTH_REPEATS=$SPEC gpurun_out/r06h/spec_base
export KMC_HIP_LIB=$PWD/kmc_amd/variants/libkmc_hip_brmid768.so
bash tools/gpu_session.sh r06h profk:27:KMC_SYNTH_REPEATS=$SPEC > /dev/null 2>&1
mv gpurun_out/r06h/profk_27_KMC_SYNTH_REPEATS=$SPEC gpurun_out/r06h/spec_brmid768
ls gpurun_out/r06h

#!/bin/bash
# round 6, session j: rocprofv3 kernel statistics: uniform, skew, spectrum with k_bucket_detect + the arena (BR_MID 255)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
SPEC=300:100000:120,6000:5000:20,171:100000:20,H20000
bash tools/gpu_session.sh r06j profk:27:A=1 profk:27:KMC_SYNTH_REPEATS=10000:2000:10 profk:27:KMC_SYNTH_REPEATS=$SPEC > /dev/null 2>&1
mv "gpurun_out/r06j/profk_27_A=1" gpurun_out/r06j/uniform; mv "gpurun_out/r06j/profk_27_KMC_SYNTH_REPEATS=10000:2000:10" gpurun_out/r06j/skew; mv "gpurun_out/r06j/profk_27_KMC_SYNTH_REPEATS=$SPEC" gpurun_out/r06j/spectrum
ls gpurun_out/r06j

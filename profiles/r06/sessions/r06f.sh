#!/bin/bash
# round 6, session f: rocprofv3 kernel statistics of the skew and spectrum legs with the arena path
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_session.sh r06f profk:27:KMC_SYNTH_REPEATS=10000:2000:10 profk:27:KMC_SYNTH_REPEATS=300:100000:120,6000:5000:20,171:100000:20,H20000 2>&1 | tail -50

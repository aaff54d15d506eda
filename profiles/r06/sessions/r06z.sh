#!/bin/bash
# round 6, session z: (1) the drop-in tests with 8 completer writers as the default; (2) ONE scatter workgroup per CU (KMC_HIP_SCATTER_LDS_PAD: dynamic LDS beyond what
# k_onesweep uses) so that a finisher workgroup of another group in flight fits beside it: quarter workload, 1-4 groups in flight, with and without the pad
OUT=gpurun_out/r06z; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stage1_e2e.py -m gpu -q -k "writes_the_reference_database or reader_plugin or stage1 or dropin or kff" > $OUT/pytest_dropin.txt 2>&1; tail -2 $OUT/pytest_dropin.txt | cut -c1-200
bash tools/gpu_session.sh r06z qs:1:A=1 qs:2:A=1 qs:3:A=1 qs:1:KMC_HIP_SCATTER_LDS_PAD=24576 qs:2:KMC_HIP_SCATTER_LDS_PAD=24576 qs:3:KMC_HIP_SCATTER_LDS_PAD=24576 qs:4:KMC_HIP_SCATTER_LDS_PAD=24576 qs:3:KMC_HIP_SCATTER_LDS_PAD=12288 2>&1 | cut -c1-170
